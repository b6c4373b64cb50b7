"""Oracle (test infrastructure): the visualisation blend, numpy CPU.

Follows video_features_pytorch/visualisation.py: create_image_arrays :96-130,
vizualize_results_on_gradcam :35-64, find_temp_mask_red_dots :67-93.  The two names the
reference leaves undefined (`perturb_sequence` at :115, `args` at :10) are taken from
mask.py and from the caller.  OpenCV is absent in the build container, so
`cv2.applyColorMap(., COLORMAP_JET)` is restated from the colour map's definition (MATLAB
jet(64) interpolated linearly to 256 entries, rounded to uint8, BGR order) -- PARITY
UNPINNED for that table; everything around it is pinned by tests/golden/viz.npz, which the
reference's own functions produced with this table bound as cv2.applyColorMap.
"""
import numpy as np


def jet_lut_bgr():
    """[256,3] uint8, BGR, COLORMAP_JET."""
    r = [0] * 24 + [0.0625 * k for k in range(1, 17)] + [1] * 16 + [1 - 0.0625 * k for k in range(1, 9)]
    g = [0] * 8 + [0.0625 * k for k in range(1, 17)] + [1] * 16 + [1 - 0.0625 * k for k in range(1, 16)] + [0] * 9
    b = [0.5 + 0.0625 * k for k in range(1, 9)] + [1] * 16 + [1 - 0.0625 * k for k in range(1, 16)] + [0] * 25
    x64, x256 = np.linspace(0, 1, 64), np.linspace(0, 1, 256)
    rgb = np.stack([np.interp(x256, x64, c) for c in (r, g, b)], axis=1)
    return np.rint(rgb * 255).astype(np.uint8)[:, ::-1].copy()


def apply_colormap_jet(gray_u8):
    """cv2.applyColorMap(gray_u8, cv2.COLORMAP_JET): [H,W] uint8 -> [H,W,3] uint8 BGR."""
    return jet_lut_bgr()[np.asarray(gray_u8, dtype=np.uint8)]


def find_temp_mask_red_dots(image_width, image_height, mask, round_up_mask):
    """visualisation.py:67-93.  `mask` (numpy [T], modified IN PLACE when round_up_mask, as the
    reference does to the caller's tensor).  Returns the dot dicts."""
    n = len(mask)
    dot_w = int(image_width // (n + 4))
    dot_pad = int((image_width - dot_w * n) // n)
    dot_h = int(image_height // 20)
    dots = []
    for i in range(n):
        if round_up_mask:
            mask[i] = 1 if mask[i] > 0.5 else 0
        dots.append({'yStart': -dot_h, 'yEnd': image_height, 'xStart': i * (dot_w + dot_pad),
                     'xEnd': i * (dot_w + dot_pad) + dot_w, 'channel': 1 if mask[i] == 0 else 2})
    return dots


def draw_dots(img, mask, image_width=224, image_height=224, round_up_mask=True):
    """vizualize_results_on_gradcam :35-64 on img [3,T,H,3W] uint8 (BGR planes), in place.  The
    reference keeps its DEFAULT imageWidth/imageHeight of 224 whatever the frame size (the drivers
    do not pass them): numpy clips the slices that run past the image."""
    dots = find_temp_mask_red_dots(image_width, image_height, mask, round_up_mask)
    off = image_width * 2
    for i in range(len(mask)):
        for j, d in enumerate(dots):
            inten = 255 if i == j else 150
            img[:, i, d['yStart']:, off + d['xStart']:off + d['xEnd']] = 0
            img[d['channel'], i, d['yStart']:, off + d['xStart']:off + d['xEnd']] = inten
    return img


def combine_frames(clip, cam, perturbed):
    """create_image_arrays :96-130 with RESIZE_FLAG = 0 (both drivers), before the dots.
    clip, perturbed [3,T,H,W] float32 (RGB planes, 0..255); cam [T,H,W] float32 in [0,1].
    Returns [T,H,3W,3] uint8 BGR: original | heat-map overlay | perturbed."""
    T = clip.shape[1]
    img_all = np.flip(np.transpose(clip, (1, 2, 3, 0)), 3)            # [T,H,W,3] BGR  (:99-100)
    pert_all = np.transpose(perturbed, (1, 2, 3, 0))                 # RGB; [:, :, ::-1] below
    out = []
    with np.errstate(invalid='ignore'):
        for i in range(T):
            img = img_all[i]
            heat = apply_colormap_jet(np.uint8(255 * cam[i]))        # :104
            c = np.float32(heat) + np.float32(img)                   # :108-109
            c = c / np.max(c)                                        # :110
            out.append(np.concatenate((np.uint8(img), np.uint8(255 * c), np.uint8(pert_all[i])[:, :, ::-1]), axis=1))
    return np.array(out)
