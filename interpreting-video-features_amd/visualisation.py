"""Drop-in for the saliency-path functions of video_features_pytorch/visualisation.py
(SURVEY 8f N3): `create_image_arrays`, `vizualize_results_on_gradcam`,
`find_temp_mask_red_dots`, `vizualize_results` -- same names, arguments, return values and
side effects, with the pixel work in csrc/viz.hip.  Who sees the 0/1 snap of the dot row
(visualisation.py:77-81): `find_temp_mask_red_dots` snaps the tensor it is GIVEN, in place;
`vizualize_results_on_gradcam` first rebinds `mask = mask.detach().cpu()` (:39), which for the
drivers' CUDA masks (smth:216) is a COPY -- so the snap lands on that host copy (and in the
MASKVALS file), never on the caller's tensor.  This module only takes GPU masks, hence the
caller's mask is always left as it was.

The reference's two undefined names are resolved the obvious way: `perturb_sequence`
(visualisation.py:115) is `mask.perturb_sequence`, `args.subDir` (:10) becomes the
`subDir` keyword.  Image files are written with PIL (OpenCV is not a dependency here):
`cv2.imwrite(img%02d.jpg)` -> PIL JPEG, ImageMagick `convert` -> PIL animated GIF.
`cv2.applyColorMap(., COLORMAP_JET)` is a 256-entry BGR table built from the colour map's
definition (MATLAB jet(64), linearly interpolated) -- the one step of this module whose
parity with OpenCV is unpinned.  RESIZE_FLAG must be 0, as both drivers hard-code it
(smth:294, KTH:352).
"""
import os

import numpy as np
import torch

import ivf_lib as L
import mask as _mask

_LUT = {}


def jet_lut_bgr():
    """[256,3] uint8 BGR table of COLORMAP_JET."""
    r = [0] * 24 + [0.0625 * k for k in range(1, 17)] + [1] * 16 + [1 - 0.0625 * k for k in range(1, 9)]
    g = [0] * 8 + [0.0625 * k for k in range(1, 17)] + [1] * 16 + [1 - 0.0625 * k for k in range(1, 16)] + [0] * 9
    b = [0.5 + 0.0625 * k for k in range(1, 9)] + [1] * 16 + [1 - 0.0625 * k for k in range(1, 16)] + [0] * 25
    x64, x256 = np.linspace(0, 1, 64), np.linspace(0, 1, 256)
    rgb = np.stack([np.interp(x256, x64, c) for c in (r, g, b)], axis=1)
    return np.ascontiguousarray(np.rint(rgb * 255).astype(np.uint8)[:, ::-1])


def _lut(device):
    key = str(device)
    if key not in _LUT:
        _LUT[key] = torch.from_numpy(jet_lut_bgr()).to(device)
    return _LUT[key]


def find_temp_mask_red_dots(imageWidth, imageHeight, mask, roundUpMask):
    """visualisation.py:67-93: dot geometry and colour channel per mask entry; with roundUpMask the
    CALLER's mask is snapped to 0/1 in place."""
    maskLen = len(mask)
    dotWidth = int(imageWidth // (maskLen + 4))
    dotPadding = int((imageWidth - (dotWidth * maskLen)) // maskLen)
    dotHeight = int(imageHeight // 20)
    if roundUpMask:
        with torch.no_grad():
            mask.copy_((mask > 0.5).to(mask.dtype))
    host = mask.detach().cpu()
    dots = []
    for i in range(maskLen):
        dots.append({'yStart': -dotHeight, 'yEnd': imageHeight, 'xStart': i * (dotWidth + dotPadding),
                     'xEnd': i * (dotWidth + dotPadding) + dotWidth, 'channel': 1 if host[i] == 0 else 2})
    return dots


def _save_pngs(strip_bgr, rootDir, case):
    from PIL import Image
    for i in range(strip_bgr.shape[0]):
        Image.fromarray(np.ascontiguousarray(strip_bgr[i][:, :, ::-1]), mode="RGB").save(
            os.path.join(rootDir, "case" + case + "_" + str(i) + ".png"))


def vizualize_results_on_gradcam(gradCamImage, mask, rootDir, case="0", roundUpMask=True, imageWidth=224,
                                 imageHeight=224):
    """visualisation.py:35-64.  gradCamImage [3,T,H,3W] uint8 (BGR planes; numpy, or a CUDA tensor) is
    modified in place like the reference's array.  `mask` [T] on the GPU is NOT modified: the reference
    snaps its host copy (`mask.detach().cpu()`, :39), which is what the dots and the MASKVALS file show."""
    L.require_gpu(mask)
    if not os.path.exists(rootDir):
        os.makedirs(rootDir)
    dev = mask.device
    mask = mask.detach().cpu().clone()                                           # :39 (a copy for a CUDA tensor)
    find_temp_mask_red_dots(imageWidth, imageHeight, mask, roundUpMask)          # snaps the host copy
    is_np = isinstance(gradCamImage, np.ndarray)
    planes = torch.from_numpy(gradCamImage).to(dev) if is_np else gradCamImage
    strip = planes.permute(1, 2, 3, 0).contiguous()                              # [T,H,3W,3]
    T, H, W3 = strip.shape[:3]
    m = L.f32c(mask.to(dev))
    with torch.cuda.device(dev):
        L.check(L.lib().ivf_viz_dots(L.ptr(strip), L.ptr(m), T, H, W3, int(imageWidth), int(imageHeight), L.stream()))
    host = strip.cpu().numpy()
    if is_np:
        gradCamImage[...] = host.transpose(3, 0, 1, 2)
    else:
        gradCamImage.copy_(strip.permute(3, 0, 1, 2))
    _save_pngs(host, rootDir, case)
    with open(os.path.join(rootDir, "MASKVALScase" + case + ".txt"), "w+") as f:
        f.write(str(mask))
    return gradCamImage


def create_image_arrays(input_sequence, gradcamMask, timeMask, intraBidx, temporalMaskType, output_folder, targTag,
                        RESIZE_FLAG, RESIZE_SIZE_WIDTH, RESIZE_SIZE_HEIGHT):
    """visualisation.py:96-130.  input_sequence [B,3,T,H,W] float 0..255 on the GPU, gradcamMask [T,H,W]
    float32 (numpy or tensor), timeMask [T] on the GPU (left unchanged: the dot row snaps a host copy).  Writes
    img%02d.jpg, mygif.gif, case<type><tag>_<i>.png and MASKVALScase<type><tag>.txt into output_folder and
    returns the [3,T,H,3W] uint8 array (BGR planes, dots included)."""
    from PIL import Image
    L.require_gpu(input_sequence, timeMask)
    if RESIZE_FLAG:
        raise L.IvfError("create_image_arrays: RESIZE_FLAG != 0 is not built (both reference drivers pass 0)")
    dev = input_sequence.device
    clip = L.f32c(input_sequence[intraBidx])
    C, T, H, W = clip.shape
    if C != 3:
        raise L.IvfError("create_image_arrays expects 3-channel clips")
    cam = L.f32c(torch.as_tensor(gradcamMask).to(dev))
    if tuple(cam.shape) != (T, H, W):
        raise L.IvfError(f"gradcamMask must be [{T},{H},{W}], got {tuple(cam.shape)}")
    # the third panel: the clip perturbed with the SNAPPED mask (snap_values on a clone), visualisation.py:115-117
    pert = L.f32c(_mask.perturb_sequence(input_sequence, timeMask.detach().clone(),
                                         perturbation_type=temporalMaskType, snap_values=True)[intraBidx])
    strip = torch.empty(T, H, 3 * W, 3, dtype=torch.uint8, device=dev)
    fmax = torch.empty(T, device=dev)
    with torch.cuda.device(dev):
        L.check(L.lib().ivf_viz_blend(L.ptr(clip), L.ptr(cam), L.ptr(pert), L.ptr(_lut(dev)), L.ptr(fmax),
                                      L.ptr(strip), T, H, W, L.stream()))
    host = strip.cpu().numpy()
    os.makedirs(output_folder, exist_ok=True)
    frames = []
    for i in range(T):
        im = Image.fromarray(np.ascontiguousarray(host[i][:, :, ::-1]), mode="RGB")
        im.save(os.path.join(output_folder, "img%02d.jpg" % (i + 1)))
        frames.append(im)
    frames[0].save(os.path.join(output_folder, "mygif.gif"), save_all=True, append_images=frames[1:],
                   duration=100, loop=0)                                          # convert -delay 10 -loop 0
    combined_images = np.ascontiguousarray(host.transpose(3, 0, 1, 2))            # :127
    vizualize_results_on_gradcam(combined_images, timeMask, rootDir=output_folder,
                                 case=temporalMaskType + str(targTag))
    return combined_images


def vizualize_results(orig_seq, pert_seq, mask, rootDir=None, case="0", markImgs=True, iterTest=False,
                      subDir="run0"):
    """visualisation.py:8-32: the perturbed frames as PNGs with the mask value painted into the top-left
    10 x 10 square of the red channel (host-side file writing only)."""
    from PIL import Image
    if rootDir is None:
        rootDir = "vizualisations/" + subDir + "/"
    rootDir += "/PerturbImgs/"
    if not os.path.exists(rootDir):
        os.makedirs(rootDir)
    pertPy = pert_seq.cpu().detach().numpy().copy()
    mk = mask.detach().cpu()
    for i in range(orig_seq.shape[1]):
        if markImgs:
            pertPy[1:, i, :10, :10] = 0
            pertPy[0, i, :10, :10] = float(mk[i]) * 255
        Image.fromarray(pertPy[:, i, :, :].transpose(1, 2, 0).astype(np.uint8)).save(
            rootDir + "case" + case + "pert" + str(i) + ".png")
    with open(rootDir + "case" + case + ".txt", "w+") as f:
        f.write(str(mk))
