"""Oracle (test infrastructure): Grad-CAM for video, numpy/torch CPU.

Follows video_features_pytorch/grad_cam_videos.py:64-142 (GradCamVideo.__call__)
with the feature split of pytorch-grad-cam/grad-cam.py:23-54.
"""
import numpy as np
import torch

from . import i3d_ref


def resize_bilinear(img, width, height):
    """OpenCV INTER_LINEAR for float32 (call site grad_cam_videos.py:119-120;
    dsize is (width, height)): half-pixel centres, source index clamped, weights
    in float32.  PARITY UNPINNED: OpenCV is not installed in the build container."""
    img = np.asarray(img, dtype=np.float32)
    sh, sw = img.shape

    def coeffs(dst, src):
        scale = np.float32(src) / np.float32(dst)
        f = (np.arange(dst, dtype=np.float32) + np.float32(0.5)) * scale - np.float32(0.5)
        i0 = np.floor(f).astype(np.int64)
        frac = (f - i0.astype(np.float32)).astype(np.float32)
        lo = i0 < 0
        frac[lo] = 0
        i0[lo] = 0
        hi = i0 >= src - 1
        frac[hi] = 0
        i0[hi] = src - 1
        i1 = np.minimum(i0 + 1, src - 1)
        return i0, i1, frac

    y0, y1, fy = coeffs(height, sh)
    x0, x1, fx = coeffs(width, sw)
    fx = fx[None, :]
    top = img[y0][:, x0] * (np.float32(1) - fx) + img[y0][:, x1] * fx
    bot = img[y1][:, x0] * (np.float32(1) - fx) + img[y1][:, x1] * fx
    fy = fy[:, None]
    return (top * (np.float32(1) - fy) + bot * fy).astype(np.float32)


def cam_from_activations(act, grad, clip_size, width, height, normalize_per_frame):
    """grad_cam_videos.py:96-140.  act,grad [C,T',h,w] numpy -> [T,H,W] float32."""
    weights = grad.mean(axis=(1, 2, 3))                        # :98
    cam = np.zeros(act.shape[1:], dtype=np.float32)
    for i, w in enumerate(weights):                            # :103-108 (sequential fp32 sum)
        cam += w * act[i]
    cam = np.maximum(cam, 0)                                   # :110
    step = clip_size // act.shape[1]                           # :113
    blocks = []
    for i in range(cam.shape[0]):                              # :116-125
        m = resize_bilinear(cam[i], width, height)
        blocks.append(np.repeat(m[None], step, axis=0))
    vid = np.array(blocks)
    with np.errstate(invalid='ignore', divide='ignore'):
        if normalize_per_frame:                                # :128-132 (per slice-block)
            for i in range(vid.shape[0]):
                vid[i] = vid[i] - np.min(vid[i])
                vid[i] = vid[i] / np.max(vid[i])
        else:                                                  # :134-135
            vid = vid - np.min(vid)
            vid = vid / np.max(vid)
    if vid.shape[0] > 1:
        vid = np.concatenate(vid, axis=0)                      # :137-138
    elif vid.shape[0] == 1:
        vid = np.squeeze(vid, 0)
    return vid.astype(np.float32), weights.astype(np.float32), cam


def gradcam_i3d(x, sd, index=None, pool_kernel=(2, 7, 7), softmax=True,
                width=224, height=224, normalize_per_frame=True, layer='Mixed_5c', **kw):
    """GradCamVideo.__call__ for archType 'I3D' and ONE target layer (any endpoint: the hook of
    pytorch-grad-cam/grad-cam.py:50-51 sits on the named module's output).
    x [1,C,T,H,W].  Returns (cam_vid [T,H,W], output [1,K], extras)."""
    if layer != 'Mixed_5c':
        eps = {}
        xin = x.detach().clone().requires_grad_()
        last = i3d_ref.features(xin, sd, endpoints=eps, **kw)
        feat = eps[layer]
        _, out = i3d_ref.head(last, sd, pool_kernel, softmax)
        if index is None:
            index = int(np.argmax(out.detach().numpy()))
        grad, = torch.autograd.grad(out[0, index], feat)
        vid, weights, cam = cam_from_activations(
            feat.detach().numpy()[0], grad.numpy()[0], x.shape[2], width, height, normalize_per_frame)
        return vid, out.detach(), dict(weights=weights, cam=cam, index=index, feat=feat.detach(), grad=grad)
    feat = i3d_ref.features(x, sd, **kw).detach().requires_grad_()
    _, out = i3d_ref.head(feat, sd, pool_kernel, softmax)
    if index is None:
        index = int(np.argmax(out.detach().numpy()))          # :69-70
    score = out[0, index] if out.shape[0] == 1 else out.reshape(-1)[index]
    grad, = torch.autograd.grad(score, feat)                   # :73-83
    vid, weights, cam = cam_from_activations(
        feat.detach().numpy()[0], grad.numpy()[0], x.shape[2], width, height, normalize_per_frame)
    return vid, out.detach(), dict(weights=weights, cam=cam, index=index,
                                   feat=feat.detach(), grad=grad)
