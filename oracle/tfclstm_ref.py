"""Oracle (test infrastructure) for the TF-style ConvLSTM search variant.

PARITY UNPINNED.  The reference's TensorFlow half (video_features_tf/models/clstm.py:9-52, 87-126;
mask/find_mask_kth.py:300-372, 431-452; mask/gradcam.py:28-111) needs TensorFlow 1.12 with Keras and
cannot run in the build container; no fixture of it exists.  This file restates, in torch CPU fp32, the
published Keras ConvLSTM2D arithmetic (TF 1.12 keras/layers/convolutional_recurrent.py: ConvLSTM2DCell.call)
as that call site configures it, and the reference's own host code around it (freeze recurrence, TV / L1,
tf.train.AdamOptimizer, init_mask through the sigmoid, per-frame Grad-CAM).  It checks the HIP extension's
INTERNAL consistency (forward, BPTT via autograd, search, Grad-CAM) -- it is not evidence of parity with TF.
"""
import numpy as np
import torch
import torch.nn.functional as F


def hard_sigmoid(z):
    """keras.backend.hard_sigmoid: clip(0.2 z + 0.5, 0, 1) (the recurrent_activation default of TF 1.12)."""
    return torch.clamp(0.2 * z + 0.5, 0.0, 1.0)


def _pad_same(n, k, s):
    out = -(-n // s)
    total = max((out - 1) * s + k - n, 0)
    return total // 2, total - total // 2


def conv2d_tf(x, w_hwio, stride, padding):
    """tf.nn.conv2d on NCHW x with a Keras HWIO kernel; padding 'valid' | 'same' (TensorFlow's asymmetric rule)."""
    w = w_hwio.permute(3, 2, 0, 1)
    if padding == 'same':
        pt, pb = _pad_same(x.shape[2], w.shape[2], stride)
        pl, pr = _pad_same(x.shape[3], w.shape[3], stride)
        x = F.pad(x, (pl, pr, pt, pb))
    return F.conv2d(x, w, stride=stride)


def clstm_block(x, kernel, rkernel, bias, stride, padding, hard=True):
    """clstm.py:9-52 with pooling 'max', no batch norm: x [B,T,C,H,W] -> (pooled [B,T,F,Hp,Wp], outputs [B,T,F,Ho,Wo])."""
    B, T = x.shape[:2]
    Fu = rkernel.shape[2]
    act = hard_sigmoid if hard else torch.sigmoid
    h = c = None
    outs = []
    for t in range(T):
        z = conv2d_tf(x[:, t], kernel, stride, padding) + bias.view(1, -1, 1, 1)
        if h is not None:
            z = z + conv2d_tf(h, rkernel, 1, 'same')
        zi, zf, zc, zo = torch.split(z, Fu, dim=1)                 # gate order i, f, c, o
        i, f, o = act(zi), act(zf), act(zo)
        g = torch.tanh(zc)
        c = i * g if c is None else f * c + i * g
        h = o * torch.tanh(c)
        outs.append(h)
    out = torch.stack(outs, dim=1)
    Ho, Wo = out.shape[-2:]
    pooled = F.max_pool2d(out.reshape(B * T, Fu, Ho, Wo), 2).reshape(B, T, Fu, Ho // 2, Wo // 2)
    return pooled, out


def model(x_ncthw, weights, stride=2, padding='valid', hard=True, only_last=True):
    """clstm.clstm(x, bn=False): returns (logits [B,K], output sequence of the last ConvLSTM2D [B,T,F,Ho,Wo])."""
    x = x_ncthw.permute(0, 2, 1, 3, 4)                              # [B,T,C,H,W]
    last = None
    for (k, rk, b) in weights['layers']:
        x, last = clstm_block(x, k, rk, b, stride, padding, hard)
    B, T, Fu, Hp, Wp = x.shape
    nhwc = x.permute(0, 1, 3, 4, 2)                                 # tf.layers.flatten of NHWC maps
    flat = nhwc[:, -1].reshape(B, -1) if only_last else nhwc.reshape(B, -1)
    return flat @ weights['dense_w'] + weights['dense_b'], last


def freeze(x_ncthw, m):
    """the tf.scan recurrence of find_mask_kth.py:318-327 (mask values as given; m[0] unused)."""
    T = x_ncthw.shape[2]
    frames = [x_ncthw[:, :, 0]]
    for u in range(1, T):
        frames.append((1 - m[u]) * x_ncthw[:, :, u] + m[u] * frames[-1])
    return torch.stack(frames, dim=2)


def tv_norm(m, p=3, q=3):
    val = 0
    for u in range(1, m.shape[0] - 1):
        val = val + torch.abs(m[u - 1] - m[u]) ** p + torch.abs(m[u + 1] - m[u]) ** p
    return (val ** (1 / p)) ** q


def init_mask_central(x, weights, target, thresh=0.9, **kw):
    """mask.py:83-130 of the TF half: every score goes through the graph's sigmoid(mask_var)."""
    T = x.shape[2]

    def score(mv):
        with torch.no_grad():
            lg, _ = model(freeze(x, torch.sigmoid(mv)), weights, **kw)
            return float(torch.softmax(lg, 1)[0, target])
    full, orig = score(torch.ones(T)), score(torch.zeros(T))
    new = None
    for i in range(1, T // 2):
        new = torch.ones(T)
        new[:i] = 0
        new[-i:] = 0
        if (orig - score(new)) / (orig - full) < thresh:
            break
    return torch.where(new == 0, torch.tensor(-5.0), torch.tensor(5.0))


def search(x, weights, target, init, lam1, lam2, N, lr=0.2, b1=0.9, b2=0.999, eps=1e-8, **kw):
    """find_mask_kth.py:356-372, 431-452 with tf.train.AdamOptimizer's update rule."""
    mv = init.clone().requires_grad_()
    m1 = torch.zeros_like(mv)
    v1 = torch.zeros_like(mv)
    traj = []
    for n in range(1, N + 1):
        mc = torch.sigmoid(mv)
        l1 = lam1 * torch.sum(torch.abs(mc))
        tv = lam2 * tv_norm(mc)
        lg, _ = model(freeze(x, mc), weights, **kw)
        cl = torch.softmax(lg, 1)[0, target]
        loss = l1 + tv + cl
        g, = torch.autograd.grad(loss, mv)
        with torch.no_grad():
            m1 = b1 * m1 + (1 - b1) * g
            v1 = b2 * v1 + (1 - b2) * g * g
            lr_t = lr * np.sqrt(1 - b2 ** n) / (1 - b1 ** n)
            mv -= lr_t * m1 / (torch.sqrt(v1) + eps)
        traj.append([float(loss), float(l1), float(tv), float(cl)])
    return np.array(traj), torch.sigmoid(mv.detach())


def resize_bilinear(img, height, width):
    """skimage.transform.resize(order=1) restated as half-pixel bilinear with clamped coordinates (unpinned)."""
    t = torch.from_numpy(np.asarray(img, dtype=np.float32))[None, None]
    sh, sw = t.shape[-2:]
    ys = torch.clamp((torch.arange(height, dtype=torch.float32) + 0.5) * (sh / height) - 0.5, 0, sh - 1)
    xs = torch.clamp((torch.arange(width, dtype=torch.float32) + 0.5) * (sw / width) - 0.5, 0, sw - 1)
    y0, x0 = ys.floor().long(), xs.floor().long()
    y1, x1 = torch.clamp(y0 + 1, max=sh - 1), torch.clamp(x0 + 1, max=sw - 1)
    wy, wx = (ys - y0.float())[:, None], (xs - x0.float())[None, :]
    a = t[0, 0]
    top = a[y0][:, x0] * (1 - wx) + a[y0][:, x1] * wx
    bot = a[y1][:, x0] * (1 - wx) + a[y1][:, x1] * wx
    return (top * (1 - wy) + bot * wy).numpy()


def gradcam_frames(x, weights, target, mask=None, per_frame=True, out_hw=None, **kw):
    """gradcam.py:28-111: gradient of the class LOGIT w.r.t. the last ConvLSTM2D's output sequence as the layers
    above it see it (tf.gradients w.r.t. the layer's output tensor), per-frame cams, 'frame' | 'sequence' max."""
    xin = freeze(x, mask) if mask is not None else x
    xs = xin.permute(0, 2, 1, 3, 4)
    with torch.no_grad():
        for (k, rk, b) in weights['layers'][:-1]:
            xs, _ = clstm_block(xs, k, rk, b, kw.get('stride', 2), kw.get('padding', 'valid'), kw.get('hard', True))
        k, rk, b = weights['layers'][-1]
        _, out = clstm_block(xs, k, rk, b, kw.get('stride', 2), kw.get('padding', 'valid'), kw.get('hard', True))
    out = out.detach().requires_grad_()                  # the layer OUTPUT tensor: upstream of pool + dense only
    B, T, Fu, Ho, Wo = out.shape
    pooled = F.max_pool2d(out.reshape(B * T, Fu, Ho, Wo), 2).reshape(B, T, Fu, Ho // 2, Wo // 2).permute(0, 1, 3, 4, 2)
    flat = pooled[:, -1].reshape(B, -1) if kw.get('only_last', True) else pooled.reshape(B, -1)
    logits = flat @ weights['dense_w'] + weights['dense_b']
    grad, = torch.autograd.grad(logits[0, target], out)
    cams = []
    for t in range(T):
        w = grad[0, t].mean(dim=(1, 2))
        cam = torch.zeros(Ho, Wo)
        for j in range(Fu):
            cam = cam + w[j] * out[0, t, j].detach()
        cams.append(torch.clamp(cam, min=0).numpy())
    seq_max = max(float(c.max()) for c in cams)
    H, W = out_hw if out_hw is not None else x.shape[-2:]
    res = []
    with np.errstate(invalid='ignore', divide='ignore'):
        for c in cams:
            res.append(resize_bilinear(c, H, W) / np.float32(c.max() if per_frame else seq_max))
    return np.stack(res), torch.softmax(logits, 1).detach()
