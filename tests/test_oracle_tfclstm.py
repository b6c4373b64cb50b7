"""CPU: self-checks of the TF-style ConvLSTM restatement (oracle/tfclstm_ref.py; SURVEY 8f N4, parity unpinned):
the pieces with a closed form or an independent torch equivalent."""
import numpy as np
import torch
import torch.nn.functional as F

from oracle import tfclstm_ref as ref


def test_hard_sigmoid_and_same_padding():
    z = torch.tensor([-3.0, -2.5, -1.0, 0.0, 1.0, 2.5, 3.0])
    assert torch.allclose(ref.hard_sigmoid(z), torch.tensor([0.0, 0.0, 0.3, 0.5, 0.7, 1.0, 1.0]))
    g = torch.Generator().manual_seed(0)
    x, w = torch.randn(1, 2, 9, 11, generator=g), torch.randn(3, 5, 2, 4, generator=g)
    # odd kernel, stride 1: TensorFlow 'same' is the symmetric padding torch calls (1, 2)
    assert torch.allclose(ref.conv2d_tf(x, w, 1, 'same'), F.conv2d(x, w.permute(3, 2, 0, 1), padding=(1, 2)), atol=1e-6)
    # stride 2, even size: out = ceil(n / s), the odd pad cell at the BACK (TensorFlow's rule)
    y = ref.conv2d_tf(torch.randn(1, 2, 30, 40, generator=g), w, 2, 'same')
    assert tuple(y.shape[2:]) == (15, 20)
    assert ref._pad_same(30, 3, 2) == (0, 1) and ref._pad_same(40, 5, 2) == (1, 2)
    assert tuple(ref.conv2d_tf(torch.zeros(1, 2, 120, 160), w, 2, 'valid').shape[2:]) == (59, 78)


def test_freeze_recurrence_and_tf_adam_step():
    g = torch.Generator().manual_seed(1)
    x = torch.rand(1, 1, 5, 2, 2, generator=g)
    m = torch.tensor([0.3, 1.0, 0.0, 0.5, 0.25])
    p = ref.freeze(x, m)
    assert torch.equal(p[:, :, 0], x[:, :, 0]) and torch.equal(p[:, :, 1], x[:, :, 0]) and torch.equal(p[:, :, 2], x[:, :, 2])
    assert torch.allclose(p[:, :, 3], 0.5 * x[:, :, 3] + 0.5 * x[:, :, 2])
    # tf.train.AdamOptimizer, first step: m = (1-b1) g, v = (1-b2) g^2, lr_t = lr sqrt(1-b2)/(1-b1)
    gr, lr, b1, b2, eps = 0.3, 0.2, 0.9, 0.999, 1e-8
    step = lr * np.sqrt(1 - b2) / (1 - b1) * ((1 - b1) * gr) / (np.sqrt((1 - b2) * gr * gr) + eps)
    assert abs(step - lr * gr / (abs(gr) + eps / np.sqrt(1 - b2))) < 1e-12      # the form ivf_tfclstm_search uses


def test_model_shapes_and_gradcam_normalisation():
    g = torch.Generator().manual_seed(2)
    C, T, H, W, units, K = 1, 4, 20, 24, (3, 2), 4
    layers, cin = [], C
    for Fu in units:
        layers.append((torch.randn(3, 5, cin, 4 * Fu, generator=g) * 0.3, torch.randn(3, 5, Fu, 4 * Fu, generator=g) * 0.2,
                       torch.zeros(4 * Fu)))
        cin = Fu
    x = torch.rand(1, C, T, H, W, generator=g)
    # valid, stride 2: 20x24 -> 9x10 -> pool 4x5 -> 1x1?? second layer: (4-3)//2+1 = 1, (5-5)//2+1 = 1 -> too small to pool
    w1 = dict(layers=layers[:1], dense_w=torch.randn(4 * 5 * 3 * T, K, generator=g), dense_b=torch.zeros(K))
    lg, out = ref.model(x, w1, stride=2, padding='valid', hard=True, only_last=False)
    assert tuple(lg.shape) == (1, K) and tuple(out.shape) == (1, T, 3, 9, 10)
    cam_f, _ = ref.gradcam_frames(x, w1, 1, per_frame=True, stride=2, padding='valid', hard=True, only_last=False)
    cam_s, _ = ref.gradcam_frames(x, w1, 1, per_frame=False, stride=2, padding='valid', hard=True, only_last=False)
    assert cam_f.shape == (T, H, W)
    ok = ~np.isnan(cam_f).any(axis=(1, 2))
    assert np.nanmax(cam_s) <= 1.0 + 1e-6
    for t in np.where(ok)[0]:
        # per frame the SOURCE map peaks at 1 (gradcam.py:17: cam / cam_max before the resize); the bilinear
        # samples of it stay at or below that peak
        assert 0.5 < np.max(cam_f[t]) <= 1.0 + 1e-6
