"""Deterministic, framework-independent weights and clips.

The reference ships no trained checkpoint (configs/config_i3d_smth.py:50
`'pretrained_model_path': 'no_ckpt'`) and an I3D state_dict is 49.9 MB, too big
to commit as a fixture.  So parity tests, the benchmark and the golden-vector
generator all rebuild the same tensors from a counter-based hash of
(key, flat index): the generator loads them INTO the reference model in the
build container, the GPU box regenerates them from this file alone.

Key scheme follows the reference state_dict:
`Conv3d_1a_7x7.conv3d.weight`, `Mixed_3b.b1a.bn.running_var`,
`logits.conv3d.bias`, `clstm.cell0.Wxi.weight`, `endFC.weight`, ...
"""
import zlib

import numpy as np

import ivf_arch as arch

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x):
    with np.errstate(over='ignore'):
        x = x + np.uint64(0x9E3779B97F4A7C15)
        z = x
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def uniform(key, shape, lo=-1.0, hi=1.0):
    """float32 array, element i = lo + (hi-lo) * u24(hash(crc32(key), i)) / 2^24."""
    n = int(np.prod(shape)) if len(shape) else 1
    seed = np.uint64(zlib.crc32(key.encode('utf-8'))) << np.uint64(32)
    idx = np.arange(n, dtype=np.uint64)
    with np.errstate(over='ignore'):
        z = _splitmix64(_splitmix64(seed) ^ idx)
    u = (z >> np.uint64(40)).astype(np.float64) / float(1 << 24)
    return (lo + (hi - lo) * u).astype(np.float32).reshape(shape)


# Scales chosen once (see DESIGN.md "weight recipe") so that with raw 0..255
# clips the activations stay O(1..10) through the net and the softmax is
# peaked but not saturated, which keeps d(score)/d(mask) comparable to the
# regulariser gradient (SURVEY.md §7 hard part (c)).
STEM_GAIN = 1.0 / 96.0
LOGIT_GAIN = 3.0


def i3d_state_dict(num_classes=174, in_channels=3, stride_mod_layers="",
                   last_stride=1, tag="i3d", logit_gain=LOGIT_GAIN):
    """dict key -> np.float32 array in the reference layout
    ([Cout,Cin,kT,kH,kW] conv weights, I3D_doubled.py:64-71)."""
    sd = {}
    for (name, cin, cout, k, s, has_bn) in arch.conv_units(
            in_channels, num_classes, stride_mod_layers, last_stride):
        fan_in = cin * k[0] * k[1] * k[2]
        bound = float(np.sqrt(6.0 / fan_in))  # He-uniform
        if name == 'Conv3d_1a_7x7':
            bound *= STEM_GAIN
        if name == 'logits':
            bound = float(logit_gain / np.sqrt(fan_in))
        w = uniform(f"{tag}/{name}.conv3d.weight", (cout, cin) + tuple(k), -bound, bound)
        sd[f"{name}.conv3d.weight"] = w
        if has_bn:
            sd[f"{name}.bn.weight"] = uniform(f"{tag}/{name}.bn.weight", (cout,), 0.8, 1.2)
            sd[f"{name}.bn.bias"] = uniform(f"{tag}/{name}.bn.bias", (cout,), -0.1, 0.1)
            sd[f"{name}.bn.running_mean"] = uniform(f"{tag}/{name}.bn.running_mean", (cout,), -0.1, 0.1)
            sd[f"{name}.bn.running_var"] = uniform(f"{tag}/{name}.bn.running_var", (cout,), 0.7, 1.3)
            sd[f"{name}.bn.num_batches_tracked"] = np.zeros((), dtype=np.int64)
        else:
            sd[f"{name}.conv3d.bias"] = uniform(f"{tag}/{name}.conv3d.bias", (cout,), -0.5, 0.5)
    return sd


def clstm_state_dict(num_classes=6, hidden=4, channels=1, kernel=5, layers=2,
                     image_size=(160, 120), conv_stride=2, pool=2, tag="clstm",
                     fc_gain=4.0, fc_mult=1):
    """Reference key scheme of models/CLSTM_4.py + convolution_lstm.py:22-29, 85."""
    sd = {}
    sd["clstm.bn.weight"] = uniform(f"{tag}/bn.weight", (hidden,), 0.8, 1.2)
    sd["clstm.bn.bias"] = uniform(f"{tag}/bn.bias", (hidden,), -0.1, 0.1)
    sd["clstm.bn.running_mean"] = uniform(f"{tag}/bn.running_mean", (hidden,), -0.05, 0.05)
    sd["clstm.bn.running_var"] = uniform(f"{tag}/bn.running_var", (hidden,), 0.05, 0.15)
    sd["clstm.bn.num_batches_tracked"] = np.zeros((), dtype=np.int64)
    cin = channels
    for i in range(layers):
        for g in "ifco":
            bx = float(np.sqrt(3.0 / (cin * kernel * kernel)))
            bh = float(np.sqrt(3.0 / (hidden * kernel * kernel)))
            sd[f"clstm.cell{i}.Wx{g}.weight"] = uniform(
                f"{tag}/cell{i}.Wx{g}.weight", (hidden, cin, kernel, kernel), -bx, bx)
            sd[f"clstm.cell{i}.Wx{g}.bias"] = uniform(f"{tag}/cell{i}.Wx{g}.bias", (hidden,), -0.2, 0.2)
            sd[f"clstm.cell{i}.Wh{g}.weight"] = uniform(
                f"{tag}/cell{i}.Wh{g}.weight", (hidden, hidden, kernel, kernel), -bh, bh)
        cin = hidden
    red = (conv_stride * pool) ** layers
    feat = fc_mult * hidden * int(image_size[0] / red) * int(image_size[1] / red)   # fc_mult: use_entire_seq
    bf = float(fc_gain / np.sqrt(feat))
    sd["endFC.weight"] = uniform(f"{tag}/endFC.weight", (num_classes, feat), -bf, bf)
    sd["endFC.bias"] = uniform(f"{tag}/endFC.bias", (num_classes,), -0.1, 0.1)
    return sd


def clip(clip_id, channels=3, frames=16, height=224, width=224, tag="clip"):
    """Synthetic decoded-JPEG-like clip: integers 0..255 as float32, NCTHW without
    the batch dim (data_loader_jpg.py:29-37 feeds raw un-normalised 0..255).
    A smooth per-frame drift is added so that frames differ coherently and the
    freeze perturbation is non-trivial."""
    base = uniform(f"{tag}/{clip_id}/base", (channels, 1, height, width), 0.0, 1.0)
    noise = uniform(f"{tag}/{clip_id}/noise", (channels, frames, height, width), 0.0, 1.0)
    t = np.arange(frames, dtype=np.float32).reshape(1, frames, 1, 1) / max(frames - 1, 1)
    yy = np.linspace(0, 1, height, dtype=np.float32).reshape(1, 1, height, 1)
    xx = np.linspace(0, 1, width, dtype=np.float32).reshape(1, 1, 1, width)
    phase = uniform(f"{tag}/{clip_id}/phase", (4,), 0.0, 1.0)
    wave = 0.5 + 0.5 * np.sin(2 * np.pi * (xx * (1 + 2 * phase[0]) + yy * (1 + 2 * phase[1])
                                           + t * (0.5 + phase[2])) + 6.28 * phase[3])
    img = 0.45 * base + 0.2 * noise + 0.35 * wave
    return np.rint(np.clip(img, 0, 1) * 255.0).astype(np.float32)


def label(clip_id, num_classes):
    return int(clip_id % num_classes)


def to_torch(sd):
    import torch
    return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()}
