"""Dev tool (GPU box): what ARE the module-wise backward outliers?  Mixed_3b on the K32 geometry, per arithmetic mode:
the CPU module (plain torch.nn.functional, fp32) runs on the GPU's own input activation and upstream gradient; an inner
ReLU whose pre-activation sits within rounding of zero may come out on the other side on the CPU, and ONE such flip at
(position, channel) of b1a / b2a changes the module's input gradient at that position for all 192 input channels.
Counts: inner gate disagreements, outlier elements (> thr * max), and outliers left at positions without a flip."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "interpreting-video-features_amd"))
os.environ["IVF_AUTOTUNE"] = "0"
import numpy as np
import torch
import torch.nn.functional as F
import ivf_engine
import ivf_recipe as R

sdn = R.i3d_state_dict(num_classes=6, tag='i3d_kth')
sd = R.to_torch(sdn)


def unit(x, name, k):
    w = sd[name + '.conv3d.weight']
    p = (k - 1) // 2
    x = F.conv3d(F.pad(x, (p, k - 1 - p) * 3), w)
    x = F.batch_norm(x, sd[name + '.bn.running_mean'], sd[name + '.bn.running_var'], sd[name + '.bn.weight'],
                     sd[name + '.bn.bias'], training=False, eps=1e-3)
    return x


x = torch.from_numpy(R.clip(9, 3, 32, 120, 160))[None].cuda()
for m in ("fp32", "bf16x6", "bf16x3"):
    eng = ivf_engine.I3DEngine(6, (3, 32, 120, 160), max_batch=1, head_hw=(4, 5), head_time_base=4, softmax=True, math=m)
    eng.load_state_dict(sdn)
    p = eng.forward(x)
    eng.backward(1, target=[int(torch.argmax(p[0]))], want_dx=False)
    v = eng.endpoint("MaxPool3d_3a_3x3", 1).cpu().requires_grad_()
    g = eng.endpoint("Mixed_3b:grad", 1).cpu()
    got = eng.endpoint("MaxPool3d_3a_3x3:grad", 1).cpu().numpy()
    inner_gpu = eng.endpoint("Mixed_3b.b12a", 1).cpu()
    n = "Mixed_3b"
    p1, p2 = unit(v, n + '.b1a', 1), unit(v, n + '.b2a', 1)
    b0 = F.relu(unit(v, n + '.b0', 1))
    b1 = F.relu(unit(F.relu(p1), n + '.b1b', 3))
    b2 = F.relu(unit(F.relu(p2), n + '.b2b', 3))
    b3 = F.relu(unit(F.max_pool3d(F.pad(v, (1, 1) * 3), 3, 1), n + '.b3b', 1))
    y = torch.cat([b0, b1, b2, b3], 1)
    (y * g).sum().backward()
    ref = v.grad.numpy()
    pre = torch.cat([p1, p2], 1).detach()
    flips = ((pre > 0) != (inner_gpu > 0))
    nf = int(flips.sum())
    pos_flip = flips.any(dim=1, keepdim=True).numpy()          # positions with at least one flipped gate
    # a flipped b1a/b2a gate reaches the input through the 3x3x3 convs' own backward only at that position (1x1x1 unit)
    sc = np.abs(ref).max()
    for thr in (1e-4, 1e-3):
        bad = np.abs(got - ref) > thr * sc
        print(f"{m}: thr {thr:g}: outliers {int(bad.sum())} ({bad.mean():.5f}); inner gate flips {nf} of {flips.numel()} at "
              f"{int(pos_flip.sum())} positions (|pre-activation| at the flips <= {float(pre[flips].abs().max()) if nf else 0:.2e}); "
              f"outliers at positions WITHOUT a flip: {int((bad & ~pos_flip).sum())}", flush=True)
