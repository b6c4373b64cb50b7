"""Dev tool (GPU box): accuracy of one convolution per arithmetic mode against torch fp64 (RMS and max error relative
to the RMS of the result), default kernel choice and every variant's RMS error for the 6-pass mode."""
import ctypes
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "interpreting-video-features_amd"))
import numpy as np
import torch
import torch.nn.functional as F
import ivf_lib as L

lib = L.lib()
gen = torch.Generator().manual_seed(1)
for k, cin, cout, thw in ((1, 192, 112, (8, 15, 20)), (3, 64, 192, (4, 14, 14)), (3, 96, 128, (8, 15, 20))):
    B = 2
    x = torch.relu(torch.randn((B, cin) + thw, generator=gen)) * 3
    w = torch.randn(cout, cin, k, k, k, generator=gen) * 0.05
    pf, pb = (k - 1) // 2, k - 1 - (k - 1) // 2
    ref = F.conv3d(F.pad(x.double(), (pf, pb, pf, pb, pf, pb)), w.double()).permute(0, 2, 3, 4, 1)
    rms = float(ref.pow(2).mean().sqrt())
    xcl = x.cuda().permute(0, 2, 3, 4, 1).contiguous()
    wd = w.cuda()
    for math in ("fp32", "bf16x6", "bf16x3"):
        mm = L.MATH_MODES[math]
        wf = torch.empty(lib.ivf_conv3d_pack_fwd_elems(cout, cin, k, k, k, mm), device='cuda')
        L.check(lib.ivf_conv3d_pack_fwd(L.ptr(wd), L.ptr(wf), cout, cin, cin, k, k, k, mm, L.stream()))
        d = L.ConvDesc()
        d.B, d.Ti, d.Hi, d.Wi = B, *thw
        d.Cin, d.in_ld, d.in_coff = cin, cin, 0
        d.To, d.Ho, d.Wo = thw
        d.Cout, d.out_ld, d.out_coff = cout, cout, 0
        d.kT = d.kH = d.kW = k
        d.sT = d.sH = d.sW = 1
        d.pT = d.pH = d.pW = pf
        d.math = mm
        ids = (ctypes.c_int * 96)()
        n = lib.ivf_conv3d_variants(ctypes.byref(d), ids, 96)
        out = []
        for v in [0] + list(ids)[:n]:
            d.variant = v
            y = torch.full((B,) + thw + (cout,), float('nan'), device='cuda')
            if lib.ivf_conv3d(ctypes.byref(d), L.ptr(xcl), L.ptr(wf), None, None, None, L.ptr(y), L.stream()) != 0:
                continue
            e = (y.double().cpu() - ref)
            out.append((v, float(e.pow(2).mean().sqrt()) / rms, float(e.abs().max()) / rms))
        print(f"k={k} cin={cin} cout={cout} {math}: default rms {out[0][1]:.2e} max {out[0][2]:.2e}; variants rms "
              f"min {min(o[1] for o in out[1:]):.2e} max {max(o[1] for o in out[1:]):.2e} (worst id {max(out[1:], key=lambda o: o[1])[0]})")
