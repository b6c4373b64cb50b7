#!/usr/bin/env python3
"""Dev tool: the stem forward (7x7x7 stride-2 conv on 4-channel pixels) alone: stem_fwd_micro.py [B] [math] [variants]."""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "interpreting-video-features_amd"))
import torch, ivf_lib as L
if os.environ.get("IVF_DIAG_LIB"): L.LIB_PATH = os.path.join(ROOT, "interpreting-video-features_amd", os.environ["IVF_DIAG_LIB"])
lib = L.lib(); B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
mm = L.MATH_MODES[sys.argv[2] if len(sys.argv) > 2 else "bf16x6"]
variants = [int(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else [2, 15]
cout, cin, cinp, k = 64, 3, 4, (7, 7, 7)
w = torch.randn(cout, cin, *k, device='cuda') * 0.05
wf = torch.empty(lib.ivf_conv3d_pack_fwd_elems(cout, cinp, *k, mm), device='cuda')
L.check(lib.ivf_conv3d_pack_fwd(L.ptr(w), L.ptr(wf), cout, cin, cinp, *k, mm, L.stream()))
x = torch.zeros(B, 16, 224, 224, cinp, device='cuda'); x[..., :3] = torch.rand(B, 16, 224, 224, 3, device='cuda') * 2 - 1
y = torch.empty(B, 8, 112, 112, cout, device='cuda')
sc, sh = torch.ones(cout, device='cuda'), torch.zeros(cout, device='cuda')
d = L.ConvDesc(); d.B, d.Ti, d.Hi, d.Wi = B, 16, 224, 224; d.Cin, d.in_ld, d.in_coff = cinp, cinp, 0
d.To, d.Ho, d.Wo = 8, 112, 112; d.Cout, d.out_ld, d.out_coff = cout, cout, 0
d.kT, d.kH, d.kW = k; d.sT = d.sH = d.sW = 2; d.pT, d.pH, d.pW = 2, 2, 2
d.relu, d.math = 1, mm
ref = None
for v in variants:
    d.variant = v
    for _ in range(2): L.check(lib.ivf_conv3d(ctypes.byref(d), L.ptr(x), L.ptr(wf), L.ptr(sc), L.ptr(sh), None, L.ptr(y), L.stream()))
    torch.cuda.synchronize(); t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(5): L.check(lib.ivf_conv3d(ctypes.byref(d), L.ptr(x), L.ptr(wf), L.ptr(sc), L.ptr(sh), None, L.ptr(y), L.stream()))
    t1.record(); torch.cuda.synchronize()
    if ref is None: ref = y.clone()
    ms = t0.elapsed_time(t1) / 5
    print(f"variant {v}: {ms:.3f} ms  {2 * B * 8 * 112 * 112 * cout * 1029 / ms / 1e9:.1f} TFLOP/s  max|d vs first| {float((y - ref).abs().max()):.2e}")
