#!/usr/bin/env python3
"""Dev tool: the stem backward-data (4x4x4 depth-to-space conv) alone, chosen variants, for PMC passes."""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "interpreting-video-features_amd"))
import torch, ivf_lib as L
lib = L.lib(); B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
variants = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [33, 48]
mm = L.MATH_MODES["bf16x3"]; cout, cin, cinp, k, s, pads, thw = 64, 3, 4, (7, 7, 7), (2, 2, 2), (2, 2, 2), (16, 224, 224)
w = torch.randn(cout, cin, *k, device='cuda') * 0.05
wb = torch.empty(lib.ivf_conv3d_pack_bwd_elems(cout, cinp, *k, *s, *pads, mm), device='cuda')
geom = L.BwdGeom()
L.check(lib.ivf_conv3d_pack_bwd(L.ptr(w), None, L.ptr(wb), cout, cin, cinp, *k, *s, *pads, mm, ctypes.byref(geom), L.stream()))
outs = (8, 112, 112)
g = torch.randn(B, *outs, cout, device='cuda'); dx = torch.empty(B, *thw, cinp, device='cuda')
e = L.ConvDesc(); e.B, e.Ti, e.Hi, e.Wi = B, *outs; e.Cin, e.in_ld, e.in_coff = cout, cout, 0
e.kT, e.kH, e.kW = geom.kT, geom.kH, geom.kW; e.sT = e.sH = e.sW = 1; e.pT, e.pH, e.pW = geom.pT, geom.pH, geom.pW
e.out_ld, e.out_coff, e.math, e.d2s = cinp, 0, mm, 1; e.bsT, e.bsH, e.bsW = s
e.To, e.Ho, e.Wo = 8, 112, 112; e.Cout = geom.rows; e.dT, e.dH, e.dW = thw; e.dC = cinp
ref = None
for v in variants:
    e.variant = v
    for _ in range(2): L.check(lib.ivf_conv3d(ctypes.byref(e), L.ptr(g), L.ptr(wb), None, None, None, L.ptr(dx), L.stream()))
    torch.cuda.synchronize(); t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(3): L.check(lib.ivf_conv3d(ctypes.byref(e), L.ptr(g), L.ptr(wb), None, None, None, L.ptr(dx), L.stream()))
    t1.record(); torch.cuda.synchronize()
    if ref is None: ref = dx.clone()
    print(f"variant {v}: {t0.elapsed_time(t1) / 3:.3f} ms  max|d vs first| {float((dx - ref).abs().max()):.2e}")
