#!/usr/bin/env python3
"""Dev tool: where a workgroup of the LDS-halo conv kernel spends its cycles (diagnostic build
`make -C interpreting-video-features_amd/csrc stamps`).  Runs single layers of the I3D plan at batch B with a
chosen variant and prints staging / tap-loop / epilogue shares (thread 0's s_memtime per workgroup)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "interpreting-video-features_amd"))
import ivf_lib as L
L.LIB_PATH = os.path.join(ROOT, "interpreting-video-features_amd", os.environ.get("IVF_DIAG_LIB", "libivf_hip_stamps.so"))
import torch
lib = L.lib()
HAVE_STAMPS = hasattr(lib, "ivf_debug_halo_stamps")   # a product build only gets timed
if HAVE_STAMPS:
    lib.ivf_debug_halo_stamps.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
MATH = os.environ.get("IVF_STAMPS_MATH", "bf16x3")   # build with STAMP_TU=conv3d_halo_x6 for bf16x6
mm = L.MATH_MODES[MATH]


def run(name, cin, cout, k, thw, variants, bwd_d2s=False):
    dev = 'cuda'
    if bwd_d2s:      # the stem backward-data
        w = torch.randn(64, 3, 7, 7, 7, device=dev) * 0.05
        wb = torch.empty(lib.ivf_conv3d_pack_bwd_elems(64, 4, 7, 7, 7, 2, 2, 2, 2, 2, 2, mm), device=dev)
        geom = L.BwdGeom()
        L.check(lib.ivf_conv3d_pack_bwd(L.ptr(w), None, L.ptr(wb), 64, 3, 4, 7, 7, 7, 2, 2, 2, 2, 2, 2, mm,
                                        ctypes.byref(geom), L.stream()))
        x = torch.randn(B, 8, 112, 112, 64, device=dev)
        if os.environ.get('IVF_STAMPS_ZERO'): x.zero_()
        y = torch.empty(B, 16, 224, 224, 4, device=dev)
        d = L.ConvDesc()
        d.B, d.Ti, d.Hi, d.Wi = B, 8, 112, 112
        d.Cin, d.in_ld, d.in_coff = 64, 64, 0
        d.kT, d.kH, d.kW = geom.kT, geom.kH, geom.kW
        d.sT = d.sH = d.sW = 1
        d.pT, d.pH, d.pW = geom.pT, geom.pH, geom.pW
        d.out_ld, d.out_coff, d.math, d.d2s = 4, 0, mm, 1
        d.bsT = d.bsH = d.bsW = 2
        d.To, d.Ho, d.Wo = 8, 112, 112
        d.Cout = geom.rows
        d.dT, d.dH, d.dW, d.dC = 16, 224, 224, 4
        wf, sc, sh = wb, None, None
    else:
        w = torch.randn(cout, cin, k, k, k, device=dev) * 0.05
        wf = torch.empty(lib.ivf_conv3d_pack_fwd_elems(cout, cin, k, k, k, mm), device=dev)
        L.check(lib.ivf_conv3d_pack_fwd(L.ptr(w), L.ptr(wf), cout, cin, cin, k, k, k, mm, L.stream()))
        x = torch.randn(B, *thw, cin, device=dev)
        if os.environ.get('IVF_STAMPS_ZERO'): x.zero_()   # zero operands: the clock the chip holds when the matrix pipe toggles nothing
        y = torch.empty(B, *thw, cout, device=dev)
        sc, sh = torch.ones(cout, device=dev), torch.zeros(cout, device=dev)
        d = L.ConvDesc()
        d.B, d.Ti, d.Hi, d.Wi = B, *thw
        d.Cin, d.in_ld, d.in_coff = cin, cin, 0
        d.To, d.Ho, d.Wo = thw
        d.Cout, d.out_ld, d.out_coff = cout, cout, 0
        d.kT = d.kH = d.kW = k
        d.sT = d.sH = d.sW = 1
        d.pT = d.pH = d.pW = (k - 1) // 2
        d.relu, d.math = 1, mm
    for v in variants:
        d.variant = v
        out = (ctypes.c_ulonglong * 8)()
        for _ in range(2):
            L.check(lib.ivf_conv3d(ctypes.byref(d), L.ptr(x), L.ptr(wf), L.ptr(sc), L.ptr(sh), None, L.ptr(y), L.stream()))
        torch.cuda.synchronize()
        if HAVE_STAMPS: lib.ivf_debug_halo_stamps(out, 1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        NREP = 1 if HAVE_STAMPS else 5
        e0.record()
        for _ in range(NREP):
            L.check(lib.ivf_conv3d(ctypes.byref(d), L.ptr(x), L.ptr(wf), L.ptr(sc), L.ptr(sh), None, L.ptr(y), L.stream()))
        e1.record()
        torch.cuda.synchronize()
        if not HAVE_STAMPS:
            print(f"{name:18s} v{v:2d} {e0.elapsed_time(e1) / NREP:7.3f} ms")
            continue
        lib.ivf_debug_halo_stamps(out, 1)
        tot, stg, taps, epi, n = out[0], out[1], out[2], out[3], max(out[4], 1)
        clk = tot / max(out[5], 1) * 0.1
        print(f"{name:18s} v{v:2d} {e0.elapsed_time(e1):7.3f} ms  {n} WGs  per WG {tot / n / 1e3:7.1f}k shader cycles at {clk:4.2f} GHz: "
              f"staging {stg / tot * 100:4.1f}%  tap loops {taps / tot * 100:4.1f}%  reduction+epilogue {epi / tot * 100:4.1f}%  "
              f"other {100 - (stg + taps + epi) / tot * 100:4.1f}%")


ONLY = os.environ.get("IVF_STAMPS_ONLY")
if MATH == "bf16x6":
  HB = 16   # IVF_CONV_HALO_BASE
  run("Conv3d_2c fwd", 64, 192, 3, (8, 56, 56), [HB + 60, HB + 68, HB + 61, HB + 69, HB + 20])
  run("Mixed_3c.b1b fwd", 128, 192, 3, (8, 28, 28), [HB + 60, HB + 68, HB + 61, HB + 69])
  run("Mixed_4f.b1b fwd", 160, 320, 3, (4, 14, 14), [HB + 60, HB + 68, HB + 50])
  run("stem bwd", 0, 0, 0, None, [HB + 37, HB + 36], bwd_d2s=True)
  sys.exit(0)
if not ONLY:
  run("Conv3d_2c fwd", 64, 192, 3, (8, 56, 56), [16, 39])
  run("Mixed_3c.b1b fwd", 128, 192, 3, (8, 28, 28), [16, 39, 40])
  run("Mixed_4f.b1b fwd", 160, 320, 3, (4, 14, 14), [16, 18])
  run("Mixed_3b.b2b fwd", 16, 32, 3, (8, 28, 28), [37, 62, 58, 59, 63])
run("stem bwd", 0, 0, 0, None, [33, 23, 55] if ONLY else [33, 23, 54, 55, 56, 57, 60], bwd_d2s=True)
