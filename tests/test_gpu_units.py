"""HIP conv / pool kernels through the C-ABI vs the reference's own outputs
(tests/golden/units.npz, produced by I3D_doubled.Unit3D / MaxPool3dSamePadding)."""
import ctypes

import numpy as np
import pytest
import torch

from conftest import rel_err

pytestmark = pytest.mark.gpu

UNIT_CASES = {  # name: (cin, cout, k, stride, in T,H,W) -- same table as make_golden.py
    'k1': (24, 40, (1, 1, 1), (1, 1, 1), (3, 5, 6)),
    'k3': (8, 12, (3, 3, 3), (1, 1, 1), (4, 7, 9)),
    'k3_c16': (16, 48, (3, 3, 3), (1, 1, 1), (2, 7, 7)),
    'k7s2_even': (3, 16, (7, 7, 7), (2, 2, 2), (8, 16, 18)),
    'k7s2_odd': (3, 16, (7, 7, 7), (2, 2, 2), (7, 15, 17)),
    'k7s1t': (3, 8, (7, 7, 7), (1, 2, 2), (6, 12, 12)),
}
POOL_CASES = {
    'p133': ((1, 3, 3), (1, 2, 2), (3, 8, 10)),
    'p133_odd': ((1, 3, 3), (1, 2, 2), (3, 7, 9)),
    'p333s2': ((3, 3, 3), (2, 2, 2), (4, 8, 8)),
    'p333s2_odd': ((3, 3, 3), (2, 2, 2), (5, 15, 7)),
    'p222': ((2, 2, 2), (2, 2, 2), (4, 6, 8)),
    'p222_odd': ((2, 2, 2), (2, 2, 2), (3, 7, 5)),
    'p333s1': ((3, 3, 3), (1, 1, 1), (3, 5, 6)),
    'p333s1t1': ((3, 3, 3), (1, 2, 2), (4, 8, 8)),
}


def to_cl(x, cpad=None):
    """NCTHW -> channels-last [B,T,H,W,Cpad] on the GPU."""
    B, C, T, H, W = x.shape
    cpad = cpad or C
    y = torch.zeros(B, T, H, W, cpad, device='cuda')
    y[..., :C] = x.permute(0, 2, 3, 4, 1)
    return y.contiguous()


def from_cl(y, C):
    return y[..., :C].permute(0, 4, 1, 2, 3).contiguous()


@pytest.mark.parametrize("math", ["fp32", "bf16x3", "bf16x6"])
@pytest.mark.parametrize("name", list(UNIT_CASES))
def test_unit3d_fwd_bwd(name, math, golden):
    import ivf_arch as arch
    import ivf_lib as L
    import ivf_recipe as R
    lib = L.lib()
    g = golden('units')
    mm = L.MATH_MODES[math]
    tol = 1e-4 if math == "bf16x3" else 1e-5    # split-bf16 x3: ~2^-17 per product; the 6-pass split is fp32-class
    cin, cout, k, s, thw = UNIT_CASES[name]
    cinp = (cin + 3) // 4 * 4
    B = 2
    dev = 'cuda'
    w = torch.from_numpy(R.uniform(f'g/unit/{name}/w', (cout, cin) + k, -0.2, 0.2)).to(dev)
    bn = [torch.from_numpy(R.uniform(f'g/unit/{name}/{t}', (cout,), lo, hi)).to(dev)
          for t, lo, hi in (('g', 0.5, 1.5), ('b', -0.3, 0.3), ('m', -0.3, 0.3), ('v', 0.5, 1.5))]
    x = torch.from_numpy(R.uniform(f'g/unit/{name}/x', (B, cin) + thw, -1, 1)).to(dev)
    pads = [arch.same_pad(n, kk, ss)[0] for n, kk, ss in zip(thw, k, s)]
    outs = [arch.out_size(n, kk, ss) for n, kk, ss in zip(thw, k, s)]
    gy = torch.from_numpy(R.uniform(f'g/unit/{name}/gy', (B, cout) + tuple(outs), -1, 1)).to(dev)

    scale = torch.empty(cout, device=dev)
    shift = torch.empty(cout, device=dev)
    L.check(lib.ivf_bn_fold(L.ptr(bn[0]), L.ptr(bn[1]), L.ptr(bn[2]), L.ptr(bn[3]), 1e-3, L.ptr(scale),
                            L.ptr(shift), cout, L.stream()))
    taps = k[0] * k[1] * k[2]
    wf = torch.empty(lib.ivf_conv3d_pack_fwd_elems(cout, cinp, *k, mm), device=dev)
    L.check(lib.ivf_conv3d_pack_fwd(L.ptr(w), L.ptr(wf), cout, cin, cinp, *k, mm, L.stream()))
    xcl = to_cl(x, cinp)
    ycl = torch.full((B,) + tuple(outs) + (cout,), float('nan'), device=dev)
    d = L.ConvDesc()
    d.B, d.Ti, d.Hi, d.Wi = B, *thw
    d.Cin, d.in_ld, d.in_coff = cinp, cinp, 0
    d.To, d.Ho, d.Wo = outs
    d.Cout, d.out_ld, d.out_coff = cout, cout, 0
    d.kT, d.kH, d.kW = k
    d.sT, d.sH, d.sW = s
    d.pT, d.pH, d.pW = pads
    d.relu = 1
    d.math = mm
    L.check(lib.ivf_conv3d(ctypes.byref(d), L.ptr(xcl), L.ptr(wf), L.ptr(scale), L.ptr(shift), None,
                           L.ptr(ycl), L.stream()))
    y = from_cl(ycl, cout).cpu().numpy()
    assert rel_err(y, g[f'unit_{name}_y']) < tol

    # backward-data: gradient gated by the unit's own ReLU, BN scale folded in the pack
    gate = (ycl > 0).float()
    gcl = (to_cl(gy) * gate).contiguous()
    n_el = lib.ivf_conv3d_pack_bwd_elems(cout, cinp, *k, *s, *pads, mm)
    wb = torch.empty(n_el, device=dev)
    geom = L.BwdGeom()
    L.check(lib.ivf_conv3d_pack_bwd(L.ptr(w), L.ptr(scale), L.ptr(wb), cout, cin, cinp, *k, *s, *pads, mm,
                                    ctypes.byref(geom), L.stream()))
    dxcl = torch.full((B,) + thw + (cinp,), float('nan'), device=dev)
    e = L.ConvDesc()
    e.B, e.Ti, e.Hi, e.Wi = B, *outs
    e.Cin, e.in_ld, e.in_coff = cout, cout, 0
    e.kT, e.kH, e.kW = geom.kT, geom.kH, geom.kW
    e.sT = e.sH = e.sW = 1
    e.pT, e.pH, e.pW = geom.pT, geom.pH, geom.pW
    e.out_ld, e.out_coff = cinp, 0
    e.math = mm
    if geom.d2s:
        e.d2s = 1
        e.bsT, e.bsH, e.bsW = s
        e.To, e.Ho, e.Wo = [-(-n // ss) for n, ss in zip(thw, s)]
        e.Cout = geom.rows
        e.dT, e.dH, e.dW = thw
        e.dC = cinp
    else:
        e.To, e.Ho, e.Wo = thw
        e.Cout = cinp
    L.check(lib.ivf_conv3d(ctypes.byref(e), L.ptr(gcl), L.ptr(wb), None, None, None, L.ptr(dxcl), L.stream()))
    dx = from_cl(dxcl, cin).cpu().numpy()
    assert np.isfinite(dxcl.cpu().numpy()).all()
    assert rel_err(dx, g[f'unit_{name}_dx']) < tol


@pytest.mark.parametrize("name", list(POOL_CASES))
def test_maxpool_fwd_bwd(name, golden):
    import ivf_arch as arch
    import ivf_lib as L
    import ivf_recipe as R
    lib = L.lib()
    g = golden('units')
    k, s, thw = POOL_CASES[name]
    B, C = 2, 6
    xv = np.maximum(R.uniform(f'g/pool/{name}/x', (B, C) + thw, -1, 1), 0)
    x = torch.from_numpy(xv).cuda()
    pads = [arch.same_pad(n, kk, ss)[0] for n, kk, ss in zip(thw, k, s)]
    outs = [arch.out_size(n, kk, ss) for n, kk, ss in zip(thw, k, s)]
    gy = torch.from_numpy(R.uniform(f'g/pool/{name}/gy', (B, C) + tuple(outs), -1, 1)).cuda()
    cp = 8
    xcl = to_cl(x, cp)
    ycl = torch.zeros((B,) + tuple(outs) + (cp,), device='cuda')
    idx = torch.zeros((B,) + tuple(outs) + (cp,), dtype=torch.uint8, device='cuda')
    d = L.PoolDesc()
    d.B, d.Ti, d.Hi, d.Wi, d.C, d.in_ld, d.in_coff = B, *thw, cp, cp, 0
    d.To, d.Ho, d.Wo, d.out_ld, d.out_coff = *outs, cp, 0
    d.kT, d.kH, d.kW = k
    d.sT, d.sH, d.sW = s
    d.pT, d.pH, d.pW = pads
    L.check(lib.ivf_maxpool3d_fwd(ctypes.byref(d), L.ptr(xcl), L.ptr(ycl), L.ptr(idx), L.stream()))
    assert np.array_equal(from_cl(ycl, C).cpu().numpy(), g[f'pool_{name}_y'])   # bit-exact
    dxcl = torch.full_like(xcl, float('nan'))
    gycl = to_cl(gy, cp)
    L.check(lib.ivf_maxpool3d_bwd(ctypes.byref(d), L.ptr(gycl), L.ptr(idx), L.ptr(dxcl), None, 0,
                                  L.stream()))
    dx = from_cl(dxcl, C).cpu().numpy()
    ref = g[f'pool_{name}_dx']
    # same winners (ties incl. zero padding) -> identical routing; sums of <= 27 terms
    assert np.array_equal(dx != 0, ref != 0)
    assert np.allclose(dx, ref, rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("thw,C", [((4, 16, 9), 40), ((8, 28, 28), 32), ((2, 7, 7), 136), ((1, 3, 2), 8)])
def test_maxpool_333_stride1_many_tiles_with_ties(thw, C):
    """The separable 3x3x3 stride-1 kernels (several row tiles, partial channel slabs) against
    torch's CPU max_pool3d over the zero-padded input (models/I3D_doubled.py:8-40), on inputs
    quantised so that windows hold many exact ties: forward bit-exact, backward routing
    identical."""
    import torch.nn.functional as F
    import ivf_lib as L
    lib = L.lib()
    gen = torch.Generator().manual_seed(7)
    B = 2
    x = torch.relu(torch.round(torch.randn((B, C) + thw, generator=gen) * 3) / 3)
    gy = torch.randn((B, C) + thw, generator=gen)
    xr = x.clone().requires_grad_()
    y = F.max_pool3d(F.pad(xr, (1, 1, 1, 1, 1, 1)), 3, 1)
    y.backward(gy)
    xcl = to_cl(x.cuda(), C)
    ycl = torch.zeros_like(xcl)
    idx = torch.zeros(xcl.shape, dtype=torch.uint8, device='cuda')
    d = L.PoolDesc()
    d.B, d.Ti, d.Hi, d.Wi, d.C, d.in_ld, d.in_coff = B, *thw, C, C, 0
    d.To, d.Ho, d.Wo, d.out_ld, d.out_coff = *thw, C, 0
    d.kT = d.kH = d.kW = 3
    d.sT = d.sH = d.sW = 1
    d.pT = d.pH = d.pW = 1
    L.check(lib.ivf_maxpool3d_fwd(ctypes.byref(d), L.ptr(xcl), L.ptr(ycl), L.ptr(idx), L.stream()))
    assert np.array_equal(from_cl(ycl, C).cpu().numpy(), y.detach().numpy())
    dxcl = torch.full_like(xcl, float('nan'))
    gycl = to_cl(gy.cuda(), C)
    L.check(lib.ivf_maxpool3d_bwd(ctypes.byref(d), L.ptr(gycl), L.ptr(idx), L.ptr(dxcl), None, 0,
                                  L.stream()))
    dx = from_cl(dxcl, C).cpu().numpy()
    ref = xr.grad.numpy()
    assert np.array_equal(dx != 0, ref != 0)
    assert np.allclose(dx, ref, rtol=1e-5, atol=1e-6)
    # accumulate + ReLU gate path
    base = torch.randn_like(xcl)
    acc = base.clone()
    L.check(lib.ivf_maxpool3d_bwd(ctypes.byref(d), L.ptr(gycl), L.ptr(idx), L.ptr(acc), L.ptr(xcl), 1,
                                  L.stream()))
    want = torch.where(xcl > 0, base + dxcl, torch.zeros_like(base))
    assert torch.allclose(acc, want, rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("k,st,thw", [((1, 3, 3), (1, 2, 2), (3, 9, 10)), ((3, 3, 3), (1, 1, 1), (4, 9, 8)),
                                      ((3, 3, 3), (2, 2, 2), (5, 8, 7)), ((2, 2, 2), (2, 2, 2), (4, 6, 6))])
def test_maxpool_gate_nonpos_equals_relu_mask(k, st, thw):
    """desc.gate_nonpos (dead windows recorded in the forward) gives the same input gradient as
    the explicit ReLU gate on a post-ReLU input with many zeros."""
    import ivf_arch as arch
    import ivf_lib as L
    lib = L.lib()
    gen = torch.Generator().manual_seed(11)
    B, C = 2, 12
    x = torch.relu(torch.randn((B, C) + thw, generator=gen) - 0.8)   # mostly zeros
    pads = [arch.same_pad(n, kk, ss)[0] for n, kk, ss in zip(thw, k, st)]
    outs = [arch.out_size(n, kk, ss) for n, kk, ss in zip(thw, k, st)]
    gy = to_cl(torch.randn((B, C) + tuple(outs), generator=gen).cuda(), C)
    xcl = to_cl(x.cuda(), C)
    res = []
    for flag in (0, 1):
        d = L.PoolDesc()
        d.B, d.Ti, d.Hi, d.Wi, d.C, d.in_ld, d.in_coff = B, *thw, C, C, 0
        d.To, d.Ho, d.Wo, d.out_ld, d.out_coff = *outs, C, 0
        d.kT, d.kH, d.kW = k
        d.sT, d.sH, d.sW = st
        d.pT, d.pH, d.pW = pads
        d.gate_nonpos = flag
        y = torch.zeros((B,) + tuple(outs) + (C,), device='cuda')
        idx = torch.zeros(y.shape, dtype=torch.uint8, device='cuda')
        L.check(lib.ivf_maxpool3d_fwd(ctypes.byref(d), L.ptr(xcl), L.ptr(y), L.ptr(idx), L.stream()))
        dx = torch.full_like(xcl, float('nan'))
        L.check(lib.ivf_maxpool3d_bwd(ctypes.byref(d), L.ptr(gy), L.ptr(idx), L.ptr(dx), None if flag else L.ptr(xcl),
                                      0, L.stream()))
        res.append((y, dx, idx))
    assert torch.equal(res[0][0], res[1][0])
    assert torch.equal(res[0][1], res[1][1])
    assert bool((res[1][2] == 255).any()) and not bool((res[0][2] == 255).any())


@pytest.mark.parametrize("math", ["fp32", "bf16x3", "bf16x6"])
@pytest.mark.parametrize("k,cin,cout,thw", [(3, 32, 40, (5, 15, 30)), (3, 16, 200, (3, 9, 15)), (4, 8, 32, (4, 9, 10)),
                                            (1, 24, 72, (2, 5, 7)),
                                            # channel counts that neither divide nor are divided by the 32-deep chunk of the
                                            # implicit GEMM: its running (tap, channel) state carries across chunks
                                            (3, 24, 40, (3, 7, 9)), (3, 12, 16, (2, 6, 7)), (2, 40, 24, (3, 6, 5))])
def test_every_conv_variant_matches_torch(k, cin, cout, thw, math):
    """Every kernel variant the tuner may pick (ivf_conv3d_variants: implicit GEMM tiles, all
    LDS-halo boxes) computes the same convolution: compared with torch's fp64 conv3d on
    ragged sizes (partial boxes in every dimension, partial channel tiles), with the fused
    scale/shift + ReLU epilogue and with the accumulate + ReLU-gate epilogue."""
    import torch.nn.functional as F
    import ivf_lib as L
    lib = L.lib()
    gen = torch.Generator().manual_seed(5)
    B = 2
    x = torch.randn((B, cin) + thw, generator=gen)
    w = torch.randn(cout, cin, k, k, k, generator=gen) * 0.1
    scale = torch.rand(cout, generator=gen) + 0.5
    shift = torch.randn(cout, generator=gen) * 0.1
    pf, pb = (k - 1) // 2, k - 1 - (k - 1) // 2
    ref = F.conv3d(F.pad(x.double(), (pf, pb, pf, pb, pf, pb)), w.double())
    ref_relu = torch.relu(ref * scale.double().view(1, -1, 1, 1, 1) + shift.double().view(1, -1, 1, 1, 1))
    mm = L.MATH_MODES[math]
    tol = 1e-4 if math == "bf16x3" else 1e-5   # the exact-fp32 MFMA and the 6-pass split: fp32-class
    wf = torch.empty(lib.ivf_conv3d_pack_fwd_elems(cout, cin, k, k, k, mm), device='cuda')
    wd, scd, shd = w.cuda(), scale.cuda(), shift.cuda()   # (kept alive: L.ptr only takes the address)
    L.check(lib.ivf_conv3d_pack_fwd(L.ptr(wd), L.ptr(wf), cout, cin, cin, k, k, k, mm, L.stream()))
    xcl = to_cl(x.cuda())
    d = L.ConvDesc()
    d.B, d.Ti, d.Hi, d.Wi = B, *thw
    d.Cin, d.in_ld, d.in_coff = cin, cin, 0
    d.To, d.Ho, d.Wo = thw
    d.Cout, d.out_ld, d.out_coff = cout, cout, 0
    d.kT = d.kH = d.kW = k
    d.sT = d.sH = d.sW = 1
    d.pT = d.pH = d.pW = pf
    d.math = mm
    d.mask_ld, d.mask_coff = cout, 0
    ids = (ctypes.c_int * 96)()
    n = lib.ivf_conv3d_variants(ctypes.byref(d), ids, 96)
    assert n >= 3
    base = torch.randn((B,) + thw + (cout,), generator=gen).cuda()
    gate = (torch.rand((B,) + thw + (cout,), generator=gen) > 0.3).float().cuda()
    want_acc = (base.double().cpu() + ref.permute(0, 2, 3, 4, 1)) * gate.double().cpu()
    ran = 0
    for v in list(ids)[:n]:
        d.variant = v
        d.relu, d.accumulate = 1, 0
        y = torch.full((B,) + thw + (cout,), float('nan'), device='cuda')
        rc = lib.ivf_conv3d(ctypes.byref(d), L.ptr(xcl), L.ptr(wf), L.ptr(scd), L.ptr(shd), None, L.ptr(y), L.stream())
        if rc != 0:
            continue   # variant not applicable to this shape (LDS budget)
        ran += 1
        got = from_cl(y, cout).double().cpu()
        assert rel_err(got.numpy(), ref_relu.numpy()) < tol, f"variant {v}"
        d.relu, d.accumulate = 0, 1
        acc = base.clone()
        L.check(lib.ivf_conv3d(ctypes.byref(d), L.ptr(xcl), L.ptr(wf), None, None, L.ptr(gate), L.ptr(acc), L.stream()))
        assert rel_err(acc.double().cpu().numpy(), want_acc.numpy()) < tol, f"variant {v} (accumulate)"
    assert ran >= 3


@pytest.mark.parametrize("k0,k1,cout,thw,math", [(40, 24, 72, (3, 7, 9), "bf16x3"), (96, 16, 200, (2, 5, 11), "bf16x3"),
                                                 (32, 8, 48, (1, 9, 7), "bf16x3"), (40, 24, 72, (3, 7, 9), "fp32")])
def test_pointwise_conv_two_sources(k0, k1, cout, thw, math):
    """The pointwise (1x1x1) path of the implicit GEMM with its second input (ivf_conv3d_desc.in2 / K0: one
    backward GEMM over gradients living in two buffers): K0 off the 32-wide chunk grid, a k tail, row counts off the
    tile grid, sources with row lengths / channel offsets of their own, every tile variant, with the
    accumulate + ReLU-gate epilogue -- against torch fp64 on the concatenated input."""
    import ivf_lib as L
    lib = L.lib()
    gen = torch.Generator().manual_seed(11)
    B = 2
    K = k0 + k1
    x1 = torch.randn((B,) + thw + (k0 + 8,), generator=gen)     # channels [4, 4 + k0) are the first source
    x2 = torch.randn((B,) + thw + (k1 + 12,), generator=gen)    # channels [8, 8 + k1) the second
    w = torch.randn(cout, K, 1, 1, 1, generator=gen) * 0.1
    xin = torch.cat([x1[..., 4:4 + k0], x2[..., 8:8 + k1]], dim=-1).double()
    ref = torch.einsum('bthwk,nk->bthwn', xin, w.view(cout, K).double())
    mm = L.MATH_MODES[math]
    wf = torch.empty(lib.ivf_conv3d_pack_fwd_elems(cout, K, 1, 1, 1, mm), device='cuda')
    wd = w.cuda()
    L.check(lib.ivf_conv3d_pack_fwd(L.ptr(wd), L.ptr(wf), cout, K, K, 1, 1, 1, mm, L.stream()))
    x1d, x2d = x1.cuda(), x2.cuda()
    d = L.ConvDesc()
    d.B, d.Ti, d.Hi, d.Wi = B, *thw
    d.Cin, d.in_ld, d.in_coff = K, k0 + 8, 4
    d.K0, d.in2_ld, d.in2_coff, d.in2 = k0, k1 + 12, 8, x2d.data_ptr()
    d.To, d.Ho, d.Wo = thw
    d.Cout, d.out_ld, d.out_coff = cout, cout + 4, 4
    d.kT = d.kH = d.kW = 1
    d.sT = d.sH = d.sW = 1
    d.math = mm
    d.mask_ld, d.mask_coff = cout, 0
    ids = (ctypes.c_int * 96)()
    n = lib.ivf_conv3d_variants(ctypes.byref(d), ids, 96)
    base = torch.randn((B,) + thw + (cout + 4,), generator=gen)
    gate = (torch.rand((B,) + thw + (cout,), generator=gen) > 0.3).float()
    want = (base[..., 4:].double() + ref) * gate.double()
    gated, ran = gate.cuda(), 0
    tol = 1e-4 if math == "bf16x3" else 1e-5
    for v in list(ids)[:n]:
        d.variant = v
        d.relu, d.accumulate = 0, 1
        acc = base.clone().cuda()
        rc = lib.ivf_conv3d(ctypes.byref(d), L.ptr(x1d), L.ptr(wf), None, None, L.ptr(gated), L.ptr(acc), L.stream())
        if rc != 0:
            continue
        ran += 1
        got = acc.cpu()
        assert torch.equal(got[..., :4], base[..., :4]), f"variant {v} wrote outside its channel window"
        assert rel_err(got[..., 4:].double().numpy(), want.numpy()) < tol, f"variant {v}"
    assert ran >= 3


@pytest.mark.parametrize("cin,n0,n1,thw,math", [(40, 24, 50, (3, 7, 9), "bf16x3"), (64, 112, 40, (2, 5, 11), "bf16x3"),
                                                (40, 24, 50, (3, 7, 9), "fp32")])
def test_conv_two_output_windows(cin, n0, n1, thw, math):
    """ivf_conv3d_desc.out2 / N0: one GEMM whose columns [0, N0) land in one buffer and [N0, Cout) in another (the
    forward of an Inception module's b0 | b1a | b2a), N0 off the 32-column tile grid, with the BN + ReLU epilogue,
    every implicit-GEMM tile; the bytes around both windows must stay untouched."""
    import ivf_lib as L
    lib = L.lib()
    gen = torch.Generator().manual_seed(13)
    B, cout = 2, n0 + n1
    x = torch.randn((B,) + thw + (cin,), generator=gen)
    w = torch.randn(cout, cin, 1, 1, 1, generator=gen) * 0.1
    scale = torch.rand(cout, generator=gen) + 0.5
    shift = torch.randn(cout, generator=gen) * 0.1
    ref = torch.relu(torch.einsum('bthwk,nk->bthwn', x.double(), w.view(cout, cin).double()) * scale.double() + shift.double())
    mm = L.MATH_MODES[math]
    wf = torch.empty(lib.ivf_conv3d_pack_fwd_elems(cout, cin, 1, 1, 1, mm), device='cuda')
    wd, scd, shd, xd = w.cuda(), scale.cuda(), shift.cuda(), x.cuda()
    L.check(lib.ivf_conv3d_pack_fwd(L.ptr(wd), L.ptr(wf), cout, cin, cin, 1, 1, 1, mm, L.stream()))
    d = L.ConvDesc()
    d.B, d.Ti, d.Hi, d.Wi = B, *thw
    d.Cin, d.in_ld, d.in_coff = cin, cin, 0
    d.To, d.Ho, d.Wo = thw
    d.Cout, d.out_ld, d.out_coff = cout, n0 + 12, 8          # first window: channels [8, 8 + n0) of a wider buffer
    d.kT = d.kH = d.kW = 1
    d.sT = d.sH = d.sW = 1
    d.relu, d.math = 1, mm
    ids = (ctypes.c_int * 96)()
    n = lib.ivf_conv3d_variants(ctypes.byref(d), ids, 96)
    ran = 0
    tol = 1e-4 if math == "bf16x3" else 1e-5
    for v in list(ids)[:n]:
        y1 = torch.full((B,) + thw + (n0 + 12,), 7.0, device='cuda')
        y2 = torch.full((B,) + thw + (n1 + 4,), 9.0, device='cuda')
        d.variant = v
        d.N0, d.out2_ld, d.out2_coff, d.out2 = n0, n1 + 4, 4, y2.data_ptr()
        rc = lib.ivf_conv3d(ctypes.byref(d), L.ptr(xd), L.ptr(wf), L.ptr(scd), L.ptr(shd), None, L.ptr(y1), L.stream())
        if rc != 0:
            continue   # the LDS-halo / pix4 variants refuse a second window
        ran += 1
        g1, g2 = y1.cpu(), y2.cpu()
        assert bool((g1[..., :8] == 7.0).all()) and bool((g1[..., 8 + n0:] == 7.0).all()), f"variant {v}"
        assert bool((g2[..., :4] == 9.0).all())
        assert rel_err(g1[..., 8:8 + n0].double().numpy(), ref[..., :n0].numpy()) < tol, f"variant {v}"
        assert rel_err(g2[..., 4:].double().numpy(), ref[..., n0:].numpy()) < tol, f"variant {v} (second window)"
    assert ran >= 3


@pytest.mark.parametrize("k,cin,n0,n1,thw", [(1, 40, 24, 48, (3, 7, 9)), (3, 32, 48, 0, (4, 9, 10)), (1, 64, 112, 40, (2, 5, 11))])
def test_one_bit_relu_gates(k, cin, n0, n1, thw):
    """ivf_conv3d_desc.gate_out / gate_out2 / gate_in: a forward epilogue records (stored value > 0) as 1 bit per
    element (1 byte per 8 channels), for the main and the second output window; a backward epilogue gated by that
    record must equal, bit for bit, the one gated by the fp32 activation (relu_mask).  Every variant that serves the
    shape; channel windows inside wider rows; ragged positions."""
    import torch.nn.functional as F
    import ivf_lib as L
    lib = L.lib()
    gen = torch.Generator().manual_seed(17)
    B, cout = 2, n0 + n1
    mm = L.MATH_MODES["bf16x3"]
    x = torch.randn((B,) + thw + (cin,), generator=gen).cuda()
    w = (torch.randn(cout, cin, k, k, k, generator=gen) * 0.1).cuda()
    wf = torch.empty(lib.ivf_conv3d_pack_fwd_elems(cout, cin, k, k, k, mm), device='cuda')
    L.check(lib.ivf_conv3d_pack_fwd(L.ptr(w), L.ptr(wf), cout, cin, cin, k, k, k, mm, L.stream()))
    d = L.ConvDesc()
    d.B, d.Ti, d.Hi, d.Wi = B, *thw
    d.Cin, d.in_ld, d.in_coff = cin, cin, 0
    d.To, d.Ho, d.Wo = thw
    d.kT = d.kH = d.kW = k
    d.sT = d.sH = d.sW = 1
    d.pT = d.pH = d.pW = (k - 1) // 2
    d.relu, d.math = 1, mm
    d.Cout, d.out_ld, d.out_coff = cout, n0 + 16, 8
    ids = (ctypes.c_int * 96)()
    P = B * thw[0] * thw[1] * thw[2]
    ran = 0
    for v in list(ids)[:lib.ivf_conv3d_variants(ctypes.byref(d), ids, 96)]:
        y1 = torch.zeros((B,) + thw + (n0 + 16,), device='cuda')
        y2 = torch.zeros((B,) + thw + (max(n1, 8) + 8,), device='cuda')
        b1 = torch.full((P, (n0 + 16) // 8), 0xAA, dtype=torch.uint8, device='cuda')
        b2 = torch.full((P, (max(n1, 8) + 8) // 8), 0xAA, dtype=torch.uint8, device='cuda')
        d.variant = v
        d.gate_out, d.gate_out_ld, d.gate_out_coff = b1.data_ptr(), (n0 + 16) // 8, 8
        if n1:
            d.N0, d.out2_ld, d.out2_coff, d.out2 = n0, n1 + 8, 8, y2.data_ptr()
            d.gate_out2, d.gate_out2_ld = b2.data_ptr(), (n1 + 8) // 8
        rc = lib.ivf_conv3d(ctypes.byref(d), L.ptr(x), L.ptr(wf), None, None, None, L.ptr(y1), L.stream())
        if rc != 0:
            continue
        ran += 1
        for y, bits, c0, n in ((y1, b1, 8, n0), (y2, b2, 8, n1)):
            if not n:
                continue
            want = (y.view(P, -1)[:, c0:c0 + n] > 0)
            got = ((bits[:, :, None] >> torch.arange(8, device='cuda', dtype=torch.uint8)) & 1).bool().view(P, -1)
            assert torch.equal(got[:, c0:c0 + n], want), f"variant {v}"
            assert bool((bits[:, :c0 // 8] == 0xAA).all()) and bool((bits[:, (c0 + n) // 8:] == 0xAA).all()), f"variant {v}"
        # consumer side: a 1x1x1 "backward" GEMM producing n0 channels, accumulate + gate, bits vs fp32 activation
        if ran == 1:
            gin = torch.randn(P, 32, generator=gen).cuda()
            wb = (torch.randn(n0, 32, 1, 1, 1, generator=gen) * 0.1).cuda()
            wbp = torch.empty(lib.ivf_conv3d_pack_fwd_elems(n0, 32, 1, 1, 1, mm), device='cuda')
            L.check(lib.ivf_conv3d_pack_fwd(L.ptr(wb), L.ptr(wbp), n0, 32, 32, 1, 1, 1, mm, L.stream()))
            e = L.ConvDesc()
            e.B, e.Ti, e.Hi, e.Wi = B, *thw
            e.Cin, e.in_ld, e.in_coff = 32, 32, 0
            e.To, e.Ho, e.Wo = thw
            e.kT = e.kH = e.kW = 1
            e.sT = e.sH = e.sW = 1
            e.math, e.accumulate = mm, 1
            e.Cout, e.out_ld, e.out_coff = n0, n0 + 16, 8
            e.mask_ld, e.mask_coff = n0 + 16, 8
            base = torch.randn((P, n0 + 16), generator=gen).cuda()
            o1, o2 = base.clone(), base.clone()
            L.check(lib.ivf_conv3d(ctypes.byref(e), L.ptr(gin), L.ptr(wbp), None, None, L.ptr(y1), L.ptr(o1), L.stream()))
            e.gate_in, e.gate_in_ld, e.gate_in_coff = b1.data_ptr(), (n0 + 16) // 8, 8
            L.check(lib.ivf_conv3d(ctypes.byref(e), L.ptr(gin), L.ptr(wbp), None, None, None, L.ptr(o2), L.stream()))
            assert torch.equal(o1, o2)
            assert bool((o1[:, 8:8 + n0][y1.view(P, -1)[:, 8:8 + n0] <= 0] == 0).all())
    assert ran >= 3


@pytest.mark.parametrize("k,st,thw", [((1, 3, 3), (1, 2, 2), (3, 17, 20)), ((3, 3, 3), (2, 2, 2), (5, 15, 14)),
                                      ((2, 2, 2), (2, 2, 2), (4, 14, 14)), ((3, 3, 3), (1, 2, 2), (4, 15, 9)),
                                      ((2, 2, 2), (1, 2, 2), (4, 8, 7)), ((2, 2, 2), (2, 2, 2), (3, 7, 7))])
def test_strided_maxpool_with_ties_matches_torch(k, st, thw):
    """Strided pools (fixed-geometry backward kernels) against torch's CPU max_pool3d over the
    zero-padded input (TF-'same' padding, I3D_doubled.py:8-40) on tie-rich inputs."""
    import torch.nn.functional as F
    import ivf_arch as arch
    import ivf_lib as L
    lib = L.lib()
    gen = torch.Generator().manual_seed(9)
    B, C = 2, 40
    x = torch.relu(torch.round(torch.randn((B, C) + thw, generator=gen) * 3) / 3)
    pads = [arch.same_pad(n, kk, ss) for n, kk, ss in zip(thw, k, st)]
    outs = [arch.out_size(n, kk, ss) for n, kk, ss in zip(thw, k, st)]
    gy = torch.randn((B, C) + tuple(outs), generator=gen)
    xr = x.clone().requires_grad_()
    y = F.max_pool3d(F.pad(xr, (pads[2][0], pads[2][1], pads[1][0], pads[1][1], pads[0][0], pads[0][1])), k, st)
    assert list(y.shape[2:]) == outs
    y.backward(gy)
    xcl = to_cl(x.cuda(), C)
    ycl = torch.zeros((B,) + tuple(outs) + (C,), device='cuda')
    idx = torch.zeros(ycl.shape, dtype=torch.uint8, device='cuda')
    d = L.PoolDesc()
    d.B, d.Ti, d.Hi, d.Wi, d.C, d.in_ld, d.in_coff = B, *thw, C, C, 0
    d.To, d.Ho, d.Wo, d.out_ld, d.out_coff = *outs, C, 0
    d.kT, d.kH, d.kW = k
    d.sT, d.sH, d.sW = st
    d.pT, d.pH, d.pW = [p[0] for p in pads]
    L.check(lib.ivf_maxpool3d_fwd(ctypes.byref(d), L.ptr(xcl), L.ptr(ycl), L.ptr(idx), L.stream()))
    assert np.array_equal(from_cl(ycl, C).cpu().numpy(), y.detach().numpy())
    gycl = to_cl(gy.cuda(), C)
    dxcl = torch.full_like(xcl, float('nan'))
    L.check(lib.ivf_maxpool3d_bwd(ctypes.byref(d), L.ptr(gycl), L.ptr(idx), L.ptr(dxcl), None, 0, L.stream()))
    dx = from_cl(dxcl, C).cpu().numpy()
    ref = xr.grad.numpy()
    assert np.array_equal(dx != 0, ref != 0)
    assert np.allclose(dx, ref, rtol=1e-5, atol=1e-6)
    base = torch.randn_like(xcl)
    acc = base.clone()
    L.check(lib.ivf_maxpool3d_bwd(ctypes.byref(d), L.ptr(gycl), L.ptr(idx), L.ptr(acc), L.ptr(xcl), 1, L.stream()))
    want = torch.where(xcl > 0, base + dxcl, torch.zeros_like(base))
    assert torch.allclose(acc, want, rtol=1e-6, atol=1e-6)


# ------------------------------------------------------------------ bf16 activation storage (IVF_MATH_BF16ACT)
def _pool_desc(L, B, thw, C, k, s, bf16):
    import ivf_arch as arch
    pads = [arch.same_pad(n, kk, ss)[0] for n, kk, ss in zip(thw, k, s)]
    outs = [arch.out_size(n, kk, ss) for n, kk, ss in zip(thw, k, s)]
    d = L.PoolDesc()
    d.B, d.Ti, d.Hi, d.Wi, d.C, d.in_ld, d.in_coff = B, *thw, C, C, 0
    d.To, d.Ho, d.Wo, d.out_ld, d.out_coff = *outs, C, 0
    d.kT, d.kH, d.kW = k
    d.sT, d.sH, d.sW = s
    d.pT, d.pH, d.pW = pads
    d.act_bf16 = 1 if bf16 else 0
    return d, outs


@pytest.mark.parametrize("k,st,thw,C", [((1, 3, 3), (1, 2, 2), (3, 17, 20), 24), ((3, 3, 3), (2, 2, 2), (5, 15, 14), 40),
                                        ((2, 2, 2), (2, 2, 2), (4, 8, 10), 16), ((3, 3, 3), (1, 1, 1), (4, 9, 8), 48),
                                        # channel counts that are not a multiple of 8: the 4-channel fallbacks of the
                                        # 16-byte-per-lane bf16 kernels (no lane pairs, generic forward, 4-channel backward)
                                        ((3, 3, 3), (1, 1, 1), (4, 9, 8), 12), ((1, 3, 3), (1, 2, 2), (3, 17, 20), 20),
                                        ((3, 3, 3), (2, 2, 2), (5, 15, 14), 28)])
def test_maxpool_bf16_storage_equals_fp32_path(k, st, thw, C):
    """Every pool kernel family with bf16 storage (ivf_pool3d_desc.act_bf16) against the fp32 path on the same
    bf16-representable tensors: the forward selects (identical values and arg-max, ties included), the backward sums in
    fp32 and rounds once (= the fp32 result rounded to bf16), with and without accumulate + ReLU gate."""
    import ivf_lib as L
    lib = L.lib()
    gen = torch.Generator().manual_seed(11)
    B = 2
    x = (torch.randint(0, 6, (B,) + thw + (C,), generator=gen).float() / 4).bfloat16().cuda()   # many exact ties
    res = {}
    for bf in (False, True):
        d, outs = _pool_desc(L, B, thw, C, k, st, bf)
        dt = torch.bfloat16 if bf else torch.float32
        xin = x.to(dt).contiguous()
        y = torch.zeros((B,) + tuple(outs) + (C,), dtype=dt, device='cuda')
        idx = torch.zeros((B,) + tuple(outs) + (C,), dtype=torch.uint8, device='cuda')
        L.check(lib.ivf_maxpool3d_fwd(ctypes.byref(d), L.ptr(xin), L.ptr(y), L.ptr(idx), L.stream()))
        gy = torch.randn((B,) + tuple(outs) + (C,), generator=gen).bfloat16().to(dt).cuda() if not res else res['gy'].to(dt)
        dx = torch.full((B,) + thw + (C,), float('nan'), dtype=dt, device='cuda')
        L.check(lib.ivf_maxpool3d_bwd(ctypes.byref(d), L.ptr(gy), L.ptr(idx), L.ptr(dx), None, 0, L.stream()))
        base = (torch.randn((B,) + thw + (C,), generator=torch.Generator().manual_seed(3)).bfloat16()).to(dt).cuda()
        acc = base.clone()
        L.check(lib.ivf_maxpool3d_bwd(ctypes.byref(d), L.ptr(gy), L.ptr(idx), L.ptr(acc), L.ptr(xin), 1, L.stream()))
        if not res:
            res = dict(y=y, idx=idx, dx=dx, acc=acc, gy=gy)
        else:
            assert torch.equal(y.float(), res['y']) and torch.equal(idx, res['idx'])
            assert torch.equal(dx, res['dx'].bfloat16())
            assert torch.equal(acc, res['acc'].bfloat16())


@pytest.mark.parametrize("k,cin,cout,thw", [(3, 32, 40, (5, 15, 30)), (3, 16, 200, (3, 9, 15)), (1, 24, 72, (2, 5, 7)),
                                            (1, 64, 136, (4, 14, 14))])
def test_every_conv_variant_bf16_activations(k, cin, cout, thw):
    """IVF_MATH_BF16ACT: every kernel variant on bf16 inputs against torch fp64 on the SAME bf16 values -- the
    products are as exact as in the split modes (weights hi/lo), so the only new error is the one RNE to bf16 of the
    stored result (2^-9 relative): forward with BN + ReLU, and the accumulate + ReLU-gate (bf16 gate tensor) form."""
    import torch.nn.functional as F
    import ivf_lib as L
    lib = L.lib()
    gen = torch.Generator().manual_seed(5)
    B = 2
    x = torch.randn((B, cin) + thw, generator=gen).bfloat16()
    w = torch.randn(cout, cin, k, k, k, generator=gen) * 0.1
    scale = torch.rand(cout, generator=gen) + 0.5
    shift = torch.randn(cout, generator=gen) * 0.1
    pf, pb = (k - 1) // 2, k - 1 - (k - 1) // 2
    ref = F.conv3d(F.pad(x.double(), (pf, pb, pf, pb, pf, pb)), w.double())
    ref_relu = torch.relu(ref * scale.double().view(1, -1, 1, 1, 1) + shift.double().view(1, -1, 1, 1, 1))
    mm = L.MATH_MODES["bf16act"]
    wf = torch.empty(lib.ivf_conv3d_pack_fwd_elems(cout, cin, k, k, k, mm), device='cuda')
    wd, scd, shd = w.cuda(), scale.cuda(), shift.cuda()
    L.check(lib.ivf_conv3d_pack_fwd(L.ptr(wd), L.ptr(wf), cout, cin, cin, k, k, k, mm, L.stream()))
    xcl = x.cuda().permute(0, 2, 3, 4, 1).contiguous()          # bf16 channels-last
    d = L.ConvDesc()
    d.B, d.Ti, d.Hi, d.Wi = B, *thw
    d.Cin, d.in_ld, d.in_coff = cin, cin, 0
    d.To, d.Ho, d.Wo = thw
    d.Cout, d.out_ld, d.out_coff = cout, cout, 0
    d.kT = d.kH = d.kW = k
    d.sT = d.sH = d.sW = 1
    d.pT = d.pH = d.pW = pf
    d.math = mm
    d.mask_ld, d.mask_coff = cout, 0
    ids = (ctypes.c_int * 96)()
    n = lib.ivf_conv3d_variants(ctypes.byref(d), ids, 96)
    assert n >= 3
    base = torch.randn((B,) + thw + (cout,), generator=gen).bfloat16().cuda()
    gate = (torch.rand((B,) + thw + (cout,), generator=gen) > 0.3).to(torch.bfloat16).cuda()
    want_acc = (base.double().cpu() + ref.permute(0, 2, 3, 4, 1)) * gate.double().cpu()

    def close(got, want, what):
        got, want = got.double().cpu(), want.double()
        tol = 2.0 ** -8 * want.abs() + 1e-5 * want.abs().max()      # half an ulp of bf16 is 2^-9 relative
        assert bool(((got - want).abs() <= tol).all()), what

    ran = 0
    for v in list(ids)[:n]:
        d.variant = v
        d.relu, d.accumulate = 1, 0
        y = torch.full((B,) + thw + (cout,), float('nan'), dtype=torch.bfloat16, device='cuda')
        rc = lib.ivf_conv3d(ctypes.byref(d), L.ptr(xcl), L.ptr(wf), L.ptr(scd), L.ptr(shd), None, L.ptr(y), L.stream())
        if rc != 0:
            continue
        ran += 1
        close(y, ref_relu.permute(0, 2, 3, 4, 1), f"variant {v}")
        d.relu, d.accumulate = 0, 1
        acc = base.clone()
        L.check(lib.ivf_conv3d(ctypes.byref(d), L.ptr(xcl), L.ptr(wf), None, None, L.ptr(gate), L.ptr(acc), L.stream()))
        close(acc, want_acc, f"variant {v} (accumulate)")
    assert ran >= 3


def test_bf16_activation_stem_ends_keep_fp32():
    """The two mixed-storage ends of an IVF_MATH_BF16ACT network: the 4-channel-pixel strided conv reads fp32 pixels
    and writes bf16 (pix4 only), its depth-to-space backward reads bf16 gradients and writes the fp32 input gradient
    (LDS-halo only): against torch fp64 on the same values; every stem-backward variant the tuner may pick."""
    import torch.nn.functional as F
    import ivf_arch as arch
    import ivf_lib as L
    lib = L.lib()
    gen = torch.Generator().manual_seed(9)
    B, cin, cinp, cout, k, s, thw = 2, 3, 4, 64, (7, 7, 7), (2, 2, 2), (8, 30, 36)
    x = (torch.rand((B, cin) + thw, generator=gen) * 255).round()
    w = torch.randn((cout, cin) + k, generator=gen) * 0.02
    scale = torch.rand(cout, generator=gen) + 0.5
    shift = torch.randn(cout, generator=gen) * 0.1
    pads = [arch.same_pad(n, kk, ss) for n, kk, ss in zip(thw, k, s)]
    outs = [arch.out_size(n, kk, ss) for n, kk, ss in zip(thw, k, s)]
    xp = F.pad(x.double(), (pads[2][0], pads[2][1], pads[1][0], pads[1][1], pads[0][0], pads[0][1])).requires_grad_()
    y_ref = F.conv3d(xp, w.double(), stride=s) * scale.double().view(1, -1, 1, 1, 1) + shift.double().view(1, -1, 1, 1, 1)
    mm = L.MATH_MODES["bf16act"]
    wd, scd, shd = w.cuda(), scale.cuda(), shift.cuda()
    wf = torch.empty(lib.ivf_conv3d_pack_fwd_elems(cout, cinp, *k, mm), device='cuda')
    L.check(lib.ivf_conv3d_pack_fwd(L.ptr(wd), L.ptr(wf), cout, cin, cinp, *k, mm, L.stream()))
    xcl = to_cl(x.cuda(), cinp)                                   # fp32 pixels
    d = L.ConvDesc()
    d.B, d.Ti, d.Hi, d.Wi = B, *thw
    d.Cin, d.in_ld, d.in_coff = cinp, cinp, 0
    d.To, d.Ho, d.Wo = outs
    d.Cout, d.out_ld, d.out_coff = cout, cout, 0
    d.kT, d.kH, d.kW = k
    d.sT, d.sH, d.sW = s
    d.pT, d.pH, d.pW = [p[0] for p in pads]
    d.relu, d.math = 1, mm
    ids = (ctypes.c_int * 96)()
    assert lib.ivf_conv3d_variants(ctypes.byref(d), ids, 96) == 1 and ids[0] == 15      # IVF_CONV_PIX4
    y = torch.full((B,) + tuple(outs) + (cout,), float('nan'), dtype=torch.bfloat16, device='cuda')
    L.check(lib.ivf_conv3d(ctypes.byref(d), L.ptr(xcl), L.ptr(wf), L.ptr(scd), L.ptr(shd), None, L.ptr(y), L.stream()))
    want = torch.relu(y_ref).detach().permute(0, 2, 3, 4, 1)
    got = y.double().cpu()
    assert bool(((got - want).abs() <= 2.0 ** -8 * want.abs() + 1e-5 * want.abs().max()).all())
    # backward-data: bf16 upstream gradient -> fp32 input gradient, BN scale folded in the pack
    gy = torch.randn((B,) + tuple(outs) + (cout,), generator=gen).bfloat16()
    (y_ref * gy.double().permute(0, 4, 1, 2, 3)).sum().backward()
    dx_ref = xp.grad[:, :, pads[0][0]:pads[0][0] + thw[0], pads[1][0]:pads[1][0] + thw[1], pads[2][0]:pads[2][0] + thw[2]]
    wb = torch.empty(lib.ivf_conv3d_pack_bwd_elems(cout, cinp, *k, *s, *[p[0] for p in pads], mm), device='cuda')
    geom = L.BwdGeom()
    L.check(lib.ivf_conv3d_pack_bwd(L.ptr(wd), L.ptr(scd), L.ptr(wb), cout, cin, cinp, *k, *s, *[p[0] for p in pads], mm,
                                    ctypes.byref(geom), L.stream()))
    assert geom.d2s == 1
    e = L.ConvDesc()
    e.B, e.Ti, e.Hi, e.Wi = B, *outs
    e.Cin, e.in_ld, e.in_coff = cout, cout, 0
    e.kT, e.kH, e.kW = geom.kT, geom.kH, geom.kW
    e.sT = e.sH = e.sW = 1
    e.pT, e.pH, e.pW = geom.pT, geom.pH, geom.pW
    e.out_ld, e.out_coff = cinp, 0
    e.math = mm
    e.d2s = 1
    e.bsT, e.bsH, e.bsW = s
    e.To, e.Ho, e.Wo = [-(-n // ss) for n, ss in zip(thw, s)]
    e.Cout = geom.rows
    e.dT, e.dH, e.dW = thw
    e.dC = cinp
    n = lib.ivf_conv3d_variants(ctypes.byref(e), ids, 96)
    assert n >= 3 and all(v >= 16 for v in list(ids)[:n])          # LDS-halo variants only
    gcl = gy.cuda().contiguous()
    ran = 0
    for v in list(ids)[:n]:
        e.variant = v
        dxcl = torch.full((B,) + thw + (cinp,), float('nan'), device='cuda')
        if lib.ivf_conv3d(ctypes.byref(e), L.ptr(gcl), L.ptr(wb), None, None, None, L.ptr(dxcl), L.stream()) != 0:
            continue
        ran += 1
        assert rel_err(from_cl(dxcl, cin).double().cpu().numpy(), dx_ref.numpy()) < 1e-4, f"variant {v}"
    assert ran >= 3


@pytest.mark.parametrize("math", ["bf16x3", "bf16x6"])
def test_every_stem_backward_variant_matches_torch(math):
    """The depth-to-space (stride-2 backward-data) epilogue through EVERY variant the tuner may pick for the stem
    (the 7x7x7 / stride-2 unit as a 4x4x4 block conv over dY), against torch fp64 autograd."""
    import torch.nn.functional as F
    import ivf_arch as arch
    import ivf_lib as L
    lib = L.lib()
    gen = torch.Generator().manual_seed(21)
    B, cin, cinp, cout, k, s, thw = 2, 3, 4, 64, (7, 7, 7), (2, 2, 2), (7, 30, 35)     # odd sizes: partial blocks
    w = torch.randn((cout, cin) + k, generator=gen) * 0.02
    scale = torch.rand(cout, generator=gen) + 0.5
    pads = [arch.same_pad(n, kk, ss) for n, kk, ss in zip(thw, k, s)]
    outs = [arch.out_size(n, kk, ss) for n, kk, ss in zip(thw, k, s)]
    xp = torch.zeros((B, cin) + tuple(n + p[0] + p[1] for n, p in zip(thw, pads)), dtype=torch.float64, requires_grad=True)
    gy = torch.randn((B, cout) + tuple(outs), generator=gen)
    (F.conv3d(xp, w.double(), stride=s) * scale.double().view(1, -1, 1, 1, 1) * gy.double()).sum().backward()
    dx_ref = xp.grad[:, :, pads[0][0]:pads[0][0] + thw[0], pads[1][0]:pads[1][0] + thw[1], pads[2][0]:pads[2][0] + thw[2]]
    mm = L.MATH_MODES[math]
    wd, scd = w.cuda(), scale.cuda()
    wb = torch.empty(lib.ivf_conv3d_pack_bwd_elems(cout, cinp, *k, *s, *[p[0] for p in pads], mm), device='cuda')
    geom = L.BwdGeom()
    L.check(lib.ivf_conv3d_pack_bwd(L.ptr(wd), L.ptr(scd), L.ptr(wb), cout, cin, cinp, *k, *s, *[p[0] for p in pads], mm,
                                    ctypes.byref(geom), L.stream()))
    e = L.ConvDesc()
    e.B, e.Ti, e.Hi, e.Wi = B, *outs
    e.Cin, e.in_ld, e.in_coff = cout, cout, 0
    e.kT, e.kH, e.kW = geom.kT, geom.kH, geom.kW
    e.sT = e.sH = e.sW = 1
    e.pT, e.pH, e.pW = geom.pT, geom.pH, geom.pW
    e.out_ld, e.out_coff = cinp, 0
    e.math = mm
    e.d2s = 1
    e.bsT, e.bsH, e.bsW = s
    e.To, e.Ho, e.Wo = [-(-n // ss) for n, ss in zip(thw, s)]
    e.Cout = geom.rows
    e.dT, e.dH, e.dW = thw
    e.dC = cinp
    ids = (ctypes.c_int * 96)()
    n = lib.ivf_conv3d_variants(ctypes.byref(e), ids, 96)
    gcl = to_cl(gy.cuda())
    ran = 0
    for v in list(ids)[:n]:
        e.variant = v
        dxcl = torch.full((B,) + thw + (cinp,), float('nan'), device='cuda')
        if lib.ivf_conv3d(ctypes.byref(e), L.ptr(gcl), L.ptr(wb), None, None, None, L.ptr(dxcl), L.stream()) != 0:
            continue
        ran += 1
        assert rel_err(from_cl(dxcl, cin).double().cpu().numpy(), dx_ref.numpy()) < (1e-4 if math == "bf16x3" else 1e-5), \
            f"variant {v}"
    assert ran >= 8
