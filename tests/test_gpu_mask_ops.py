"""HIP mask kernels through the C-ABI vs outputs of the reference's mask.py
(tests/golden/mask_ops.npz) and vs the CPU oracle on seeded inputs."""
import numpy as np
import pytest
import torch

from conftest import rel_err

pytestmark = pytest.mark.gpu


def _freeze(x, m, per_clip=False, cpad=0):
    import ivf_lib as L
    B, C, T, H, W = x.shape
    out = torch.empty_like(x) if cpad == 0 else torch.empty(B, T, H, W, cpad, device='cuda')
    L.check(L.lib().ivf_freeze_fwd(L.ptr(x), L.ptr(m), L.ptr(out), B, C, T, H * W, int(per_clip), cpad,
                                   L.stream()))
    return out


def _freeze_bwd(x, m, g, per_clip=False, cpad=0, want_dx=True):
    import ivf_lib as L
    B, C, T, H, W = x.shape
    ws = torch.empty(L.lib().ivf_freeze_bwd_workspace_bytes(B, T), dtype=torch.uint8, device='cuda')
    dm = torch.empty(B, T, device='cuda')
    dx = torch.empty_like(x) if want_dx else None
    L.check(L.lib().ivf_freeze_bwd(L.ptr(x), L.ptr(m), L.ptr(g), L.ptr(dm), L.ptr(dx), B, C, T, H * W,
                                   int(per_clip), cpad, L.ptr(ws), L.stream()))
    return dm, dx


@pytest.mark.parametrize("T", [16, 32])
def test_freeze_fwd_bwd_vs_reference(T, golden):
    import ivf_recipe as R
    g = golden('mask_ops')
    x = torch.from_numpy(R.uniform(f'g/freeze/x{T}', (2, 3, T, 12, 20), 0, 255)).cuda()
    m = torch.from_numpy(R.uniform(f'g/freeze/m{T}', (T,), 0, 1)).cuda()
    gy = torch.from_numpy(R.uniform(f'g/freeze/g{T}', (2, 3, T, 12, 20), -1, 1)).cuda()
    p = _freeze(x, m)
    assert rel_err(p.cpu().numpy(), g[f'freeze{T}_p']) < 1e-6
    dm, dx = _freeze_bwd(x, m, gy)
    assert rel_err(dm.sum(0).cpu().numpy(), g[f'freeze{T}_dm']) < 1e-5
    assert rel_err(dx.cpu().numpy(), g[f'freeze{T}_dx']) < 1e-6
    # channels-last variants (what the network plan consumes / produces)
    pcl = _freeze(x, m, cpad=4)
    assert torch.equal(pcl[..., :3].permute(0, 4, 1, 2, 3), p)
    assert float(pcl[..., 3].abs().max()) == 0.0
    gcl = torch.zeros(2, T, 12, 20, 4, device='cuda')
    gcl[..., :3] = gy.permute(0, 2, 3, 4, 1)
    dm2, _ = _freeze_bwd(x, m, gcl.contiguous(), cpad=4, want_dx=False)
    # the channels-last kernel sums pixels in a different order (one thread per pixel)
    assert rel_err(dm2.cpu().numpy(), dm.cpu().numpy()) < 1e-6
    assert rel_err(dm2.sum(0).cpu().numpy(), g[f'freeze{T}_dm']) < 1e-5


def test_freeze_per_clip_masks_match_oracle():
    import ivf_recipe as R
    from oracle import mask_ref
    x = torch.from_numpy(R.uniform('t/freeze/pc/x', (3, 3, 16, 9, 7), 0, 255))
    m = torch.from_numpy(R.uniform('t/freeze/pc/m', (3, 16), 0, 1))
    gy = torch.from_numpy(R.uniform('t/freeze/pc/g', (3, 3, 16, 9, 7), -1, 1))
    mr = m.clone().requires_grad_()
    pr = mask_ref.freeze(x, mr)
    (pr * gy).sum().backward()
    p = _freeze(x.cuda(), m.cuda(), per_clip=True)
    dm, _ = _freeze_bwd(x.cuda(), m.cuda(), gy.cuda(), per_clip=True)
    assert rel_err(p.cpu().numpy(), pr.detach().numpy()) < 1e-6
    assert rel_err(dm.cpu().numpy(), mr.grad.numpy()) < 1e-5
    # determinism: the two-stage reduction is bitwise reproducible
    dm_b, _ = _freeze_bwd(x.cuda(), m.cuda(), gy.cuda(), per_clip=True)
    assert torch.equal(dm, dm_b)


@pytest.mark.parametrize("T", [9, 24, 40])
def test_freeze_ragged_lengths_match_oracle(T):
    """Clip lengths that are not a kernel template size (16 / 32 / 64 frames): the guarded frames of the backward's
    up-front loads stay out of the scans, in both gradient layouts."""
    import ivf_recipe as R
    from oracle import mask_ref
    x = torch.from_numpy(R.uniform(f't/freeze/rag/x{T}', (2, 3, T, 6, 10), 0, 255))
    m = torch.from_numpy(R.uniform(f't/freeze/rag/m{T}', (2, T), 0, 1))
    gy = torch.from_numpy(R.uniform(f't/freeze/rag/g{T}', (2, 3, T, 6, 10), -1, 1))
    mr = m.clone().requires_grad_()
    xr = x.clone().requires_grad_()
    pr = mask_ref.freeze(xr, mr)
    (pr * gy).sum().backward()
    p = _freeze(x.cuda(), m.cuda(), per_clip=True)
    dm, dx = _freeze_bwd(x.cuda(), m.cuda(), gy.cuda(), per_clip=True)
    assert rel_err(p.cpu().numpy(), pr.detach().numpy()) < 1e-6
    assert rel_err(dm.cpu().numpy(), mr.grad.numpy()) < 1e-5
    assert rel_err(dx.cpu().numpy(), xr.grad.numpy()) < 1e-6
    gcl = torch.zeros(2, T, 6, 10, 4, device='cuda')
    gcl[..., :3] = gy.cuda().permute(0, 2, 3, 4, 1)
    dm2, _ = _freeze_bwd(x.cuda(), m.cuda(), gcl.contiguous(), per_clip=True, cpad=4, want_dx=False)
    assert rel_err(dm2.cpu().numpy(), mr.grad.numpy()) < 1e-5


def test_freeze_edge_masks():
    """all-zero mask = identity, all-one mask = every frame equals frame 0"""
    import ivf_recipe as R
    x = torch.from_numpy(R.uniform('t/freeze/edge', (1, 3, 16, 5, 5), 0, 255)).cuda()
    assert torch.equal(_freeze(x, torch.zeros(16, device='cuda')), x)
    p1 = _freeze(x, torch.ones(16, device='cuda'))
    assert torch.equal(p1, x[:, :, :1].expand_as(x))


REV = ['even_mid', 'odd_mid', 'ends', 'thresh', 'all_on', 'all_off', 'single']


@pytest.mark.parametrize("case", REV)
def test_reverse_and_submasks_bit_exact(case, golden):
    import ivf_lib as L
    import ivf_recipe as R
    g = golden('mask_ops')
    x = torch.from_numpy(R.uniform('g/reverse/x', (2, 3, 16, 6, 10), 0, 255)).cuda()
    m = torch.from_numpy(g[f'rev_{case}_mask']).cuda()
    run = torch.empty(16, dtype=torch.int32, device='cuda')
    partner = torch.empty(16, dtype=torch.int32, device='cuda')
    weight = torch.empty(16, device='cuda')
    L.check(L.lib().ivf_submask_pairs(L.ptr(m), 16, 0.1, L.ptr(run), L.ptr(partner), L.ptr(weight), L.stream()))
    # integer output: the run lists must equal find_submasks_from_mask exactly
    runs = {}
    for t, r in enumerate(run.cpu().tolist()):
        if r >= 0:
            runs.setdefault(r, []).append(t)
    flat = [-1]
    for r in sorted(runs):
        flat += runs[r] + [-1]
    assert flat == g[f'rev_{case}_subs'].tolist()
    p = torch.empty_like(x)
    L.check(L.lib().ivf_reverse_fwd(L.ptr(x), L.ptr(partner), L.ptr(weight), L.ptr(p), 2, 3, 16, 60, 0,
                                    L.stream()))
    assert rel_err(p.cpu().numpy(), g[f'rev_{case}_p']) < 1e-6


TV = ['rand16', 'rand32', 'mono', 'near_const', 'sig_pm5', 'const']


@pytest.mark.parametrize("case", TV)
def test_tv_norm_value_and_grad(case, golden):
    import ivf_lib as L
    g = golden('mask_ops')
    m = torch.from_numpy(g[f'tv_{case}_in']).cuda()[None].contiguous()
    T = m.shape[1]
    val = torch.empty(1, device='cuda')
    grad = torch.empty(1, T, device='cuda')
    L.check(L.lib().ivf_tv_norm(L.ptr(m), 1, T, 3.0, 3.0, L.ptr(val), L.ptr(grad), L.stream()))
    ref_v, ref_g = g[f'tv_{case}_val'], g[f'tv_{case}_grad']
    if case == 'const':
        # val == 0: the reference's autograd yields NaN gradients (mask.py:97)
        assert float(val) == 0.0 and np.isnan(ref_g).all() and torch.isnan(grad).all()
        return
    assert abs(float(val) - float(ref_v)) <= 2e-5 * abs(float(ref_v))
    assert rel_err(grad.cpu().numpy()[0], ref_g) < 2e-4


def test_regulariser_and_adam(golden):
    import ivf_lib as L
    import ivf_recipe as R
    g = golden('mask_ops')
    tm = torch.from_numpy(R.uniform('g/reg/tm', (16,), -5, 5)).cuda()[None].contiguous()
    sig = torch.empty(1, 16, device='cuda')
    terms = torch.empty(1, 2, device='cuda')
    dreg = torch.empty(1, 16, device='cuda')
    L.check(L.lib().ivf_mask_reg(L.ptr(tm), 1, 16, 0.01, 0.02, L.ptr(sig), L.ptr(terms), L.ptr(dreg), L.stream()))
    assert abs(float(terms.sum()) - float(g['reg_loss'])) < 1e-5 * abs(float(g['reg_loss']))
    grad_raw = (dreg * sig * (1 - sig)).cpu().numpy()[0]
    assert rel_err(grad_raw, g['reg_grad']) < 2e-4
    # Adam vs torch.optim.Adam trajectory
    p = torch.from_numpy(R.uniform('g/adam/p', (16,), -5, 5)).cuda()
    grads = torch.from_numpy(R.uniform('g/adam/g', (12, 16), -1e-2, 1e-2)).cuda()
    m = torch.zeros(16, device='cuda')
    v = torch.zeros(16, device='cuda')
    traj = [p.cpu().numpy().copy()]
    for i in range(12):
        L.check(L.lib().ivf_adam_step(L.ptr(p), L.ptr(grads[i].contiguous()), L.ptr(m), L.ptr(v), 16, i + 1, 0.2,
                                      0.9, 0.999, 1e-8, L.stream()))
        traj.append(p.cpu().numpy().copy())
    assert np.allclose(np.array(traj), g['adam_traj'], rtol=2e-6, atol=2e-6)


@pytest.mark.parametrize("T", [16, 32, 100])
def test_rank_frames_is_stable_descending_argsort(T):
    """ivf_rank_frames == torch.argsort(-mask, stable=True): ties by frame index, NaN last (the integer ranking
    the drivers report, FindMasksComparison_I3D_smth.py:216-230)."""
    import ivf_lib as L
    import ivf_search
    gen = torch.Generator().manual_seed(T)
    m = torch.rand(7, T, generator=gen)
    m[1] = (m[1] * 4).round() / 4            # many ties
    m[2] = 0.5                               # all equal
    m[3, ::3] = float('nan')                 # NaN entries (a val == 0 TV gradient poisons the mask, mask.py:97)
    m[4] = torch.linspace(1, 0, T)           # already sorted
    m[5] = torch.linspace(0, 1, T)           # reversed
    mg = m.cuda()
    order = torch.empty(7, T, dtype=torch.int32, device='cuda')
    L.check(L.lib().ivf_rank_frames(L.ptr(mg), 7, T, L.ptr(order), L.stream()))
    ref = torch.argsort(-m, dim=-1, stable=True)
    assert torch.equal(order.cpu().long(), ref)
    assert torch.equal(ivf_search.frame_ranking(mg).cpu(), ref)


def test_init_central_select_matches_reference_rule():
    """ivf_init_central_select == mask.py:134-154 restated with torch: first candidate whose score ratio drops below
    the threshold, else the last one; NaN ratios (orig == full) compare false."""
    import ivf_lib as L
    T, n, B = 16, 7, 9
    gen = torch.Generator().manual_seed(3)
    orig = torch.rand(B, generator=gen) * 0.5 + 0.5
    full = torch.rand(B, generator=gen) * 0.3
    cen = orig[:, None] - (orig - full)[:, None] * torch.sort(torch.rand(B, n, generator=gen), dim=1).values
    cen[0] = orig[0]                 # ratio 0 everywhere -> below the threshold at i = 1
    cen[1] = full[1]                 # ratio 1 everywhere -> never below: last candidate
    full[2] = orig[2]                # 0 / 0 and x / 0: NaN / inf, never below
    thr = 0.5
    ratio = (orig[:, None] - cen) / (orig[:, None] - full[:, None])
    below = ratio < thr
    first = torch.where(below.any(dim=1), below.float().argmax(dim=1), torch.full((B,), n - 1))
    t = torch.arange(T)[None]
    pick = (first + 1)[:, None]
    raw_ref = torch.where((t < pick) | (t >= T - pick), torch.tensor(-5.0), torch.tensor(5.0))
    raw = torch.empty(B, T, device='cuda')
    chosen = torch.empty(B, dtype=torch.int32, device='cuda')
    rat = torch.empty(B, n, device='cuda')
    og, fg, cg = orig.cuda(), full.cuda(), cen.cuda().contiguous()      # (kept alive across the launch)
    L.check(L.lib().ivf_init_central_select(L.ptr(og), L.ptr(fg), L.ptr(cg), B, n, T, thr, L.ptr(raw), L.ptr(chosen),
                                            L.ptr(rat), L.stream()))
    assert torch.equal(chosen.cpu().long(), first + 1)
    assert torch.equal(raw.cpu(), raw_ref)
    assert torch.allclose(rat.cpu(), ratio, rtol=1e-6, atol=0, equal_nan=True)
