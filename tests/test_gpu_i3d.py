"""Whole-I3D parity on the GPU: HIP plan vs the reference model's outputs
(tests/golden/i3d.npz, gradcam.npz, search.npz) at the BASELINE shapes."""
import numpy as np
import pytest
import torch

from conftest import note, ranking_consistent, rel_err, rel_err_elem

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=["fp32", "bf16x3"])
def s16(request):
    """Both arithmetic modes of the convolutions must meet the same gates."""
    import ivf_engine
    import ivf_recipe as R
    eng = ivf_engine.I3DEngine(174, (3, 16, 224, 224), max_batch=2, softmax=True, math=request.param)
    eng.load_state_dict(R.i3d_state_dict(num_classes=174))
    return eng


@pytest.fixture(scope="module", params=["fp32", "bf16x3"])
def k32(request):
    import ivf_engine
    import ivf_recipe as R
    eng = ivf_engine.I3DEngine(6, (3, 32, 120, 160), max_batch=1, head_hw=(4, 5), head_time_base=4,
                               softmax=True, math=request.param)
    eng.load_state_dict(R.i3d_state_dict(num_classes=6, tag='i3d_kth'))
    return eng


def _check_forward_backward(eng, tag, shape, g, clip_id=7):
    import ivf_recipe as R
    x = torch.from_numpy(R.clip(clip_id, *shape))[None].cuda()
    probs, logits = eng.forward(x, want_logits=True)
    assert rel_err(logits.cpu().numpy(), g[f'{tag}_logits']) < 1e-3        # north_star: 1e-3 relative fp32
    assert rel_err(probs.cpu().numpy(), g[f'{tag}_probs']) < 1e-3
    # element-wise as well: every logit against its own magnitude (floor = 1 % of the largest: a logit
    # that happens to sit near zero has no meaningful relative error), every probability above 1e-6 --
    # small entries are not hidden behind the largest one
    e_l = rel_err_elem(logits.cpu().numpy(), g[f'{tag}_logits'], 1e-2 * np.abs(g[f'{tag}_logits']).max())
    e_p = rel_err_elem(probs.cpu().numpy(), g[f'{tag}_probs'], 1e-6)
    note(f"{tag} {eng.math}: logits max-rel {rel_err(logits.cpu().numpy(), g[f'{tag}_logits']):.2e} "
         f"elementwise {e_l:.2e}; probs elementwise {e_p:.2e}")
    assert e_l < 1e-3 and e_p < 1e-3
    assert int(torch.argmax(probs[0])) == int(g[f'{tag}_target'])          # integer output: bit-exact
    import ivf_arch as arch
    for n in arch.ENDPOINTS:
        a = eng.endpoint(n, 1)
        ref = float(g[f'{tag}_norm_{n}'])
        assert abs(float(a.double().norm()) - ref) < 1e-4 * ref, n
    feat = eng.endpoint('Mixed_5c', 1).cpu().numpy().ravel()
    assert list(eng.endpoint('Mixed_5c', 1).shape) == g[f'{tag}_feat_shape'].tolist()
    assert rel_err(feat[g[f'{tag}_feat_idx']], g[f'{tag}_feat_val']) < 1e-3
    score, dx = eng.backward(1, target=[int(g[f'{tag}_target'])])
    assert abs(float(score) - float(g[f'{tag}_probs'][0, int(g[f'{tag}_target'])])) < 1e-5
    # d(score)/d(input) through the WHOLE chain.  The net is piecewise linear with
    # discontinuous derivatives at max-pool ties and ReLU zeros; chance near-ties
    # (two window entries equal to 1 ulp) are resolved by rounding noise, differently
    # on any two fp32 implementations (measured: the torch-CPU reference on the build
    # box vs the same reference on the GPU box's host differ by the same 1-2 %).  So
    # the whole-chain gate is loose; the strict gate is the module-wise test below,
    # which feeds both sides the same activations.
    dxn = dx.cpu().numpy()
    sample, ref = dxn.ravel()[g[f'{tag}_dx_idx']].astype(np.float64), g[f'{tag}_dx_val'].astype(np.float64)
    l2 = np.linalg.norm(sample - ref) / np.linalg.norm(ref)
    note(f"whole-chain dx {tag} {eng.math}: sampled L2 {l2:.2e}, max {rel_err(sample, ref):.2e}")
    # measured (profiles/r02_parity_measured.txt): CPU vs CPU 0.3 % L2 / 1.0 % max; exact-fp32 MFMA 0.6-1.1 % /
    # 1.3-1.7 %; split-bf16 2.3-3.1 % / 2.8-7.7 % (its ~1e-5 arithmetic noise flips more near-ties)
    assert l2 < (3e-2 if eng.math == "fp32" else 5e-2)
    assert rel_err(sample, ref) < (4e-2 if eng.math == "fp32" else 1e-1)
    assert abs(np.linalg.norm(dxn.astype(np.float64)) - float(g[f'{tag}_dx_norm'])) < 1e-2 * float(g[f'{tag}_dx_norm'])
    spf = dxn[0].astype(np.float64).sum(axis=(0, 2, 3))
    # per-frame sums cancel to ~1e-6 of the gradient's norm: the most tie-sensitive figure here
    assert rel_err(spf, g[f'{tag}_dx_sum_per_frame']) < (3e-2 if eng.math == "fp32" else 6e-2)


def _modulewise_backward(eng, sd_np, shape, pool_kernel):
    # element-level threshold: fp32 MFMA is an exact fp32 FMA chain; the split-bf16 mode
    # carries ~2^-17 per product
    thr, med_tol, frac_tol = (1e-4, 1e-6, 2e-3) if eng.math == "fp32" else (1e-3, 5e-6, 0.02)
    """Strict backward parity: for every endpoint, run the CPU oracle's module on the
    GPU's own input activation with the GPU's own upstream gradient and compare the
    downstream gradient.  Identical inputs => identical ties/gates => fp32 rounding only."""
    import ivf_arch as arch
    import ivf_recipe as R
    from oracle import i3d_ref
    sd = R.to_torch(sd_np)
    x = torch.from_numpy(R.clip(9, *shape))[None]
    probs = eng.forward(x.cuda())
    target = int(torch.argmax(probs[0]))
    _, dx = eng.backward(1, target=[target])
    names = ['input'] + list(arch.ENDPOINTS)
    acts = {n: eng.endpoint(n, 1).cpu() for n in arch.ENDPOINTS}
    acts['input'] = x
    grads = {n: eng.endpoint(n + ':grad', 1).cpu() for n in arch.ENDPOINTS}
    grads['input'] = dx.cpu()

    def module(name, v):
        if name in arch.INCEPTION:
            return i3d_ref.inception(v, sd, name)
        if name in arch.POOLS:
            k, s = arch.POOLS[name]
            return i3d_ref.maxpool_same(v, k, s)
        stride = (2, 2, 2) if name == 'Conv3d_1a_7x7' else (1, 1, 1)
        return i3d_ref.unit3d(v, sd, name, stride)

    # head: feature gradient from the class score
    f = acts['Mixed_5c'].clone().requires_grad_()
    _, out = i3d_ref.head(f, sd, pool_kernel, True)
    out[0, target].backward()
    ref = f.grad * (f > 0).float()
    assert rel_err(grads['Mixed_5c'].numpy(), ref.numpy()) < thr
    worst = 0.0
    stats = []
    for i in range(len(names) - 1, 0, -1):
        src, dst = names[i - 1], names[i]
        v = acts[src].clone().requires_grad_()
        y = module(dst, v)
        assert rel_err(acts[dst].numpy(), y.detach().numpy()) < thr / 10, dst      # forward, same input
        (y * grads[dst]).sum().backward()
        ref = v.grad
        if src in arch.INCEPTION or src.startswith('Conv3d'):
            ref = ref * (v.detach() > 0).float()       # the plan stores ReLU-gated gradients
        got = grads[src].numpy()
        refn = ref.numpy()
        scale = np.abs(refn).max()
        bad = np.abs(got - refn) > thr * scale
        # The CPU side recomputes the module's ReLU from the same input, so a gate whose
        # pre-activation is within rounding of 0 may flip; one flip moves up to
        # taps x Cin downstream entries.  Allow a small fraction of outliers, never a
        # systematic error: the median must sit at fp32 rounding and the L2 error stay small.
        # Measured (profiles/r02_parity_measured.txt): outlier fraction <= 0.0006 fp32 / 0.0063
        # split-bf16, L2 <= 5.8e-4 / 4.1e-3; the gates sit 3x above that.
        assert bad.mean() < frac_tol, (src, float(bad.mean()))
        med = np.median(np.abs(got - refn)) / scale
        assert med < med_tol, (src, med)
        l2 = np.linalg.norm((got - refn).astype(np.float64)) / np.linalg.norm(refn.astype(np.float64))
        assert l2 < (2e-3 if eng.math == "fp32" else 1e-2), (src, l2)
        worst = max(worst, float(bad.mean()))
        stats.append((src, float(bad.mean()), float(med), float(l2)))
    w = max(stats, key=lambda t: t[1])
    note(f"module-wise backward {shape} {eng.math}: worst outlier fraction {w[1]:.4f} at {w[0]} (thr {thr:g}*scale); "
         f"worst median {max(t[2] for t in stats):.2e}; worst L2 {max(t[3] for t in stats):.2e}")
    return worst


def test_i3d_s16_backward_modulewise(s16):
    import ivf_recipe as R
    _modulewise_backward(s16, R.i3d_state_dict(num_classes=174), (3, 16, 224, 224), (2, 7, 7))


def test_i3d_k32_backward_modulewise(k32):
    import ivf_recipe as R
    _modulewise_backward(k32, R.i3d_state_dict(num_classes=6, tag='i3d_kth'), (3, 32, 120, 160), (4, 4, 5))


def test_i3d_s16_forward_backward(s16, golden):
    _check_forward_backward(s16, 's16', (3, 16, 224, 224), golden('i3d'))


def test_i3d_k32_forward_backward(k32, golden):
    _check_forward_backward(k32, 'k32', (3, 32, 120, 160), golden('i3d'))


def test_whole_chain_dx_cpu_vs_cpu(golden):
    """Evidence behind the loose whole-chain d(score)/d(input) gate above: the SAME torch-CPU oracle, run
    on this box's host cores, against the fixture the build container's CPU produced.  Any difference is
    two fp32 CPU implementations routing max-pool near-ties / ReLU zeros differently; the figure is
    recorded in profiles/rNN_parity_measured.txt and must itself respect the gate the GPU is held to."""
    import ivf_recipe as R
    from oracle import i3d_ref
    g = golden('i3d')
    sd = R.to_torch(R.i3d_state_dict(num_classes=174))
    x = torch.from_numpy(R.clip(7))[None].requires_grad_()
    probs = i3d_ref.forward(x, sd)
    t = int(g['s16_target'])
    probs[0, t].backward()
    dx = x.grad.numpy()
    sample, ref = dx.ravel()[g['s16_dx_idx']].astype(np.float64), g['s16_dx_val'].astype(np.float64)
    l2 = np.linalg.norm(sample - ref) / np.linalg.norm(ref)
    spf = rel_err(dx[0].astype(np.float64).sum(axis=(0, 2, 3)), g['s16_dx_sum_per_frame'])
    note(f"whole-chain dx, torch-CPU on this host ({torch.get_num_threads()} threads) vs torch-CPU on the build "
         f"container: sampled L2 {l2:.2e}, max {rel_err(sample, ref):.2e}, per-frame sums {spf:.2e}")
    assert l2 < 3e-2 and rel_err(sample, ref) < 8e-2
    # the same for Grad-CAM maps on target layers several max-pools below the score (their channel weights
    # are position sums of that gradient): host CPU vs build-container CPU
    from oracle import gradcam_ref
    gl = golden('gradcam_layers')
    xg = torch.from_numpy(R.clip(11))[None]
    for layer in ('Mixed_3c', 'Mixed_4c', 'Mixed_4f'):
        cam, _, _ = gradcam_ref.gradcam_i3d(xg, sd, None, layer=layer)
        got, want = cam[:, ::8, ::8], gl[f'{layer}_cam_small']
        ok = ~np.isnan(want)
        note(f"gradcam target {layer}, torch-CPU on this host vs torch-CPU on the build container: "
             f"max|d| {np.max(np.abs(got[ok] - want[ok])):.2e}")


def test_batch_rows_independent(s16):
    """eval-mode BN => rows are independent: a clip's result must not
    depend on its batch neighbours, bit for bit."""
    import ivf_recipe as R
    a = torch.from_numpy(R.clip(7))[None].cuda()
    b = torch.from_numpy(R.clip(8))[None].cuda()
    pa = s16.forward(a).clone()
    pab = s16.forward(torch.cat([b, a]))
    assert torch.equal(pab[1], pa[0])


def test_tuned_plan_matches_builtin_choice(s16):
    """A plan whose kernels were picked by the autotuner on this box against the plan the other parity tests run
    (built-in choice): same logits to rounding, same input gradient up to the arg-max ties the last bits can flip."""
    import ivf_engine
    import ivf_recipe as R
    tuned = ivf_engine.I3DEngine(174, (3, 16, 224, 224), max_batch=2, softmax=True, math=s16.math)
    tuned.load_state_dict(R.i3d_state_dict(num_classes=174), autotune=True)
    assert tuned.get_tuning() != s16.get_tuning()      # the tuner did choose something
    x = torch.from_numpy(np.stack([R.clip(7), R.clip(8)])).cuda()
    tgt = torch.tensor([3, 11], dtype=torch.int32)
    res = []
    for eng in (s16, tuned):
        probs, logits = eng.forward(x, want_logits=True)
        score, dx = eng.backward(2, target=tgt)
        res.append((logits.clone(), probs.clone(), score.clone(), dx.clone()))
    tol = 1e-6 if s16.math == "fp32" else 2e-5
    e_logits = rel_err(res[1][0].cpu().numpy(), res[0][0].cpu().numpy())
    d0, d1 = res[0][3].double(), res[1][3].double()
    e_dx = float((d1 - d0).norm() / d0.norm())
    note(f"tuned vs built-in plan, {s16.math}: logits max-rel {e_logits:.2e}, dx L2 {e_dx:.2e}")
    assert e_logits < tol
    assert rel_err(res[1][2].cpu().numpy(), res[0][2].cpu().numpy()) < 10 * tol
    assert e_dx < 5e-2      # (1.5e-2 measured: max-pool near-ties rerouted by last-bit differences, as in test_whole_chain_*)


def test_tuned_plan_matches_builtin_choice_k32(k32):
    """The same on the KTH geometry (120 x 160 frames, odd map sizes: partial boxes in every kernel family)."""
    import ivf_engine
    import ivf_recipe as R
    tuned = ivf_engine.I3DEngine(6, (3, 32, 120, 160), max_batch=1, head_hw=(4, 5), head_time_base=4, softmax=True,
                                 math=k32.math)
    tuned.load_state_dict(R.i3d_state_dict(num_classes=6, tag='i3d_kth'), autotune=True)
    x = torch.from_numpy(R.clip(5, 3, 32, 120, 160))[None].cuda()
    res = []
    for eng in (k32, tuned):
        probs, logits = eng.forward(x, want_logits=True)
        score, dx = eng.backward(1, target=torch.tensor([2], dtype=torch.int32))
        res.append((logits.clone(), dx.clone()))
    e_logits = rel_err(res[1][0].cpu().numpy(), res[0][0].cpu().numpy())
    d0, d1 = res[0][1].double(), res[1][1].double()
    e_dx = float((d1 - d0).norm() / d0.norm())
    note(f"tuned vs built-in plan (K32), {k32.math}: logits max-rel {e_logits:.2e}, dx L2 {e_dx:.2e}")
    assert e_logits < (1e-6 if k32.math == "fp32" else 2e-5)
    assert e_dx < 5e-2


def test_side_stream_overlap_is_bit_identical(s16):
    """ivf_i3d_set_overlap: the pool / b0 / b3b branch of every Inception module on a side stream (fork/join per
    module) must not change a bit of the probabilities, the scores or the input gradient."""
    import ivf_recipe as R
    x = torch.from_numpy(np.stack([R.clip(7), R.clip(8)])).cuda()
    tgt = torch.tensor([3, 11], dtype=torch.int32)
    out = []
    for on in (False, True, True):
        s16.set_overlap(on)
        probs = s16.forward(x).clone()
        score, dx = s16.backward(2, target=tgt)
        out.append((probs, score.clone(), dx.clone()))
    s16.set_overlap(False)
    for k in (1, 2):
        for a, b in zip(out[0], out[k]):
            assert torch.equal(a, b)


def test_gradcam_vs_reference(s16, golden):
    import ivf_recipe as R
    g = golden('gradcam')
    x = torch.from_numpy(R.clip(11))[None].cuda()
    for tag, pf in (('pf', True), ('glob', False)):
        cam, probs = s16.gradcam(x, None, per_frame=pf)
        cam = cam[0].cpu().numpy()
        assert cam.shape == (16, 224, 224) and cam.dtype == np.float32
        assert rel_err(probs.cpu().numpy(), g[f'{tag}_output']) < 1e-3
        ref = g[f'{tag}_cam_small']
        got = cam[:, ::8, ::8]
        assert np.array_equal(np.isnan(got), np.isnan(ref))
        ok = ~np.isnan(ref)
        assert ok.any()
        note(f"gradcam s16 {tag} {s16.math}: max|d| {np.max(np.abs(got[ok] - ref[ok])):.2e} (maps in [0,1])")
        assert np.max(np.abs(got[ok] - ref[ok])) < 1e-3          # north_star: Grad-CAM maps within 1e-3
        rows = cam[[0, 7, 8, 15]][:, [0, 100, 223]]
        okr = ~np.isnan(g[f'{tag}_cam_rows'])
        assert np.max(np.abs(rows[okr] - g[f'{tag}_cam_rows'][okr])) < 1e-3
        # Grad-CAM L1 vs reference (BASELINE metric), on the sub-sampled map
        l1 = float(np.mean(np.abs(got[ok] - ref[ok])))
        assert l1 < 5e-4
    cam5, _ = s16.gradcam(x, [5], per_frame=True)
    got = cam5[0].cpu().numpy()[:, ::8, ::8]
    ref = g['idx5_cam_small']
    assert np.array_equal(np.isnan(got), np.isnan(ref))
    ok = ~np.isnan(ref)       # class 5 may have an all-negative CAM: 0/0 -> NaN on both sides
    if ok.any():
        assert np.max(np.abs(got[ok] - ref[ok])) < 1e-3


def test_search_trajectory_vs_reference(s16, golden):
    """12 iterations of the hot loop vs the reference's mask.py + model + torch Adam."""
    import ivf_recipe as R
    g = golden('search')
    x = torch.from_numpy(R.clip(21))[None].cuda()
    probs = s16.forward(x)
    target = int(torch.argmax(probs[0]))
    assert target == int(g['s16_target'])
    assert abs(float(probs[0, target]) - float(g['s16_orig'])) < 1e-3 * float(g['s16_orig'])
    # init_mask scores (mask.py:121-154)
    T = 16
    ones = torch.ones(1, T, device='cuda')
    full = s16.perturbed_forward(x, ones, 'freeze')[0, target]
    assert abs(float(full) - float(g['s16_full'])) < 1e-3 * float(g['s16_full'])
    for i, ref in enumerate(g['s16_central'], start=1):
        m = torch.ones(1, T, device='cuda')
        m[0, :i] = 0
        m[0, T - i:] = 0
        c = s16.perturbed_forward(x, m, 'freeze')[0, target]
        assert abs(float(c) - float(ref)) < 1e-3 * float(ref)
    raw = torch.from_numpy(g['s16_init'])[None].cuda().contiguous()
    traj, state = s16.search(x, [target], raw, 0.01, 0.02, 12)
    traj = traj[:, 0].cpu().numpy()
    ref = g['s16_traj']
    assert np.max(np.abs(traj - ref) / np.abs(ref)) < 1e-2        # north_star: loss trajectory within 1e-2
    assert np.max(np.abs(traj[:, 3] - ref[:, 3]) / ref[:, 3]) < 2e-3
    final = torch.sigmoid(raw)[0].cpu().numpy()
    # Mask values after 12 Adam(lr 0.2) steps: measured 2e-4 .. 2.0e-3 depending on which kernel variants the
    # tuner picked on the box (summation order differs in the last bits; one frame whose gradient passes near
    # zero in an early iteration carries the spread).  The loss trajectory above is the north_star gate.
    dm = float(np.max(np.abs(final - g['s16_mask'])))
    note(f"12-iteration search, {s16.math}: max|mask - reference| {dm:.2e} (gate 5e-3)")
    assert dm < 5e-3
    # integer outputs bit-exact: snapped mask, ranking
    assert np.array_equal(final > 0.5, g['s16_mask'] > 0.5)
    # the frame ranking is exact where the reference mask separates the frames by more than the measured spread of
    # the mask values themselves (2 x the gate above; frames 4 and 5 of this clip sit 6e-4 apart)
    assert ranking_consistent(np.argsort(-final, kind='stable'), g['s16_mask'], 1e-2)
    rev = s16.perturbed_forward(x, torch.sigmoid(raw), 'reverse')[0, target]
    assert abs(float(rev) - float(g['s16_reverse_score'])) < 2e-3 * float(g['s16_reverse_score'])


@pytest.fixture(scope="module", params=["fp32", "bf16x3"])
def s32(request):
    """BASELINE configs[4] geometry: 32-frame 224x224 clips, stride_mod_layers="none" (no endpoint
    matches, head window [4,7,7])."""
    import ivf_engine
    import ivf_recipe as R
    eng = ivf_engine.I3DEngine(174, (3, 32, 224, 224), max_batch=1, stride_mod_layers="none", last_stride=1,
                               softmax=True, math=request.param)
    eng.load_state_dict(R.i3d_state_dict(num_classes=174))
    return eng


def test_i3d_s32_refuses_the_inconsistent_head():
    """With stride_mod_layers="" the reference's head AvgPool3d([2,7,7]) leaves a [B,K,3] output that
    it squeezes inconsistently: the plan refuses that loudly."""
    import ivf_engine
    import ivf_lib as L
    with pytest.raises(L.IvfError):
        ivf_engine.I3DEngine(174, (3, 32, 224, 224), max_batch=1, stride_mod_layers="")


def test_i3d_s32_forward_backward(s32, golden):
    """vs the reference model I3D_doubled.Model(174, last_stride=1, stride_mod_layers="none")."""
    _check_forward_backward(s32, 's32', (3, 32, 224, 224), golden('i3d_s32'), clip_id=13)


def test_i3d_s32_gradcam_and_search(s32, golden):
    import ivf_recipe as R
    g = golden('i3d_s32')
    x = torch.from_numpy(R.clip(13, 3, 32, 224, 224))[None].cuda()
    for tag, pf in (('pf', True), ('glob', False)):
        cam, probs = s32.gradcam(x, None, per_frame=pf)
        cam = cam[0].cpu().numpy()
        assert list(cam.shape) == g[f'gc_{tag}_cam_shape'].tolist() == [32, 224, 224]
        assert rel_err(probs.cpu().numpy(), g[f'gc_{tag}_output']) < 1e-3
        ref, got = g[f'gc_{tag}_cam_small'], cam[:, ::8, ::8]
        assert np.array_equal(np.isnan(got), np.isnan(ref))
        ok = ~np.isnan(ref)
        note(f"gradcam s32 {tag} {s32.math}: max|d| {np.max(np.abs(got[ok] - ref[ok])):.2e}")
        assert ok.any() and np.max(np.abs(got[ok] - ref[ok])) < 1e-3
    target = int(g['s32_target'])
    raw = torch.from_numpy(g['srch_init'])[None].cuda().contiguous()
    traj, _ = s32.search(x, [target], raw, 0.01, 0.02, 3)
    got, ref = traj[:, 0].cpu().numpy(), g['srch_traj']
    assert np.max(np.abs(got - ref) / np.abs(ref)) < 1e-2


def _full_search(eng, g, tag, x, lam1, lam2, N):
    """The reference's FULL search length: gates of north_star at EVERY iteration including the last."""
    target = int(g[f'{tag}_target'])
    probs = eng.forward(x)
    assert int(torch.argmax(probs[0])) == target
    raw = torch.from_numpy(g[f'{tag}_init'])[None].cuda().contiguous()
    traj, _ = eng.search(x, [target], raw, lam1, lam2, N)
    traj = traj[:, 0].cpu().numpy()
    ref = g[f'{tag}_traj']
    assert ref.shape == (N, 4)
    rel = np.abs(traj - ref) / np.abs(ref[:, :1])           # every term against the loss it is part of
    final = torch.sigmoid(raw)[0].cpu().numpy()
    dmask = float(np.max(np.abs(final - g[f'{tag}_mask'])))
    note(f"full search {tag} {eng.math} N={N}: loss rel err max {rel[:, 0].max():.2e} last {rel[-1, 0]:.2e}; "
         f"terms max {rel.max():.2e}; score rel last {abs(traj[-1, 3] - ref[-1, 3]) / ref[-1, 3]:.2e}; "
         f"final mask max|d| {dmask:.2e}")
    # north_star: the mask-LOSS trajectory within 1e-2 after N iterations -- gated at every iteration and,
    # for all four terms, at the last one.  Mid-run the individual terms (l1 vs score trade along a flat
    # direction of the loss) and the mask itself wander more than the loss: Adam(lr=0.2) amplifies
    # last-bit gradient differences, and the EXACT-fp32 mode shows the same spread as split-bf16
    # (profiles/r02_parity_measured.txt), so those are sanity-bounded only.
    assert rel[:, 0].max() < 1e-2 and rel[-1].max() < 1e-2
    assert rel.max() < 3e-2
    assert dmask < 5e-2
    # integer outputs: snapped mask and frame ranking
    assert np.array_equal(final > 0.5, g[f'{tag}_mask'] > 0.5)
    rank = np.argsort(-final, kind='stable')
    sorted_ref = np.sort(g[f'{tag}_mask'])
    if np.min(np.diff(sorted_ref)) > 2 * dmask:               # the reference separates all frames: bit-exact
        assert np.array_equal(rank, g[f'{tag}_ranking'])
    else:                                                     # exact ties in the reference mask itself
        assert ranking_consistent(rank, g[f'{tag}_mask'], 2 * dmask + 1e-7)
    rev = eng.perturbed_forward(x, torch.sigmoid(raw), 'reverse')[0, target]
    assert abs(float(rev) - float(g[f'{tag}_reverse_score'])) < 1e-2 * float(g[f'{tag}_reverse_score'])


def test_full_length_search_s16(s16, golden):
    """N=300, lam 0.01/0.02 (smth:106-119) against the reference harness's 300-iteration run."""
    import ivf_recipe as R
    x = torch.from_numpy(R.clip(21))[None].cuda()
    _full_search(s16, golden('search_long'), 's16', x, 0.01, 0.02, 300)


def test_full_length_search_k32(k32, golden):
    """I3D-KTH, N=100, lam 0.02/0.04 (KTH:105-118)."""
    import ivf_recipe as R
    x = torch.from_numpy(R.clip(23, 3, 32, 120, 160))[None].cuda()
    _full_search(k32, golden('search_long'), 'k32', x, 0.02, 0.04, 100)


def test_gradcam_k32_vs_reference(k32, golden):
    """GradCamVideo as the KTH driver calls it (KTH:315-327): Mixed_5c [1,1024,4,4,5] -> [32,120,160]."""
    import ivf_recipe as R
    g = golden('gradcam_k32')
    x = torch.from_numpy(R.clip(11, 3, 32, 120, 160))[None].cuda()
    for tag, pf in (('pf', True), ('glob', False)):
        cam, probs = k32.gradcam(x, None, per_frame=pf, out_hw=(120, 160))
        cam = cam[0].cpu().numpy()
        assert list(cam.shape) == g[f'{tag}_cam_shape'].tolist() == [32, 120, 160]
        assert rel_err(probs.cpu().numpy(), g[f'{tag}_output']) < 1e-3
        ref, got = g[f'{tag}_cam_small'], cam[:, ::4, ::4]
        assert np.array_equal(np.isnan(got), np.isnan(ref))
        ok = ~np.isnan(ref)
        note(f"gradcam k32 {tag} {k32.math}: max|d| {np.max(np.abs(got[ok] - ref[ok])):.2e}")
        assert ok.any() and np.max(np.abs(got[ok] - ref[ok])) < 1e-3
    cam3, _ = k32.gradcam(x, [3], per_frame=False, out_hw=(120, 160))
    got, ref = cam3[0].cpu().numpy()[:, ::4, ::4], g['idx3_cam_small']
    assert np.array_equal(np.isnan(got), np.isnan(ref))
    ok = ~np.isnan(ref)
    if ok.any():
        assert np.max(np.abs(got[ok] - ref[ok])) < 1e-3


def test_reverse_mask_search_s16(s16, golden):
    """temporalMaskType='reverse' (smth:121,202): init_mask scores the fully FROZEN clip plus
    reverse-perturbed central masks, the loop optimises through the reverse operator."""
    import ivf_recipe as R
    import ivf_search
    g = golden('search_reverse')
    x = torch.from_numpy(R.clip(21))[None].cuda()
    target = int(g['s16_target'])
    probs = s16.forward(x)
    tgt = torch.tensor([target], dtype=torch.int32, device='cuda')
    raw, info = ivf_search.init_masks_central(s16, x, tgt, probs[0, target][None], 0.9, 'reverse')
    assert abs(float(info['full'][0]) - float(g['s16_full'])) < 1e-3 * float(g['s16_full'])
    assert abs(float(info['central'][0, 0]) - float(g['s16_central'][0])) < 1e-3 * float(g['s16_central'][0])
    assert np.array_equal(raw[0].cpu().numpy(), g['s16_init'])
    traj, _ = s16.search(x, [target], raw, 0.01, 0.02, 8, mode='reverse')
    got, ref = traj[:, 0].cpu().numpy(), g['s16_traj']
    assert np.max(np.abs(got - ref) / np.abs(ref)) < 1e-2
    assert np.max(np.abs(torch.sigmoid(raw)[0].cpu().numpy() - g['s16_mask'])) < 2e-3
    with pytest.raises(UnboundLocalError):
        s16.search(x, [target], raw, 0.01, 0.02, 1, mode='blur')     # as mask.py:57 fails


@pytest.mark.parametrize("layer", ['Conv3d_2c_3x3', 'MaxPool3d_3a_3x3', 'Mixed_3c', 'Mixed_4c', 'Mixed_4d', 'Mixed_4e',
                                   'Mixed_4f', 'Mixed_5b'])
def test_gradcam_other_target_layers(s16, layer, golden):
    """GradCamVideo on any endpoint (grad-cam.py:23-54) vs the reference's own output: conv endpoint, pool
    endpoint, Inception endpoints feeding a (gated) max-pool and feeding the next module."""
    import ivf_recipe as R
    g = golden('gradcam_layers')
    x = torch.from_numpy(R.clip(11))[None].cuda()
    cam, probs = s16.gradcam(x, None, per_frame=True, layer=layer)
    cam = cam[0].cpu().numpy()
    assert list(cam.shape) == g[f'{layer}_cam_shape'].tolist() == [16, 224, 224]
    assert rel_err(probs.cpu().numpy(), g[f'{layer}_output']) < 1e-3
    ref, got = g[f'{layer}_cam_small'], cam[:, ::8, ::8]
    assert np.array_equal(np.isnan(got), np.isnan(ref))
    ok = ~np.isnan(ref)
    err = float(np.max(np.abs(got[ok] - ref[ok])))
    note(f"gradcam s16 target {layer} {s16.math}: max|d| {err:.2e}, mean|d| {np.mean(np.abs(got[ok] - ref[ok])):.2e}")
    # 1e-3 (north_star) where at most one strided max-pool lies between the target and the score; further
    # down the map inherits the whole-chain gradient's sensitivity to max-pool near-ties (see
    # test_whole_chain_dx_cpu_vs_cpu, which records the same figure between two CPUs)
    deep = layer not in ('Mixed_4f', 'Mixed_5b', 'Mixed_5c')
    assert ok.any() and err < ((1e-2 if s16.math == "fp32" else 3e-2) if deep else 1e-3)
    # the ordinary passes are untouched by the ungated Grad-CAM pass before them
    p2 = s16.forward(x)
    assert torch.equal(p2, probs)
