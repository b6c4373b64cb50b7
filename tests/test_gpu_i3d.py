"""Whole-I3D parity on the GPU: HIP plan vs the reference model's outputs
(tests/golden/i3d.npz, gradcam.npz, search.npz) at the BASELINE shapes."""
import numpy as np
import pytest
import torch

from conftest import note, ranking_consistent, rel_err, rel_err_elem

pytestmark = pytest.mark.gpu

# arithmetic modes held to the exact-fp32 gates: the fp32 MFMA chain and the 6-pass 3-way bf16 split (24-bit operands)
EXACT = ("fp32", "bf16x6")


@pytest.fixture(scope="module", params=["fp32", "bf16x3", "bf16x6"])
def s16(request):
    """Both arithmetic modes of the convolutions must meet the same gates."""
    import ivf_engine
    import ivf_recipe as R
    eng = ivf_engine.I3DEngine(174, (3, 16, 224, 224), max_batch=2, softmax=True, math=request.param)
    eng.load_state_dict(R.i3d_state_dict(num_classes=174))
    return eng


@pytest.fixture(scope="module", params=["fp32", "bf16x3", "bf16x6"])
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
    assert l2 < (3e-2 if eng.math in EXACT else 5e-2)
    assert rel_err(sample, ref) < (4e-2 if eng.math in EXACT else 1e-1)
    assert abs(np.linalg.norm(dxn.astype(np.float64)) - float(g[f'{tag}_dx_norm'])) < 1e-2 * float(g[f'{tag}_dx_norm'])
    spf = dxn[0].astype(np.float64).sum(axis=(0, 2, 3))
    # per-frame sums cancel to ~1e-6 of the gradient's norm: the most tie-sensitive figure here
    assert rel_err(spf, g[f'{tag}_dx_sum_per_frame']) < (3e-2 if eng.math in EXACT else 6e-2)


def _cpu_module_with_gpu_gates(eng, sd, acts, name, v, flips):
    """Forward of one endpoint (CPU oracle arithmetic) with the ReLU gates of the GPU's activations; `flips` counts
    where the CPU's own pre-activation sign disagrees with the GPU's gate."""
    import ivf_arch as arch
    from oracle import i3d_ref

    def gated(pre, gate):
        flips[0] += int(((pre > 0) != gate).sum())
        flips[1] += gate.numel()
        return pre * gate.to(pre.dtype)

    def stored(t):
        """what the next unit reads back: the fp32 value, or (bf16 activation storage) its RNE to bf16 -- applied
        to the module's INNER buffers, with a straight-through gradient"""
        return t + (t.bfloat16().float() - t).detach() if eng.math == "bf16act" else t

    if name in arch.INCEPTION:
        out_gate = acts[name] > 0
        inner_gate = eng.endpoint(name + '.b12a', 1).cpu() > 0
        c = [sd[f'{name}.{u}.conv3d.weight'].shape[0] for u in ('b0', 'b1a', 'b1b', 'b2a', 'b2b', 'b3b')]
        o1, o2, o3 = c[0], c[0] + c[2], c[0] + c[2] + c[4]
        b0 = gated(i3d_ref.unit3d(v, sd, name + '.b0', relu=False), out_gate[:, :o1])
        t1 = stored(gated(i3d_ref.unit3d(v, sd, name + '.b1a', relu=False), inner_gate[:, :c[1]]))
        b1 = gated(i3d_ref.unit3d(t1, sd, name + '.b1b', relu=False), out_gate[:, o1:o2])
        t2 = stored(gated(i3d_ref.unit3d(v, sd, name + '.b2a', relu=False), inner_gate[:, c[1]:]))
        b2 = gated(i3d_ref.unit3d(t2, sd, name + '.b2b', relu=False), out_gate[:, o2:o3])
        b3 = gated(i3d_ref.unit3d(i3d_ref.maxpool_same(v, (3, 3, 3), (1, 1, 1)), sd, name + '.b3b', relu=False),
                   out_gate[:, o3:])
        return torch.cat([b0, b1, b2, b3], dim=1)
    if name in arch.POOLS:
        k, s = arch.POOLS[name]
        return i3d_ref.maxpool_same(v, k, s)
    stride = (2, 2, 2) if name == 'Conv3d_1a_7x7' else (1, 1, 1)
    return gated(i3d_ref.unit3d(v, sd, name, stride, relu=False), acts[name] > 0)


def _modulewise_backward(eng, sd_np, shape, pool_kernel, gates=None):
    """Strict backward parity: for every endpoint, run the CPU oracle's module on the GPU's own input activation
    with the GPU's own upstream gradient AND the GPU's own ReLU gates, and compare the downstream gradient.
    Identical inputs => identical max-pool ties; identical gates => what is left is the rounding of the arithmetic
    mode, gated element by element with NO outlier allowance.  (Round 2 let the CPU recompute the gates and allowed
    a fraction of outliers; tools/diag_modulewise.py showed those outliers to be single gate flips -- one 3x3x3
    unit's output gate within 3e-8 of zero moves 27 x 192 gradient entries -- so the flips are now counted on their
    own: the CPU's recomputed gates against the GPU's, a handful per half million.)"""
    # element-level threshold (x max |reference gradient| of the tensor): the fp32 MFMA chain and the 6-pass split
    # round like fp32; the 3-pass split carries ~2^-17 per product
    # (measured, profiles/r03_parity_measured.txt: 1.6e-6 / 6.4e-6 element error, 9e-7 / 5.8e-6 L2)
    thr, fwd_tol, l2_tol = (1e-5, 1e-5, 5e-6) if eng.math in EXACT else (5e-5, 1e-4, 3e-5)
    flip_tol = 1e-4
    if gates is not None:
        thr, fwd_tol, l2_tol, flip_tol = gates
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
    flips = [0, 0]      # (disagreeing gates, gates)

    def module(name, v):
        return _cpu_module_with_gpu_gates(eng, sd, acts, name, v, flips)

    # head: feature gradient from the class score
    f = acts['Mixed_5c'].clone().requires_grad_()
    _, out = i3d_ref.head(f, sd, pool_kernel, True)
    out[0, target].backward()
    ref = f.grad * (f > 0).float()
    assert rel_err(grads['Mixed_5c'].numpy(), ref.numpy()) < max(thr, 1e-4)
    stats = []
    for i in range(len(names) - 1, 0, -1):
        src, dst = names[i - 1], names[i]
        v = acts[src].clone().requires_grad_()
        y = module(dst, v)
        assert rel_err(acts[dst].numpy(), y.detach().numpy()) < fwd_tol, dst      # forward, same input, same gates
        (y * grads[dst]).sum().backward()
        ref = v.grad
        if src in arch.INCEPTION or src.startswith('Conv3d'):
            ref = ref * (v.detach() > 0).float()       # the plan stores ReLU-gated gradients
        got = grads[src].numpy()
        refn = ref.numpy()
        scale = np.abs(refn).max()
        err = float(np.max(np.abs(got - refn)) / scale)
        l2 = float(np.linalg.norm((got - refn).astype(np.float64)) / np.linalg.norm(refn.astype(np.float64)))
        stats.append((src, err, l2))
        assert err < thr, (src, err)                   # every element: no outlier allowance
        assert l2 < l2_tol, (src, l2)
    w = max(stats, key=lambda t: t[1])
    note(f"module-wise backward {shape} {eng.math} (GPU gates on both sides): worst element error {w[1]:.2e} x max at {w[0]} "
         f"(gate {thr:g}); worst L2 {max(t[2] for t in stats):.2e}; CPU-recomputed ReLU gates differing from the GPU's: "
         f"{flips[0]} of {flips[1]} ({flips[0] / flips[1]:.1e})")
    assert flips[0] <= flip_tol * flips[1]
    return w[1]


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
    # (fp32: the tile shape does not change the k-order of an fp32 FMA chain; the split modes re-order the sums)
    tol = {"fp32": 1e-6, "bf16x6": 5e-6}.get(s16.math, 2e-5)
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
    assert e_logits < {"fp32": 1e-6, "bf16x6": 5e-6}.get(k32.math, 2e-5)
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
    s16.set_overlap(True)      # (the default)
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


@pytest.fixture(scope="module", params=["fp32", "bf16x3", "bf16x6"])
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


@pytest.mark.parametrize("layer", ['Conv3d_2c_3x3', 'Mixed_3c', 'Mixed_4d'])
def test_gradcam_deep_targets_modulewise(s16, layer):
    """Grad-CAM on a deep target WITHOUT the tie sensitivity: the CPU oracle propagates the class-score gradient from
    the head down to the target module by module on the GPU's own activations, with the GPU's own ReLU gates (identical
    max-pool ties, identical gates), builds the map with the oracle's reduction / resize / normalisation, and must
    agree with the GPU's map to north_star's 1e-3 -- what separates the plain comparison above from 1e-3 is which
    near-ties two fp32 runs break which way, not the kernels' arithmetic."""
    import ivf_arch as arch
    import ivf_recipe as R
    from oracle import gradcam_ref, i3d_ref
    sd = R.to_torch(R.i3d_state_dict(num_classes=174))
    x = torch.from_numpy(R.clip(11))[None]
    cam, probs = s16.gradcam(x.cuda(), None, per_frame=True, layer=layer)
    target = int(torch.argmax(probs[0]))
    acts = {n: s16.endpoint(n, 1).cpu() for n in arch.ENDPOINTS}
    f = acts['Mixed_5c'].clone().requires_grad_()
    _, out = i3d_ref.head(f, sd, (2, 7, 7), True)
    out[0, target].backward()
    g = f.grad * (f > 0).float()
    names = list(arch.ENDPOINTS)
    flips = [0, 0]
    for i in range(len(names) - 1, names.index(layer), -1):
        src, dst = names[i - 1], names[i]
        v = acts[src].clone().requires_grad_()
        (_cpu_module_with_gpu_gates(s16, sd, acts, dst, v, flips) * g).sum().backward()
        g = v.grad
        if src != layer and (src in arch.INCEPTION or src.startswith('Conv3d')):
            g = g * (v.detach() > 0).float()           # below the target the plan gates; the target's gradient stays raw
    ref, _, _ = gradcam_ref.cam_from_activations(acts[layer].numpy()[0], g.numpy()[0], 16, 224, 224, True)
    got = cam[0].cpu().numpy()
    ok = ~np.isnan(ref) & ~np.isnan(got)
    err = float(np.max(np.abs(got[ok] - ref[ok])))
    note(f"gradcam s16 target {layer} {s16.math}, CPU chain on the GPU's activations and gates: max|d| {err:.2e} "
         f"(CPU-recomputed gates differing: {flips[0]} of {flips[1]})")
    assert np.array_equal(np.isnan(got), np.isnan(ref)) and err < 1e-3


def test_bf16_activation_mode_modulewise():
    """The bf16-activation mode module by module (S16): the CPU module on the GPU's own (bf16-valued) activations and
    upstream gradient.  Identical inputs => identical max-pool ties and ReLU gates, so what is left is the mode's own
    rounding: one RNE to bf16 per stored element (2^-9 relative) on top of the 2-pass products."""
    import ivf_engine
    import ivf_recipe as R
    eng = ivf_engine.I3DEngine(174, (3, 16, 224, 224), max_batch=1, softmax=True, math="bf16act")
    sd = R.i3d_state_dict(num_classes=174)
    eng.load_state_dict(sd)
    #        element thr (x max), forward max-rel (one bf16 ulp of the largest element = 2^-8), L2, gate-flip fraction
    _modulewise_backward(eng, sd, (3, 16, 224, 224), (2, 7, 7), gates=(2e-2, 4e-3, 5e-3, 1e-3))


def test_i3d_s32_bf16_activations(golden):
    """BASELINE configs[4] in its stated form: 32-frame 224x224 clips, bf16 ACTIVATION STORAGE (math="bf16act":
    every activation / gradient buffer bf16 in HBM with RNE in the epilogues, weights split hi/lo = 2 MFMA passes,
    fp32 accumulate) against the reference's fp32 outputs (i3d_s32.npz).  A storage precision of 2^-9 per layer is
    NOT the fp32 path: the figures are recorded (profiles/rNN_parity_measured.txt) and gated at what bf16 storage
    can hold; whether north_star's 1e-3 holds is stated in the note, not assumed."""
    import ivf_engine
    import ivf_recipe as R
    g = golden('i3d_s32')
    eng = ivf_engine.I3DEngine(174, (3, 32, 224, 224), max_batch=1, stride_mod_layers="none", last_stride=1,
                               softmax=True, math="bf16act")
    eng.load_state_dict(R.i3d_state_dict(num_classes=174))
    x = torch.from_numpy(R.clip(13, 3, 32, 224, 224))[None].cuda()
    probs, logits = eng.forward(x, want_logits=True)
    e_l = rel_err(logits.cpu().numpy(), g['s32_logits'])
    e_p = rel_err(probs.cpu().numpy(), g['s32_probs'])
    assert int(torch.argmax(probs[0])) == int(g['s32_target'])              # integer output: still bit-exact
    import ivf_arch as arch
    worst_norm = max(abs(float(eng.endpoint(n, 1).double().norm()) / float(g[f's32_norm_{n}']) - 1.0) for n in arch.ENDPOINTS)
    cams = {}
    for tag, pf in (('pf', True), ('glob', False)):
        cam, _ = eng.gradcam(x, None, per_frame=pf)
        got, ref = cam[0].cpu().numpy()[:, ::8, ::8], g[f'gc_{tag}_cam_small']
        assert np.array_equal(np.isnan(got), np.isnan(ref))
        ok = ~np.isnan(ref)
        cams[tag] = float(np.max(np.abs(got[ok] - ref[ok])))
    raw = torch.from_numpy(g['srch_init'])[None].cuda().contiguous()
    traj, _ = eng.search(x, [int(g['s32_target'])], raw, 0.01, 0.02, 3)
    e_t = float(np.max(np.abs(traj[:, 0].cpu().numpy() - g['srch_traj']) / np.abs(g['srch_traj'])))
    holds = max(e_l, e_p, cams['pf'], cams['glob']) < 1e-3
    note(f"configs[4] S32 bf16 activations: logits max-rel {e_l:.2e}, probs max-rel {e_p:.2e}, endpoint norms within "
         f"{worst_norm:.2e}, Grad-CAM max|d| per-frame {cams['pf']:.2e} / global {cams['glob']:.2e} (maps in [0,1]), "
         f"3-iteration trajectory max-rel {e_t:.2e} -> the 1e-3 gate of the fp32 modes "
         f"{'HOLDS' if holds else 'does NOT hold'} with bf16 storage; the 1e-2 trajectory gate "
         f"{'holds' if e_t < 1e-2 else 'does NOT hold'}")
    assert e_l < 2e-2 and e_p < 5e-2 and worst_norm < 1e-2          # bf16 storage: 2^-9 per layer
    assert cams['pf'] < 1e-1 and cams['glob'] < 1e-1
    assert e_t < 1e-2                                                 # north_star's trajectory tolerance
    # the other entry points on bf16 buffers: Grad-CAM on an inner endpoint (ungated target gradient in bf16 storage),
    # the reverse-operator search, the NCTHW input gradient
    cam4, _ = eng.gradcam(x, None, per_frame=False, layer='Mixed_4f')
    ref4, got4 = g.get('none'), cam4[0].cpu().numpy()
    assert got4.shape == (32, 224, 224) and np.isfinite(got4).all() and 0.0 <= got4.min() and got4.max() <= 1.0 + 1e-6
    raw2 = torch.from_numpy(g['srch_init'])[None].cuda().contiguous()
    tr2, _ = eng.search(x, [int(g['s32_target'])], raw2, 0.01, 0.02, 2, mode='reverse')
    assert torch.isfinite(tr2).all()
    _, dx = eng.backward(1, target=[int(g['s32_target'])])
    assert torch.isfinite(dx).all() and float(dx.abs().max()) > 0


def _full_search(eng, g, tag, x, lam1, lam2, N, spread):
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
    # direction of the loss) and the mask itself wander more than the loss: Adam(lr=0.2) amplifies last-bit
    # gradient differences.  How much is NOT taken from what the GPU measures: tests/golden/search_spread.npz
    # holds the same reference search run twice more on the CPU (make_golden.py gen_search_spread: fp64, and
    # fp32 with 4 instead of 8 threads); their distance to the committed run is the noise floor of this search
    # (S16 N=300: final mask 1.9e-2, terms 1.4e-2, loss 6.8e-3 between two fp32 CPU runs; K32: 1e-6).
    sp = spread
    cpu_dmask = max(float(sp[f'{tag}_f64_dmask']), float(sp[f'{tag}_f32t4_dmask']))
    cpu_dterms = max(float(sp[f'{tag}_f64_dterms_rel_max']), float(sp[f'{tag}_f32t4_dterms_rel_max']))
    note(f"full search {tag}: CPU-vs-CPU floor (fp64 / 4-thread fp32 vs the committed 8-thread fp32 run): final mask "
         f"{cpu_dmask:.2e}, terms {cpu_dterms:.2e}")
    assert rel[:, 0].max() < 1e-2 and rel[-1].max() < 1e-2
    assert rel.max() < max(1e-2, 2 * cpu_dterms)              # mid-run terms: within twice what two CPU runs differ by
    mask_tol = max(1e-5, 1.5 * cpu_dmask)                     # final mask: within 1.5x the CPU-vs-CPU distance (floor: fp32 rounding)
    assert dmask < mask_tol
    # integer outputs: snapped mask and frame ranking -- bit-exact wherever the reference mask separates two frames
    # by more than the CPU-vs-CPU noise of that mask (a FIXED constant of the fixture), tie-consistent below it
    assert np.array_equal(final > 0.5, g[f'{tag}_mask'] > 0.5)
    rank = np.argsort(-final, kind='stable')
    sorted_ref = np.sort(g[f'{tag}_mask'])
    tie_tol = 2 * mask_tol
    if np.min(np.diff(sorted_ref)) > tie_tol:                 # the reference separates all frames: bit-exact
        assert np.array_equal(rank, g[f'{tag}_ranking'])
    else:
        assert ranking_consistent(rank, g[f'{tag}_mask'], tie_tol)
    rev = eng.perturbed_forward(x, torch.sigmoid(raw), 'reverse')[0, target]
    assert abs(float(rev) - float(g[f'{tag}_reverse_score'])) < 1e-2 * float(g[f'{tag}_reverse_score'])


def test_full_length_search_s16(s16, golden):
    """N=300, lam 0.01/0.02 (smth:106-119) against the reference harness's 300-iteration run."""
    import ivf_recipe as R
    x = torch.from_numpy(R.clip(21))[None].cuda()
    _full_search(s16, golden('search_long'), 's16', x, 0.01, 0.02, 300, golden('search_spread'))


def test_full_length_search_k32(k32, golden):
    """I3D-KTH, N=100, lam 0.02/0.04 (KTH:105-118)."""
    import ivf_recipe as R
    x = torch.from_numpy(R.clip(23, 3, 32, 120, 160))[None].cuda()
    _full_search(k32, golden('search_long'), 'k32', x, 0.02, 0.04, 100, golden('search_spread'))


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
    # 1e-3 (north_star) where at most one strided max-pool lies between the target and the score.  Further down the
    # plain comparison measures which max-pool near-ties / ReLU zeros two fp32 runs break which way, not arithmetic:
    #  * the reference ITSELF moves by 1e-4 .. 1.3e-3 on these targets between fp32 and fp64 / another thread count
    #    (tests/golden/gradcam_spread.npz, make_golden.py gen_gradcam_spread; recorded in the note below);
    #  * with the ties and gates pinned (test_gradcam_deep_targets_modulewise: CPU chain on the GPU's activations and
    #    gates) the same maps agree to 3.5e-6 (fp32) / 1.7e-4 (6-pass split) / 1.0e-4 (3-pass split) -- THAT test
    #    carries the 1e-3 gate for the deep targets; this one bounds the tie sensitivity (sanity: 1e-2 / 3e-2).
    sp = golden('gradcam_spread')
    floor = max(float(sp[f'{layer}_f64_dmax']), float(sp[f'{layer}_f32t3_dmax']))
    note(f"gradcam s16 target {layer}: the reference's own fp64 / other-thread-count runs differ from its committed map by {floor:.2e}")
    deep = layer not in ('Mixed_4f', 'Mixed_5b', 'Mixed_5c')
    assert ok.any() and err < ((1e-2 if s16.math in EXACT else 3e-2) if deep else 1e-3)
    # the ordinary passes are untouched by the ungated Grad-CAM pass before them
    p2 = s16.forward(x)
    assert torch.equal(p2, probs)
