"""ConvLSTM (CLSTM_4) HIP plan vs the reference model's own outputs
(tests/golden/clstm.npz, search.npz) at the KTH shapes [B,C,32,120,160]."""
import numpy as np
import pytest
import torch

from conftest import note, ranking_consistent, rel_err

pytestmark = pytest.mark.gpu


def _engine(C, B=2, softmax=True):
    import ivf_engine
    import ivf_recipe as R
    eng = ivf_engine.CLSTMEngine(6, (C, 32, 120, 160), max_batch=B, hidden=4, layers=2, kernel=5, stride=2,
                                 softmax=softmax)
    eng.load_state_dict(R.clstm_state_dict(channels=C, tag=f'clstm{C}'))
    return eng


@pytest.mark.parametrize("C", [1, 3])
def test_clstm_forward_backward(C, golden):
    import ivf_recipe as R
    g = golden('clstm')
    eng = _engine(C)
    x = torch.from_numpy(np.stack([R.clip(3, C, 32, 120, 160), R.clip(4, C, 32, 120, 160)]) / 255.0).float().cuda()
    probs, logits = eng.forward(x, want_logits=True)
    assert rel_err(logits.cpu().numpy(), g[f'c{C}_logits']) < 1e-3
    assert rel_err(probs.cpu().numpy(), g[f'c{C}_probs']) < 1e-3
    # upstream gradient of y[0,2] + y[1,4]
    dout = torch.zeros(2, 6, device='cuda')
    dout[0, 2] = 1
    dout[1, 4] = 1
    _, dx = eng.backward(2, dout=dout)
    dxn = dx.cpu().numpy()
    assert rel_err(dxn.ravel()[g[f'c{C}_dx_idx']], g[f'c{C}_dx_val']) < 2e-3
    assert abs(np.linalg.norm(dxn.astype(np.float64)) - float(g[f'c{C}_dx_norm'])) < 1e-3 * float(g[f'c{C}_dx_norm'])
    assert rel_err(dxn.astype(np.float64).sum(axis=(1, 3, 4)), g[f'c{C}_dx_sum_per_frame']) < 2e-3
    # one-hot target path gives the same as the explicit dout
    s, dx2 = eng.backward(2, target=[2, 4])
    assert torch.equal(dx2, dx)
    assert abs(float(s[0]) - float(g[f'c{C}_probs'][0, 2])) < 1e-5


def test_clstm_search_trajectory(golden):
    """30 iterations of the hot loop with the KTH lambdas vs the reference harness."""
    import ivf_recipe as R
    import ivf_search
    g = golden('search')
    eng = _engine(1, B=1)
    x = torch.from_numpy(R.clip(3, 1, 32, 120, 160) / 255.0).float()[None].cuda()
    probs = eng.forward(x)
    target = int(torch.argmax(probs[0]))
    assert target == int(g['c1_target'])
    tgt = torch.tensor([target], dtype=torch.int32, device='cuda')
    raw, info = ivf_search.init_masks_central(eng, x, tgt, probs[0, target][None], 0.9, 'freeze')
    assert abs(float(info['full'][0]) - float(g['c1_full'])) < 1e-5
    assert np.array_equal(raw[0].cpu().numpy(), g['c1_init'])          # same central mask chosen
    traj, _ = eng.search(x, [target], raw, 0.02, 0.04, 30)
    traj = traj[:, 0].cpu().numpy()
    assert np.max(np.abs(traj - g['c1_traj']) / np.abs(g['c1_traj'])) < 1e-2     # north_star gate
    assert np.max(np.abs(traj - g['c1_traj']) / np.abs(g['c1_traj'])) < 1e-3     # what we actually get
    final = torch.sigmoid(raw)[0].cpu().numpy()
    assert np.max(np.abs(final - g["c1_mask"])) < 1e-2     # Adam(lr=0.2) amplifies last-bit gradient differences
    assert np.array_equal(np.argsort(-final, kind='stable'), np.argsort(-g['c1_mask'], kind='stable'))
    rev = eng.perturbed_forward(x, torch.sigmoid(raw), 'reverse')[0, target]
    assert abs(float(rev) - float(g['c1_reverse_score'])) < 1e-4


def test_clstm_dropin_model(golden):
    import ivf_recipe as R
    from models import CLSTM_4
    g = golden('clstm')
    m = CLSTM_4.Model(num_classes=6, nb_lstm_units=4, channels=3, conv_kernel_size=(5, 5), lstm_layers=2, step=32,
                      image_size=(160, 120), conv_stride=2, effective_step=[7, 15, 23, 31], add_softmax=True)
    m.load_state_dict(R.to_torch(R.clstm_state_dict(channels=3, tag='clstm3')))
    m = m.cuda().eval()
    x = torch.from_numpy(np.stack([R.clip(3, 3, 32, 120, 160), R.clip(4, 3, 32, 120, 160)]) / 255.0).float().cuda()
    x.requires_grad_()
    y = m(x)
    assert rel_err(y.detach().cpu().numpy(), g['c3_probs']) < 1e-3
    (y[0, 2] + y[1, 4]).backward()
    assert rel_err(x.grad.cpu().numpy().ravel()[g['c3_dx_idx']], g['c3_dx_val']) < 2e-3


@pytest.mark.parametrize("C,T,H,W,hidden,kernel,stride", [(2, 6, 24, 32, 3, 3, 2), (1, 5, 16, 24, 4, 5, 1),
                                                          (3, 4, 18, 26, 2, 3, 1), (3, 5, 24, 32, 8, 5, 2),
                                                          (1, 4, 16, 16, 32, 3, 2), (2, 3, 12, 20, 16, 3, 1)])
def test_clstm_other_geometries_match_oracle(C, T, H, W, hidden, kernel, stride):
    """Geometries off the reference's (k 5, stride 2, hidden 4): the generic x-conv backward, the
    hidden < 4 forms of the wave-split cell steps and the blocked kernels for hidden 8 / 16 / 32
    (CLSTM_4.py:9 default nb_lstm_units=32), against the CPU oracle (forward and dL/dx)."""
    import ivf_engine
    import ivf_recipe as R
    from oracle import clstm_ref
    layers, B, K = 2, 2, 5
    sd_np = R.clstm_state_dict(num_classes=K, hidden=hidden, channels=C, kernel=kernel, layers=layers,
                               image_size=(W, H), conv_stride=stride, tag=f'clstm_g{C}{hidden}{kernel}{stride}')
    eng = ivf_engine.CLSTMEngine(K, (C, T, H, W), max_batch=B, hidden=hidden, layers=layers, kernel=kernel,
                                 stride=stride, softmax=True)
    eng.load_state_dict(sd_np)
    x = torch.from_numpy(np.stack([R.clip(7 + i, C, T, H, W) for i in range(B)]) / 255.0).float()
    sd = R.to_torch(sd_np)
    xr = x.clone().requires_grad_()
    y = clstm_ref.forward(xr, sd, layers=layers, hidden=hidden, kernel=kernel, stride=stride, steps=T,
                          effective_step=(T - 1,), add_softmax=True)
    y[0, 1].backward(retain_graph=True)
    g0 = xr.grad.clone()
    probs = eng.forward(x.cuda())
    assert rel_err(probs.cpu().numpy(), y.detach().numpy()) < 1e-3
    s, dx = eng.backward(B, target=[1, 2])
    assert abs(float(s[0]) - float(y[0, 1].detach())) < 1e-5
    assert rel_err(dx[0].cpu().numpy(), g0[0].numpy()) < 2e-3


def test_clstm_full_length_search(golden):
    """N=100, lam 0.02/0.04 (KTH:105-118): the whole trajectory, final mask, ranking, reverse score."""
    import ivf_recipe as R
    g = golden('search_long')
    eng = _engine(1, B=1)
    x = torch.from_numpy(R.clip(3, 1, 32, 120, 160) / 255.0).float()[None].cuda()
    target = int(g['c1_target'])
    raw = torch.from_numpy(g['c1_init'])[None].cuda().contiguous()
    traj, _ = eng.search(x, [target], raw, 0.02, 0.04, 100)
    traj = traj[:, 0].cpu().numpy()
    ref = g['c1_traj']
    rel = np.abs(traj - ref) / np.abs(ref)
    final = torch.sigmoid(raw)[0].cpu().numpy()
    dmask = float(np.max(np.abs(final - g['c1_mask'])))
    note(f"full search clstm N=100: trajectory rel err max {rel.max():.2e} last {rel[-1].max():.2e}; "
         f"final mask max|d| {dmask:.2e}")
    # the noise floor of this search, measured CPU-vs-CPU (tests/golden/search_spread.npz: the reference search in
    # fp64 and in fp32 with another thread count against the committed run), not taken from the GPU's own deviation
    sp = golden('search_spread')
    cpu_dmask = max(float(sp['c1_f64_dmask']), float(sp['c1_f32t4_dmask']))
    note(f"full search clstm: CPU-vs-CPU floor of the final mask {cpu_dmask:.2e}")
    assert rel.max() < 1e-2 and rel[-1].max() < 1e-2
    mask_tol = max(1e-5, 1.5 * cpu_dmask)
    assert dmask < mask_tol
    assert np.array_equal(final > 0.5, g['c1_mask'] > 0.5)
    rank = np.argsort(-final, kind='stable')
    tie_tol = 2 * mask_tol
    if np.min(np.diff(np.sort(g['c1_mask']))) > tie_tol:
        assert np.array_equal(rank, g['c1_ranking'])
    else:
        assert ranking_consistent(rank, g['c1_mask'], tie_tol)
    rev = eng.perturbed_forward(x, torch.sigmoid(raw), 'reverse')[0, target]
    assert abs(float(rev) - float(g['c1_reverse_score'])) < 1e-3


def test_clstm_reverse_mask_search(golden):
    """temporalMaskType='reverse': 30 iterations through the reverse operator vs the reference harness."""
    import ivf_recipe as R
    import ivf_search
    g = golden('search_reverse')
    eng = _engine(1, B=1)
    x = torch.from_numpy(R.clip(3, 1, 32, 120, 160) / 255.0).float()[None].cuda()
    target = int(g['c1_target'])
    probs = eng.forward(x)
    tgt = torch.tensor([target], dtype=torch.int32, device='cuda')
    raw, info = ivf_search.init_masks_central(eng, x, tgt, probs[0, target][None], 0.9, 'reverse')
    assert abs(float(info['full'][0]) - float(g['c1_full'])) < 1e-5     # the fully FROZEN clip (mask.py:123-128)
    assert np.array_equal(raw[0].cpu().numpy(), g['c1_init'])
    traj, _ = eng.search(x, [target], raw, 0.02, 0.04, 30, mode='reverse')
    got, ref = traj[:, 0].cpu().numpy(), g['c1_traj']
    assert np.max(np.abs(got - ref) / np.abs(ref)) < 1e-2
    final = torch.sigmoid(raw)[0].cpu().numpy()
    assert np.max(np.abs(final - g['c1_mask'])) < 2e-3
    assert ranking_consistent(np.argsort(-final, kind='stable'), g['c1_mask'], 5e-3)
    # whole MaskSearch with mask_type='reverse' reproduces the same init + loop
    res = ivf_search.MaskSearch(eng, 0.02, 0.04, 30, "reverse", do_gradcam=False).run(x, [target], want_traj=True)
    assert torch.equal(res["traj"][:, 0].cpu(), traj[:, 0].cpu())


def test_clstm_out_step_is_last_effective_step_reached():
    """output[-1] of the reference is the largest effective step BELOW `step`
    (convolution_lstm.py:129-130): step=8 with effective_step=[4,8,12,15] classifies step 4."""
    import ivf_recipe as R
    from models import CLSTM_4
    from oracle import clstm_ref
    sd_np = R.clstm_state_dict(num_classes=6, hidden=4, channels=1, kernel=5, layers=2, image_size=(64, 48),
                               conv_stride=2, tag='clstm_os')
    m = CLSTM_4.Model(num_classes=6, nb_lstm_units=4, channels=1, conv_kernel_size=(5, 5), lstm_layers=2, step=8,
                      image_size=(64, 48), conv_stride=2, effective_step=[4, 8, 12, 15], add_softmax=True)
    m.load_state_dict(R.to_torch(sd_np))
    m = m.cuda().eval()
    x = torch.from_numpy(R.clip(5, 1, 8, 48, 64) / 255.0).float()[None]
    y = m(x.cuda())
    want = clstm_ref.forward(x, R.to_torch(sd_np), steps=8, effective_step=(4, 8, 12, 15), add_softmax=True)
    assert rel_err(y.detach().cpu().numpy(), want.numpy()) < 1e-4
    m2 = CLSTM_4.Model(num_classes=6, nb_lstm_units=4, channels=1, conv_kernel_size=(5, 5), lstm_layers=2, step=3,
                       image_size=(64, 48), conv_stride=2, effective_step=[4, 8], add_softmax=True).cuda().eval()
    with pytest.raises(IndexError):
        m2(torch.zeros(1, 1, 3, 48, 64).cuda())          # the reference's output[-1] on an empty list


def test_clstm_use_entire_seq(golden):
    """use_entire_seq=True (CLSTM_4.py:73-76) through the drop-in Model: forward and input gradient vs the
    reference model's own output, two clips in ONE batch (each row must be that clip's own result)."""
    import ivf_recipe as R
    from models import CLSTM_4
    g = golden('clstm_seq')
    m = CLSTM_4.Model(num_classes=6, nb_lstm_units=4, channels=3, conv_kernel_size=(5, 5), lstm_layers=2, step=32,
                      image_size=(160, 120), conv_stride=2, effective_step=[7, 15, 23, 31], use_entire_seq=True,
                      add_softmax=True)
    m.load_state_dict(R.to_torch(R.clstm_state_dict(channels=3, tag='clstm_seq', fc_mult=4)))
    m = m.cuda().eval()
    x = torch.from_numpy(np.stack([R.clip(3, 3, 32, 120, 160), R.clip(4, 3, 32, 120, 160)]) / 255.0).float().cuda()
    x.requires_grad_()
    y = m(x)
    assert rel_err(y[0:1].detach().cpu().numpy(), g['clip3_probs']) < 1e-3
    assert rel_err(y[1:2].detach().cpu().numpy(), g['clip4_probs']) < 1e-3
    (y[0, 2] + y[1, 2]).backward()
    dx = x.grad.cpu().numpy()
    for i, cid in enumerate((3, 4)):
        assert rel_err(dx[i].ravel()[g[f'clip{cid}_dx_idx']], g[f'clip{cid}_dx_val']) < 2e-3
        assert rel_err(dx[i].astype(np.float64).sum(axis=(0, 2, 3)), g[f'clip{cid}_dx_sum_per_frame'][0]) < 2e-3
    m2 = CLSTM_4.Model(num_classes=6, nb_lstm_units=4, channels=3, conv_kernel_size=(5, 5), lstm_layers=2, step=16,
                       image_size=(160, 120), conv_stride=2, effective_step=[7, 15, 23, 31], use_entire_seq=True)
    with pytest.raises(RuntimeError):          # the reference's .view fails when effective steps lie past `step`
        m2.cuda().eval()(torch.zeros(1, 3, 16, 120, 160).cuda())


@pytest.mark.parametrize("C", [1, 3])
def test_clstm_persistent_recurrence_matches_stepwise_and_reference(C, golden):
    """Batches of 64+ clips run each layer's recurrence as ONE persistent launch per direction (a workgroup per clip,
    hidden state / dG gate groups in LDS: clstm_seq_fwd/bwd_kernel) instead of one launch per time step.  Same clips
    through both forms (rows are independent): logits, probabilities and the input gradient agree to fp32 rounding,
    and the persistent form meets the reference fixture on its own."""
    import ivf_recipe as R
    g = golden('clstm')
    base = np.stack([R.clip(3, C, 32, 120, 160), R.clip(4, C, 32, 120, 160)]) / 255.0
    small = _engine(C, B=2)                       # stepwise kernels (batch below the persistent threshold)
    xs = torch.from_numpy(base).float().cuda()
    ps, ls = small.forward(xs, want_logits=True)
    dout = torch.zeros(2, 6, device='cuda')
    dout[0, 2] = 1
    dout[1, 4] = 1
    _, dxs = small.backward(2, dout=dout)
    B = 64
    big = _engine(C, B=B)                         # persistent kernels
    xb = torch.from_numpy(np.concatenate([base] * (B // 2))).float().cuda()
    pb, lb = big.forward(xb, want_logits=True)
    _, dxb = big.backward(B, dout=dout.repeat(B // 2, 1))
    for r in (0, 1, B - 2, B - 1):
        assert rel_err(lb[r].cpu().numpy(), ls[r % 2].cpu().numpy()) < 1e-4      # (v_exp_f32 gate functions in the persistent form)
        assert rel_err(pb[r].cpu().numpy(), ps[r % 2].cpu().numpy()) < 1e-4
        assert rel_err(dxb[r].cpu().numpy(), dxs[r % 2].cpu().numpy()) < 5e-4
    assert rel_err(lb[:2].cpu().numpy(), g[f'c{C}_logits']) < 1e-3
    assert rel_err(pb[:2].cpu().numpy(), g[f'c{C}_probs']) < 1e-3
    dxn = dxb[:2].cpu().numpy()
    assert rel_err(dxn.ravel()[g[f'c{C}_dx_idx']], g[f'c{C}_dx_val']) < 2e-3
    note(f"clstm persistent vs stepwise (C={C}): logits {rel_err(lb[:2].cpu().numpy(), ls.cpu().numpy()):.2e}, "
         f"dx {rel_err(dxb[:2].cpu().numpy(), dxs.cpu().numpy()):.2e}")
