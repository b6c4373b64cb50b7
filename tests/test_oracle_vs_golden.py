"""CPU: the oracle (torch-CPU restatement under oracle/) against outputs of the
REFERENCE modules themselves (tests/golden/*.npz, made by make_golden.py which
imports /root/reference).  This is what pins the oracle."""
import numpy as np
import pytest
import torch

import ivf_recipe as R
from conftest import ranking_consistent, rel_err
from oracle import clstm_ref, gradcam_ref, i3d_ref, ingest_ref, mask_ref

torch.set_num_threads(8)


@pytest.mark.parametrize("T", [16, 32])
def test_freeze_matches_reference(T, golden):
    g = golden('mask_ops')
    x = torch.from_numpy(R.uniform(f'g/freeze/x{T}', (2, 3, T, 12, 20), 0, 255)).requires_grad_()
    m = torch.from_numpy(R.uniform(f'g/freeze/m{T}', (T,), 0, 1)).requires_grad_()
    gy = torch.from_numpy(R.uniform(f'g/freeze/g{T}', (2, 3, T, 12, 20), -1, 1))
    p = mask_ref.perturb_sequence(x, m, 'freeze')
    (p * gy).sum().backward()
    assert np.array_equal(p.detach().numpy(), g[f'freeze{T}_p'])
    assert rel_err(m.grad.numpy(), g[f'freeze{T}_dm']) < 1e-5
    assert rel_err(x.grad.numpy(), g[f'freeze{T}_dx']) < 1e-6


@pytest.mark.parametrize("case", ['even_mid', 'odd_mid', 'ends', 'thresh', 'all_on', 'all_off', 'single'])
def test_reverse_and_submasks_match_reference(case, golden):
    g = golden('mask_ops')
    x = torch.from_numpy(R.uniform('g/reverse/x', (2, 3, 16, 6, 10), 0, 255))
    m = torch.from_numpy(g[f'rev_{case}_mask'])
    assert np.array_equal(mask_ref.perturb_sequence(x, m, 'reverse').numpy(), g[f'rev_{case}_p'])
    flat = [-1]
    for s in mask_ref.find_submasks_from_mask(m, 0.1):
        flat += s + [-1]
    assert flat == g[f'rev_{case}_subs'].tolist()


def test_snap_matches_reference(golden):
    g = golden('mask_ops')
    m = torch.from_numpy(g['snap_in'].copy())
    xs = torch.from_numpy(R.uniform('g/snap/x', (1, 3, 16, 4, 4), 0, 255))
    p = mask_ref.perturb_sequence(xs, m, 'freeze', snap_values=True)
    assert np.array_equal(m.numpy(), g['snap_out']) and np.array_equal(p.numpy(), g['snap_p'])


@pytest.mark.parametrize("case", ['rand16', 'rand32', 'mono', 'near_const', 'sig_pm5', 'const'])
def test_tv_norm_matches_reference(case, golden):
    g = golden('mask_ops')
    m = torch.from_numpy(g[f'tv_{case}_in']).requires_grad_()
    tv = mask_ref.calc_tv_norm(m, 3, 3)
    tv.backward()
    assert np.array_equal(tv.detach().numpy(), g[f'tv_{case}_val'])
    assert np.array_equal(m.grad.numpy(), g[f'tv_{case}_grad'], equal_nan=True)


def test_adam_matches_torch_optim(golden):
    g = golden('mask_ops')
    p = torch.from_numpy(R.uniform('g/adam/p', (16,), -5, 5))
    grads = R.uniform('g/adam/g', (12, 16), -1e-2, 1e-2)
    opt = mask_ref.Adam(p, lr=0.2)
    traj = [p.numpy().copy()]
    for gr in grads:
        opt.step(torch.from_numpy(gr.copy()))
        traj.append(p.numpy().copy())
    assert np.allclose(np.array(traj), g['adam_traj'], rtol=1e-6, atol=1e-7)


def test_i3d_s16_matches_reference_model(golden):
    g = golden('i3d')
    sd = R.to_torch(R.i3d_state_dict(num_classes=174))
    x = torch.from_numpy(R.clip(7))[None].requires_grad_()
    eps = {}
    feat = i3d_ref.features(x, sd, endpoints=eps)
    feat.retain_grad()
    logits, probs = i3d_ref.head(feat, sd, (2, 7, 7), True)
    assert rel_err(logits.detach().numpy(), g['s16_logits']) < 1e-5
    assert rel_err(probs.detach().numpy(), g['s16_probs']) < 1e-5
    t = int(probs[0].argmax())
    assert t == int(g['s16_target'])
    for n, a in eps.items():
        assert abs(float(a.detach().double().norm()) - float(g[f's16_norm_{n}'])) < 1e-5 * float(g[f's16_norm_{n}'])
    probs[0, t].backward()
    assert rel_err(feat.detach().numpy().ravel()[g['s16_feat_idx']], g['s16_feat_val']) < 1e-5
    assert rel_err(feat.grad.numpy().ravel()[g['s16_feat_idx']], g['s16_dfeat_val']) < 1e-4
    assert rel_err(x.grad.numpy().ravel()[g['s16_dx_idx']], g['s16_dx_val']) < 1e-3


@pytest.mark.parametrize("C", [1, 3])
def test_clstm_matches_reference_model(C, golden):
    g = golden('clstm')
    sd = R.to_torch(R.clstm_state_dict(channels=C, tag=f'clstm{C}'))
    x = torch.from_numpy(np.stack([R.clip(3, C, 32, 120, 160), R.clip(4, C, 32, 120, 160)]) / 255.0).float()
    x.requires_grad_()
    y = clstm_ref.forward(x, sd, add_softmax=True)
    assert rel_err(y.detach().numpy(), g[f'c{C}_probs']) < 1e-5
    with torch.no_grad():
        assert rel_err(clstm_ref.forward(x, sd, add_softmax=False).numpy(), g[f'c{C}_logits']) < 1e-5
    (y[0, 2] + y[1, 4]).backward()
    assert rel_err(x.grad.numpy().ravel()[g[f'c{C}_dx_idx']], g[f'c{C}_dx_val']) < 1e-4
    assert rel_err(x.grad.numpy().astype(np.float64).sum(axis=(1, 3, 4)), g[f'c{C}_dx_sum_per_frame']) < 1e-4


def test_gradcam_matches_reference(golden):
    g = golden('gradcam')
    sd = R.to_torch(R.i3d_state_dict(num_classes=174))
    x = torch.from_numpy(R.clip(11))[None]
    for tag, pf in (('pf', True), ('glob', False)):
        cam, out, _ = gradcam_ref.gradcam_i3d(x, sd, None, normalize_per_frame=pf)
        assert cam.shape == (16, 224, 224)
        assert rel_err(out.numpy(), g[f'{tag}_output']) < 1e-5
        assert np.allclose(cam[:, ::8, ::8], g[f'{tag}_cam_small'], atol=2e-5, equal_nan=True)
        assert np.allclose(cam[[0, 7, 8, 15]][:, [0, 100, 223]], g[f'{tag}_cam_rows'], atol=2e-5, equal_nan=True)
    cam, out, ex = gradcam_ref.gradcam_i3d(x, sd, 5, normalize_per_frame=True)
    assert rel_err(ex['weights'], g['idx5_weights']) < 1e-4
    assert np.array_equal(np.isnan(cam[:, ::8, ::8]), np.isnan(g['idx5_cam_small']))


def test_search_trajectory_matches_reference(golden):
    """The oracle's search loop (init_mask + Adam loop + reverse score) vs the harness
    around the reference's mask.py + model + torch.optim.Adam.  CLSTM variant (cheap)."""
    g = golden('search')
    sd = R.to_torch(R.clstm_state_dict(channels=1, tag='clstm1'))
    x = torch.from_numpy(R.clip(3, 1, 32, 120, 160) / 255.0).float()[None]
    target = int(g['c1_target'])

    def score_fn(v):
        return clstm_ref.forward(v, sd, add_softmax=True)[0, target]
    res = mask_ref.search_clip(x, score_fn, 0.02, 0.04, 30, 'freeze')
    assert abs(res['init']['full'] - float(g['c1_full'])) < 1e-6 and abs(res['init']['orig'] - float(g['c1_orig'])) < 1e-6
    assert np.allclose(res['init']['central'], g['c1_central'], atol=1e-6)
    assert np.array_equal(torch.where(res['raw_mask'] > 0, 1, 0).numpy() * 0 + (g['c1_init'] > 0), g['c1_init'] > 0)
    assert np.max(np.abs(res['traj'].numpy() - g['c1_traj']) / np.abs(g['c1_traj'])) < 1e-4
    assert np.max(np.abs(res['mask'].numpy() - g['c1_mask'])) < 1e-4
    assert abs(res['reverse_score'] - float(g['c1_reverse_score'])) < 1e-5
    assert np.array_equal(mask_ref.frame_ranking(res['mask']).numpy(),
                          np.argsort(-g['c1_mask'], kind='stable'))


def test_i3d_search_first_iterations_match_reference(golden):
    """Two I3D iterations of the oracle loop vs the reference harness (full size)."""
    g = golden('search')
    sd = R.to_torch(R.i3d_state_dict(num_classes=174))
    x = torch.from_numpy(R.clip(21))[None]
    target = int(g['s16_target'])

    def score_fn(v):
        return i3d_ref.forward(v, sd)[0, target]
    res = mask_ref.search_clip(x, score_fn, 0.01, 0.02, 2, 'freeze', init=torch.from_numpy(g['s16_init']))
    assert np.max(np.abs(res['traj'].numpy() - g['s16_traj'][:2]) / np.abs(g['s16_traj'][:2])) < 1e-4


def test_ingest_matches_reference_loaders(golden):
    """the oracle's decode + cast + permute equals what the reference's
    ImLoader / KTHImLoader returned for the same JPEG bytes."""
    g = golden('ingest')
    T = int(g['shape'][0])
    frames = ingest_ref.decode_frames([g[f'jpeg{i}'] for i in range(T)])
    assert frames.dtype == np.uint8 and frames.shape == (T, int(g['shape'][1]), int(g['shape'][2]), 3)
    x = ingest_ref.to_model_input(frames)
    assert x.dtype == np.float32
    assert np.array_equal(x, g['kth_data']) and np.array_equal(x, g['smth_data'])
    cl = ingest_ref.to_channels_last(frames, 4)
    assert np.array_equal(cl[..., :3].transpose(3, 0, 1, 2), x) and not cl[..., 3].any()


# ------------------------------------------------------------------ round-2 fixtures
def test_i3d_s32_branch_matches_reference_model(golden):
    """stride_mod_layers="none", head window [4,7,7] (SURVEY F13 / BASELINE configs[4]): the
    oracle branch the GPU tests lean on, against the reference model's own output."""
    g = golden('i3d_s32')
    sd = R.to_torch(R.i3d_state_dict(num_classes=174))
    x = torch.from_numpy(R.clip(13, 3, 32, 224, 224))[None].requires_grad_()
    eps = {}
    feat = i3d_ref.features(x, sd, "none", 1, endpoints=eps)
    feat.retain_grad()
    logits, probs = i3d_ref.head(feat, sd, (4, 7, 7), True)
    assert list(feat.shape) == g['s32_feat_shape'].tolist()
    assert rel_err(logits.detach().numpy(), g['s32_logits']) < 1e-5
    assert rel_err(probs.detach().numpy(), g['s32_probs']) < 1e-5
    t = int(probs[0].argmax())
    assert t == int(g['s32_target'])
    for n, a in eps.items():
        assert abs(float(a.detach().double().norm()) - float(g[f's32_norm_{n}'])) < 1e-5 * float(g[f's32_norm_{n}'])
    probs[0, t].backward()
    assert rel_err(feat.grad.numpy().ravel()[g['s32_feat_idx']], g['s32_dfeat_val']) < 1e-4
    assert rel_err(x.grad.numpy().ravel()[g['s32_dx_idx']], g['s32_dx_val']) < 1e-3
    xg = x.detach()
    for tag, pf in (('pf', True), ('glob', False)):
        cam, out, _ = gradcam_ref.gradcam_i3d(xg, sd, None, pool_kernel=(4, 7, 7), normalize_per_frame=pf,
                                              stride_mod_layers="none", last_stride=1)
        assert list(cam.shape) == g[f'gc_{tag}_cam_shape'].tolist() == [32, 224, 224]
        assert rel_err(out.numpy(), g[f'gc_{tag}_output']) < 1e-5
        assert np.allclose(cam[:, ::8, ::8], g[f'gc_{tag}_cam_small'], atol=2e-5, equal_nan=True)


def test_gradcam_k32_matches_reference(golden):
    """GradCamVideo as the KTH driver calls it (KTH:315-327): input_spatial_size=(160,120),
    Mixed_5c [1,1024,4,4,5] -> [32,120,160]."""
    g = golden('gradcam_k32')
    sd = R.to_torch(R.i3d_state_dict(num_classes=6, tag='i3d_kth'))
    x = torch.from_numpy(R.clip(11, 3, 32, 120, 160))[None]
    for tag, pf in (('pf', True), ('glob', False)):
        cam, out, _ = gradcam_ref.gradcam_i3d(x, sd, None, pool_kernel=(4, 4, 5), width=160, height=120,
                                              normalize_per_frame=pf)
        assert list(cam.shape) == g[f'{tag}_cam_shape'].tolist() == [32, 120, 160]
        assert rel_err(out.numpy(), g[f'{tag}_output']) < 1e-5
        assert np.allclose(cam[:, ::4, ::4], g[f'{tag}_cam_small'], atol=2e-5, equal_nan=True)
    # (the fixture's explicit-index call used the globally normalised instance)
    cam, out, ex = gradcam_ref.gradcam_i3d(x, sd, 3, pool_kernel=(4, 4, 5), width=160, height=120,
                                           normalize_per_frame=False)
    assert rel_err(ex['weights'], g['idx3_weights']) < 1e-4
    assert np.allclose(cam[:, ::4, ::4], g['idx3_cam_small'], atol=2e-5, equal_nan=True)


def test_resize_bilinear_against_two_independent_implementations():
    """cv2.resize (INTER_LINEAR) is absent here, so the restatement is cross-checked against two
    independent bilinear up-scalers with OpenCV's convention (half-pixel centres, edge clamp):
    torch F.interpolate(align_corners=False) and PIL mode-'F' BILINEAR, at the three geometries
    Grad-CAM uses (7x7 -> 224x224, 4x5 -> 120x160, non-square)."""
    import torch.nn.functional as F
    from PIL import Image
    for (sh, sw, dh, dw) in ((7, 7, 224, 224), (4, 5, 120, 160), (7, 7, 120, 160)):
        src = R.uniform(f't/resize/{sh}x{sw}', (sh, sw), 0, 3)
        ours = gradcam_ref.resize_bilinear(src, dw, dh)
        ti = F.interpolate(torch.from_numpy(src)[None, None], size=(dh, dw), mode='bilinear',
                           align_corners=False)[0, 0].numpy()
        pil = np.asarray(Image.fromarray(src, mode='F').resize((dw, dh), Image.BILINEAR))
        assert ours.shape == (dh, dw)
        assert np.max(np.abs(ours - ti)) < 2e-6 * 3
        assert np.max(np.abs(ours - pil)) < 2e-6 * 3


def test_reverse_mask_search_matches_reference(golden):
    """temporalMaskType='reverse': init_mask scores the fully FROZEN clip whatever the mask type
    (mask.py:123-128), the loop optimises through the reverse operator (mask.py:49-56)."""
    g = golden('search_reverse')
    sd = R.to_torch(R.clstm_state_dict(channels=1, tag='clstm1'))
    x = torch.from_numpy(R.clip(3, 1, 32, 120, 160) / 255.0).float()[None]
    target = int(g['c1_target'])

    def score_fn(v):
        return clstm_ref.forward(v, sd, add_softmax=True)[0, target]
    res = mask_ref.search_clip(x, score_fn, 0.02, 0.04, 30, 'reverse')
    assert abs(res['init']['full'] - float(g['c1_full'])) < 1e-6
    assert np.allclose(res['init']['central'], g['c1_central'], atol=1e-6)
    assert np.max(np.abs(res['traj'].numpy() - g['c1_traj']) / np.abs(g['c1_traj'])) < 1e-4
    assert np.max(np.abs(res['mask'].numpy() - g['c1_mask'])) < 1e-4
    # the interior frames of the run move under the regulariser only and stay tied to rounding
    assert ranking_consistent(mask_ref.frame_ranking(res['mask']).numpy(), g['c1_mask'], 1e-4)
    assert np.array_equal(mask_ref.frame_ranking(res['mask']).numpy()[:4], g['c1_ranking'][:4])


def test_clstm_full_length_search_matches_reference(golden):
    """N=100 (KTH:118): the whole trajectory, final mask, ranking, scores."""
    g = golden('search_long')
    sd = R.to_torch(R.clstm_state_dict(channels=1, tag='clstm1'))
    x = torch.from_numpy(R.clip(3, 1, 32, 120, 160) / 255.0).float()[None]
    target = int(g['c1_target'])

    def score_fn(v):
        return clstm_ref.forward(v, sd, add_softmax=True)[0, target]
    res = mask_ref.search_clip(x, score_fn, 0.02, 0.04, 100, 'freeze')
    assert g['c1_traj'].shape == (100, 4)
    assert np.max(np.abs(res['traj'].numpy() - g['c1_traj']) / np.abs(g['c1_traj'])) < 5e-4
    assert np.max(np.abs(res['mask'].numpy() - g['c1_mask'])) < 5e-4
    assert np.array_equal(mask_ref.frame_ranking(res['mask']).numpy(), g['c1_ranking'])
    assert np.array_equal(res['mask'].numpy() > 0.5, g['c1_mask'] > 0.5)
    assert abs(res['reverse_score'] - float(g['c1_reverse_score'])) < 1e-4


def _viz_inputs(tag, T, H, W):
    x = torch.from_numpy(R.uniform(f'g/viz/{tag}/x', (2, 3, T, H, W), 0, 255)).round()
    cam = R.uniform(f'g/viz/{tag}/cam', (T, H, W), 0, 1).astype(np.float32)
    cam[3] = 0.0
    cam[5, :4] = 1.0
    tm = torch.from_numpy(R.uniform(f'g/viz/{tag}/tm', (T,), 0, 1))
    tm[2] = 0.5
    return x, cam, tm


def test_viz_blend_matches_reference(golden):
    """oracle/viz_ref.py against what the reference's create_image_arrays +
    vizualize_results_on_gradcam returned (JET table shared: parity unpinned for that table only)."""
    from oracle import viz_ref
    g = golden('viz')
    for tag, (T, H, W), kinds in (('a', (8, 14, 224), ('freeze', 'reverse')), ('b', (32, 12, 160), ('freeze',))):
        x, cam, tm = _viz_inputs(tag, T, H, W)
        for kind in kinds:
            m = tm.clone()
            snapped = mask_ref.snap(m.clone())
            pert = mask_ref.perturb_sequence(x, snapped, kind)[1].numpy()
            strip = viz_ref.combine_frames(x[1].numpy(), cam, pert)                  # [T,H,3W,3]
            img = np.ascontiguousarray(strip.transpose(3, 0, 1, 2))
            mk = m.numpy()
            viz_ref.draw_dots(img, mk)                                               # snaps mk in place
            assert np.array_equal(mk, g[f'{tag}_{kind}_mask_after'])
            if tag == 'a':
                assert np.array_equal(img, g[f'a_{kind}_img'])
            else:
                assert np.array_equal(img[..., 2 * W:], g[f'b_{kind}_panel3'])
                assert int(img[..., :2 * W].astype(np.int64).sum()) == int(g[f'b_{kind}_sum12'])
    lut = viz_ref.jet_lut_bgr()
    assert lut.shape == (256, 3) and lut.dtype == np.uint8
    assert lut[0].tolist() == [143, 0, 0] and lut[255].tolist() == [0, 0, 128] and lut[128].tolist()[1] == 255


def test_clstm_use_entire_seq_matches_reference(golden):
    """CLSTM_4.py:73-76: endFC over the concatenated outputs of all effective steps."""
    g = golden('clstm_seq')
    sd = R.to_torch(R.clstm_state_dict(channels=3, tag='clstm_seq', fc_mult=4))
    for cid in (3, 4):
        x = torch.from_numpy(R.clip(cid, 3, 32, 120, 160) / 255.0)[None].float().requires_grad_()
        y = clstm_ref.forward(x, sd, add_softmax=True, use_entire_seq=True)
        assert rel_err(y.detach().numpy(), g[f'clip{cid}_probs']) < 1e-5
        y[0, 2].backward()
        assert rel_err(x.grad.numpy().ravel()[g[f'clip{cid}_dx_idx']], g[f'clip{cid}_dx_val']) < 1e-4


@pytest.mark.parametrize("layer", ['Conv3d_2c_3x3', 'Mixed_3c', 'Mixed_4f'])
def test_gradcam_other_target_layers_match_reference(layer, golden):
    """Target layers other than Mixed_5c (grad-cam.py:23-54): the hook sees the gradient w.r.t. the named
    module's OUTPUT (ungated by its own ReLU), also below a max-pool (dead windows route to the first cell)."""
    g = golden('gradcam_layers')
    sd = R.to_torch(R.i3d_state_dict(num_classes=174))
    x = torch.from_numpy(R.clip(11))[None]
    cam, out, ex = gradcam_ref.gradcam_i3d(x, sd, None, layer=layer)
    assert list(cam.shape) == g[f'{layer}_cam_shape'].tolist()
    assert rel_err(ex['weights'], g[f'{layer}_weights']) < 1e-4
    assert np.allclose(cam[:, ::8, ::8], g[f'{layer}_cam_small'], atol=5e-5, equal_nan=True)
