"""The reference's Python call surface on top of the HIP library: the literal
loop of FindMasksComparison_I3D_smth.py:188-214 written against `mask`,
`models.I3D_doubled.Model` and torch.optim.Adam must reproduce the reference's
own trajectory; find_masks must write the reference's result records."""
import os
import pickle

import numpy as np
import pytest
import torch

from conftest import rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def model():
    import ivf_recipe as R
    from models import I3D_doubled
    m = I3D_doubled.Model(174, last_stride=1, stride_mod_layers="", softMax=1)
    m.load_state_dict({"module." + k: v for k, v in R.to_torch(R.i3d_state_dict(num_classes=174)).items()})
    return m.cuda().eval()


def test_reference_literal_loop(model, golden):
    import ivf_recipe as R
    import mask
    g = golden('search')
    x = torch.from_numpy(R.clip(21))[None].cuda()
    output = model(x)
    target = torch.zeros((1, 1)).long()
    target[0] = torch.argmax(output[0])
    assert int(target[0]) == int(g['s16_target'])
    time_mask = mask.init_mask(x, model, 0, target, threshold=0.9, mode="central", mask_type="freeze")
    assert np.array_equal(time_mask.detach().cpu().numpy(), g['s16_init'])       # +-5 pattern: exact
    optimizer = torch.optim.Adam([time_mask], lr=0.2)
    traj = []
    for nidx in range(4):
        mask_clip = torch.sigmoid(time_mask)
        l1loss = 0.01 * torch.sum(torch.abs(mask_clip))
        tvnorm_loss = 0.02 * mask.calc_tv_norm(mask_clip, p=3, q=3)
        class_loss = model(mask.perturb_sequence(x, mask_clip, perturbation_type="freeze"))
        class_loss = class_loss[0, target[0]]
        loss = l1loss + tvnorm_loss + class_loss
        optimizer.zero_grad()
        loss.backward()
        if nidx == 0:
            assert rel_err(time_mask.grad.cpu().numpy(), g['s16_grad0']) < 2e-2
        optimizer.step()
        traj.append([loss.item(), l1loss.item(), tvnorm_loss.item(), class_loss.item()])
    ref = g['s16_traj'][:4]
    assert np.max(np.abs(np.array(traj) - ref) / np.abs(ref)) < 1e-2
    # legacy names of the KTH driver resolve to the same functions
    assert mask.calc_TVNorm is mask.calc_tv_norm and mask.perturbSequence is mask.perturb_sequence
    p = mask.perturbSequence(x, torch.sigmoid(time_mask.detach()), perbType="reverse")
    assert p.shape == x.shape


def test_model_guards(model):
    import ivf_lib as L
    with pytest.raises(L.IvfError):
        model(torch.zeros(1, 3, 16, 224, 224))          # CPU tensor: no fallback path
    model.train()
    with pytest.raises(L.IvfError):
        model(torch.zeros(1, 3, 16, 224, 224).cuda())   # train-mode BN/dropout are not on the path
    model.eval()
    with pytest.raises(TypeError):
        from models import I3D_doubled
        I3D_doubled.Model(174, stride_mod_layers=None)   # same failure as the reference


def test_gradcam_video_class(model, golden):
    import ivf_recipe as R
    from grad_cam_videos import GradCamVideo
    g = golden('gradcam')
    x = torch.from_numpy(R.clip(11))[None].cuda()
    gc = GradCamVideo(model=model, target_layer_names=['Mixed_5c'], class_dict=None, use_cuda=True,
                      input_spatial_size=(224, 224), normalizePerFrame=True, archType="I3D")
    cam, output = gc(x, None)
    assert cam.shape == (16, 224, 224) and cam.dtype == np.float32 and tuple(output.shape) == (1, 174)
    assert np.max(np.abs(cam[:, ::8, ::8] - g['pf_cam_small'])) < 2e-3
    assert rel_err(output.cpu().numpy(), g['pf_output']) < 1e-3


def test_find_masks_records(model, tmp_path, monkeypatch):
    import FindMasksComparison_I3D_smth as drv
    import ivf_find_masks
    monkeypatch.chdir(tmp_path)
    loader = ivf_find_masks.SyntheticLoader(2, 2, (3, 16, 224, 224), 174, first_id=40)
    hp = {"batch_size": 2, "gradCamType": "guessed"}
    masks = drv.find_masks(loader, torch.nn.Sequential() if False else model, hp, 0.01, 0.02, 5, "central", "freeze",
                           classOI=None, doGradCam=True, runTempMask=True, verbose=False)
    assert len(masks) == 2 and masks[0].shape == (16,)
    tm = pickle.load(open(tmp_path / "results" / "allTimeMaskResults_run0_None_.p", "rb"))
    gc = pickle.load(open(tmp_path / "results" / "allGradCamResults_run0_None_.p", "rb"))
    assert set(tm[0]) == {'true_class', 'pred_class', 'video_id', 'time_mask', 'original_score_guess',
                          'original_score_true', 'freeze_score', 'reverse_score'}
    assert set(gc[0]) == {'true_class', 'pred_class', 'video_id', 'GCHeatMap'}
    assert tm[0]['time_mask'].shape == (16,) and gc[0]['GCHeatMap'].shape == (16, 224, 224)
    assert tm[0]['original_score_guess'] == 0          # smth:218 casts the probability with int()
    files = [str(p) for p in (tmp_path / "cam_saved_images").rglob("*.txt")]
    assert any("ClassScoreFreezecase40" in f for f in files) and any("ClassScoreReversecase41" in f for f in files)
    # batched per-clip search == one-clip-at-a-time search, bit for bit (rows independent)
    loader1 = ivf_find_masks.SyntheticLoader(1, 1, (3, 16, 224, 224), 174, first_id=41)
    m1 = drv.find_masks(loader1, model, {"batch_size": 1, "gradCamType": "guessed"}, 0.01, 0.02, 5, "central",
                        "freeze", classOI=None, doGradCam=False, runTempMask=True, verbose=False)
    assert torch.equal(m1[0], masks[1])


def test_mask_module_edges(model):
    """argument handling the reference has: unknown perturbation type, random init,
    snap_values mutating the caller's mask, sub-mask lists on the device."""
    import ivf_recipe as R
    import mask
    x = torch.from_numpy(R.uniform('t/edge/x', (1, 3, 16, 8, 8), 0, 255)).cuda()
    m = torch.from_numpy(R.uniform('t/edge/m', (16,), 0, 1)).cuda()
    with pytest.raises(UnboundLocalError):
        mask.perturb_sequence(x, m, perturbation_type='blur')        # mask.py:57 returns an unset local
    m2 = m.clone()
    p = mask.perturb_sequence(x, m2, 'freeze', snap_values=True)
    assert set(m2.unique().tolist()) <= {0.0, 1.0} and torch.equal(m2, (m > 0.5).float())   # mask.py:5-10
    assert p.shape == x.shape
    runs = mask.find_submasks_from_mask(torch.tensor([0, .2, .3, 0, 0, .5, 0, .1, .11, 0, 0, 0, 0, 0, .9, .9]).cuda())
    assert runs == [[1, 2], [5], [8], [14, 15]]
    torch.manual_seed(0)
    r = mask.init_mask(x, None, 0, None if False else torch.zeros(1, 1).long(), mode="random")
    assert r.requires_grad and set(r.detach().abs().round(decimals=1).unique().tolist()) <= {2.5, 2.6}
    tv = mask.calc_tv_norm(torch.full((16,), 0.3).cuda().requires_grad_())
    assert float(tv) == 0.0


def test_find_masks_class_filter_empty(model, tmp_path, monkeypatch):
    """classOI csv that selects no clip: no search runs, empty pickles are still written."""
    import pickle as pk
    import FindMasksComparison_I3D_smth as drv
    import ivf_find_masks
    monkeypatch.chdir(tmp_path)
    (tmp_path / "sel.csv").write_text("5,7\n123,456\n")
    loader = ivf_find_masks.SyntheticLoader(2, 2, (3, 16, 224, 224), 174, first_id=40)
    masks = drv.find_masks(loader, model, {"batch_size": 2, "gradCamType": "guessed"}, 0.01, 0.02, 2, "central",
                           "freeze", classOI=str(tmp_path / "sel.csv"), doGradCam=True, runTempMask=True, verbose=False)
    assert masks == []
    files = list((tmp_path / "results").glob("allTimeMaskResults_*"))
    assert len(files) == 1 and pk.load(open(files[0], "rb")) == []
