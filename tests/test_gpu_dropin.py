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
    # batched per-clip search == one-clip-at-a-time search, bit for bit (rows independent); without Grad-CAM the
    # smth driver returns no masks (smth:296-303 appends inside `if doGradCam and runTempMask`)
    loader1 = ivf_find_masks.SyntheticLoader(1, 1, (3, 16, 224, 224), 174, first_id=41)
    m1 = drv.find_masks(loader1, model, {"batch_size": 1, "gradCamType": "guessed"}, 0.01, 0.02, 5, "central",
                        "freeze", classOI=None, doGradCam=False, runTempMask=True, verbose=False)
    assert m1 == []
    assert np.array_equal(ivf_find_masks.find_masks_impl.last_results[0][0]['time_mask'], tm[1]['time_mask'])
    # the returned masks are the sigmoid values: the visualisation's dot row snaps a host copy of the CUDA
    # time_mask (visualisation.py:39,77-81), never the caller's tensor
    assert np.array_equal(masks[1].cpu().numpy(), tm[1]['time_mask'])
    assert not set(masks[1].unique().tolist()) <= {0.0, 1.0}
    pngs = [str(p) for p in (tmp_path / "cam_saved_images").rglob("*.png")]
    assert any("casefreeze40_15.png" in f for f in pngs) and any("casereverse41_0.png" in f for f in pngs)
    assert len(list((tmp_path / "cam_saved_images").rglob("mygif.gif"))) == 2


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


# ------------------------------------------------------------------ round 2
def test_snap_values_vs_reference(golden):
    """mask.py:5-10 against the reference's own output (mask_ops.npz snap_in/snap_out/snap_p):
    the caller's mask is snapped IN PLACE and the frozen clip uses the snapped values."""
    import ivf_recipe as R
    import mask
    g = golden('mask_ops')
    m = torch.from_numpy(g['snap_in'].copy()).cuda()
    xs = torch.from_numpy(R.uniform('g/snap/x', (1, 3, 16, 4, 4), 0, 255)).cuda()
    p = mask.perturb_sequence(xs, m, 'freeze', snap_values=True)
    assert np.array_equal(m.cpu().numpy(), g['snap_out'])            # integer-valued output: bit-exact
    assert np.array_equal(p.cpu().numpy(), g['snap_p'])


@pytest.mark.parametrize("backbone", ["i3d", "clstm"])
def test_kth_driver_find_masks(backbone, tmp_path, monkeypatch):
    """FindMasksComparison_I3D_KTH.find_masks (KTH arity with `ita`, KTH:126-127) on both backbones
    (KTH:50-58): KTH result-file names (KTH:372-378), un-cast original_score_guess, [32,120,160] maps."""
    import FindMasksComparison_I3D_KTH as drv
    import ivf_find_masks
    import ivf_recipe as R
    monkeypatch.chdir(tmp_path)
    if backbone == "i3d":
        from models import I3D_doubled_kth
        m = I3D_doubled_kth.Model(6, last_stride=1, stride_mod_layers="", finalTimeLength=4, softMax=1)
        m.load_state_dict(R.to_torch(R.i3d_state_dict(num_classes=6, tag='i3d_kth')))
    else:
        from models import CLSTM_4
        m = CLSTM_4.Model(num_classes=6, nb_lstm_units=4, channels=3, conv_kernel_size=(5, 5), lstm_layers=2,
                          step=32, image_size=(160, 120), conv_stride=2, effective_step=[7, 15, 23, 31])
        m.load_state_dict(R.to_torch(R.clstm_state_dict(channels=3, tag='clstm3')))
    m = m.cuda().eval()
    loader = ivf_find_masks.SyntheticLoader(2, 2, (3, 32, 120, 160), 6, first_id=7)
    cfg = {"batch_size": 2, "gradCamType": "guessed"}
    masks = drv.find_masks(loader, m, cfg, 0.02, 0.04, 4, 1, "central", "freeze", classOI=None,
                           doGradCam=(backbone == "i3d"), runTempMask=True, verbose=False)
    assert len(masks) == 2 and masks[0].shape == (32,)
    tm = pickle.load(open(tmp_path / "results" / "I3d_KTH_allTimeMaskResults_original_run0.p", "rb"))
    gc = pickle.load(open(tmp_path / "results" / "I3d_KTH_allGradCamResults_original_run0.p", "rb"))
    assert len(tm) == 2 and tm[0]['time_mask'].shape == (32,) and tm[0]['video_id'] == "7"
    assert isinstance(tm[0]['original_score_guess'], float) and tm[0]['original_score_guess'] > 0   # no int() cast (KTH:274)
    # KTH:360-367: the masks come back unsnapped, and the PerturbImgs PNGs carry the SOFT mask value in the
    # top-left 10 x 10 square of the red channel (visualisation.py:20-25)
    assert np.array_equal(masks[0].cpu().numpy(), tm[0]['time_mask'])
    from PIL import Image
    png = next(p for p in (tmp_path / "cam_saved_images").rglob("case7pert3.png"))
    px = np.asarray(Image.open(png))
    assert px.shape == (120, 160, 3) and (px[:10, :10, 0] == np.uint8(np.float32(tm[0]['time_mask'][3]) * 255)).all()
    assert (px[:10, :10, 1:] == 0).all()
    if backbone == "i3d":
        assert len(gc) == 2 and gc[0]['GCHeatMap'].shape == (32, 120, 160) and gc[0]['GCHeatMap'].dtype == np.float32
    else:
        assert gc == []


def test_sharded_searches_equal_unsharded():
    """SURVEY 8e on one GPU: the clip_id % 2 shards run one after the other (own plans, rank 0's tuning
    vector installed through get_tuning/set_tuning, as bench.py broadcasts it), pack_records -> merge
    must equal the unsharded run bit for bit."""
    import ivf_engine
    import ivf_recipe as R
    import ivf_search
    import ivf_shard
    T, n = 16, 4
    sd = R.i3d_state_dict(num_classes=174)
    ids = list(range(n))

    def run(eng, clip_ids):
        x = torch.from_numpy(np.stack([R.clip(c % 16) for c in clip_ids])).cuda()
        s = ivf_search.MaskSearch(eng, 0.01, 0.02, 6, "freeze", do_gradcam=False)
        res = s.run(x, [R.label(c, 174) for c in clip_ids])
        return ivf_search.pack_records(clip_ids, res, T)

    full = ivf_engine.I3DEngine(174, (3, T, 224, 224), max_batch=n, softmax=True)
    full.load_state_dict(sd, autotune=True)        # autotunes at batch n ("rank 0")
    want = ivf_shard.gather_records(run(full, ids))
    tuning = full.get_tuning()
    parts = []
    for rank in range(2):
        eng = ivf_engine.I3DEngine(174, (3, T, 224, 224), max_batch=n // 2, softmax=True)
        eng.load_state_dict(sd, autotune=False)
        eng.set_tuning(tuning)
        parts.append(run(eng, ivf_shard.shard_ids(ids, rank, 2)))
    got = ivf_shard.gather_records(torch.cat(parts))
    assert got.dtype == torch.int32 and torch.equal(got, want)
    d = ivf_search.unpack_record(got[3], T)
    assert d["clip_id"] == 3 and d["time_mask"].shape == (T,) and 0 < d["freeze_score"] < 1


def test_drivers_run_on_a_jpeg_folder(tmp_path, monkeypatch):
    """main() of the smth driver on a JPEG clip folder laid out as the reference's PicDatabase expects
    (data_parser.py:121-131), decoded on the host and cast/permuted on the device."""
    from PIL import Image
    import FindMasksComparison_I3D_smth as drv
    import ivf_ingest
    import ivf_recipe as R
    monkeypatch.chdir(tmp_path)
    root = tmp_path / "data" / "validation"
    for cls, cid in ((3, 101), (5, 202)):
        d = root / str(cls) / str(cid)
        d.mkdir(parents=True)
        clip = R.clip(cid % 16).astype(np.uint8)                      # [3,T,H,W]
        for t in range(16):
            Image.fromarray(np.ascontiguousarray(clip[:, t].transpose(1, 2, 0)), 'RGB').save(
                d / "frame{:02d}.jpg".format(t + 1), quality=90)
    loader = ivf_ingest.JpegFolderLoader(str(root), clip_size=16, batch_size=2, layout="smth")
    assert len(loader) == 1
    seq, label, ids = next(iter(loader))
    assert seq.is_cuda and tuple(seq.shape) == (2, 3, 16, 224, 224) and seq.dtype == torch.float32
    assert sorted(label.tolist()) == [3, 5] and sorted(ids) == ["101", "202"]
    assert float(seq.min()) >= 0 and float(seq.max()) <= 255 and torch.equal(seq, seq.round())
    cfg = tmp_path / "cfg.py"
    cfg.write_text("config = " + repr({
        "conv_model": "models.I3D_doubled", "num_classes": 174, "batch_size": 2, "clip_size": 16,
        "data_folder": str(tmp_path / "data"), "num_workers": 0, "shuffle": 0}))
    drv.main(["-c", str(cfg), "--msl", "", "--optIter", "3", "--subDir", "jpg"])
    tm = pickle.load(open(tmp_path / "results" / "allTimeMaskResults_jpg_None_.p", "rb"))
    assert sorted(r['video_id'] for r in tm) == ["101", "202"] and sorted(r['true_class'] for r in tm) == [3, 5]


def test_gradcam_video_other_layer_and_refusals(model, golden):
    import ivf_lib as L
    import ivf_recipe as R
    from grad_cam_videos import GradCamVideo
    g = golden('gradcam_layers')
    x = torch.from_numpy(R.clip(11))[None].cuda()
    gc = GradCamVideo(model=model, target_layer_names=['Mixed_4f'], class_dict=None, use_cuda=True,
                      input_spatial_size=(224, 224), normalizePerFrame=True, archType="I3D")
    cam, output = gc(x, None)
    assert np.max(np.abs(cam[:, ::8, ::8] - g['Mixed_4f_cam_small'])) < 1e-3
    feats, _ = gc.extractor(x)
    assert tuple(feats[0].shape) == (1, 832, 4, 14, 14)
    for bad in (['Mixed_4f', 'Mixed_5c'], ['Mixed_4f.b0'], ['nope']):
        with pytest.raises(L.IvfError):
            GradCamVideo(model=model, target_layer_names=bad, class_dict=None, use_cuda=True,
                         input_spatial_size=(224, 224), archType="I3D")(x, None)
