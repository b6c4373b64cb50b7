"""SURVEY 8f N4 (documented extension, PARITY UNPINNED): the TF-style ConvLSTM classifier, its mask search and
per-frame Grad-CAM on libivf_hip against the torch restatement of the Keras ConvLSTM2D arithmetic
(oracle/tfclstm_ref.py).  TensorFlow 1.12 / Keras cannot be installed in the build container, so these tests pin
the extension's internal consistency (forward, BPTT vs autograd, search, Grad-CAM), not parity with TF."""
import numpy as np
import pytest
import torch

from conftest import note, rel_err

pytestmark = pytest.mark.gpu


def _weights(C, units, kh, kw, fc_in, K, seed=3):
    g = torch.Generator().manual_seed(seed)
    layers, cin = [], C
    for Fu in units:
        layers.append((torch.randn(kh, kw, cin, 4 * Fu, generator=g) * 0.25, torch.randn(kh, kw, Fu, 4 * Fu, generator=g) * 0.2,
                       torch.randn(4 * Fu, generator=g) * 0.1))
        cin = Fu
    return dict(layers=layers, dense_w=torch.randn(fc_in, K, generator=g) * 0.2, dense_b=torch.randn(K, generator=g) * 0.1)


def _setup(C=1, T=8, H=30, W=40, units=(4, 6), kernel=(3, 5), stride=2, padding="valid", act="hard_sigmoid", only_last=True,
           B=2, K=5):
    import ivf_engine
    eng = ivf_engine.TFCLSTMEngine(K, (C, T, H, W), units=units, kernel=kernel, stride=stride, padding=padding,
                                   recurrent_activation=act, only_last_element_for_fc=only_last, max_batch=B)
    w = _weights(C, units, kernel[0], kernel[1], eng.fc_inputs, K)
    eng.load_weights(w["layers"], w["dense_w"], w["dense_b"])
    g = torch.Generator().manual_seed(11)
    x = torch.rand(B, C, T, H, W, generator=g)
    kw = dict(stride=stride, padding=padding, hard=(act == "hard_sigmoid"), only_last=only_last)
    return eng, w, x, kw


@pytest.mark.parametrize("padding,act,only_last,C", [("valid", "hard_sigmoid", True, 1), ("same", "hard_sigmoid", False, 3),
                                                      ("valid", "sigmoid", False, 1), ("same", "sigmoid", True, 1)])
def test_tfclstm_forward_and_bptt(padding, act, only_last, C):
    from oracle import tfclstm_ref as ref
    eng, w, x, kw = _setup(C=C, padding=padding, act=act, only_last=only_last)
    probs, logits = eng.forward(x.cuda(), want_logits=True)
    xr = x.clone().requires_grad_()
    lg, _ = ref.model(xr, w, **kw)
    pr = torch.softmax(lg, 1)
    assert rel_err(logits.cpu().numpy(), lg.detach().numpy()) < 1e-4
    assert rel_err(probs.cpu().numpy(), pr.detach().numpy()) < 1e-4
    tgt = [1, 3]
    (pr[0, tgt[0]] + pr[1, tgt[1]]).backward()
    score, dx = eng.backward(2, tgt)
    assert abs(float(score[0]) - float(pr[0, tgt[0]])) < 1e-5
    e = rel_err(dx.cpu().numpy(), xr.grad.numpy())
    note(f"tf-style clstm ({padding}, {act}, only_last={only_last}, C={C}): logits {rel_err(logits.cpu().numpy(), lg.detach().numpy()):.1e}, "
         f"BPTT dx vs autograd {e:.1e}")
    assert e < 1e-3


def test_tfclstm_geometry_follows_tensorflow_rules():
    import ivf_engine
    eng = ivf_engine.TFCLSTMEngine(6, (1, 32, 120, 160), units=(32, 32), kernel=(3, 5), stride=2, padding="valid", max_batch=1)
    assert eng.layer_dims(0) == (59, 78, 29, 39, 32)          # (120-3)//2+1, (160-5)//2+1, then 2x2 pool
    assert eng.layer_dims(1) == (14, 18, 7, 9, 32)
    assert eng.fc_inputs == 7 * 9 * 32                        # only_last_element_for_fc == 'yes'
    eng = ivf_engine.TFCLSTMEngine(6, (1, 32, 120, 160), units=(8,), kernel=(3, 5), stride=2, padding="same",
                                   only_last_element_for_fc=False, max_batch=1)
    assert eng.layer_dims(0) == (60, 80, 30, 40, 8) and eng.fc_inputs == 32 * 30 * 40 * 8


def test_tfclstm_search_and_gradcam():
    """init_mask through the graph's sigmoid, 6 iterations with tf.train.Adam's update, the final mask, and the
    per-frame Grad-CAM in both normalisation modes, against the torch restatement."""
    import ivf_tf_search
    from oracle import tfclstm_ref as ref
    eng, w, x, kw = _setup(C=1, T=8, units=(4, 6), only_last=False, B=2)
    xg = x.cuda()
    labels = [1, 3]
    res = ivf_tf_search.find_mask(eng, xg, labels, lam1=0.01, lam2=0.02, n_iter=6, lr=0.2, focus_type="correct",
                                  normalization_mode="frame")
    for bi in range(2):
        xi = x[bi:bi + 1]
        init = ref.init_mask_central(xi, w, labels[bi], **kw)
        assert np.array_equal(res["init_mask"][bi].cpu().numpy(), init.numpy())
        traj, final = ref.search(xi, w, labels[bi], init, 0.01, 0.02, 6, lr=0.2, **kw)
        got = res["traj"][:, bi].cpu().numpy()
        assert np.max(np.abs(got - traj) / np.abs(traj[:, :1])) < 1e-3
        assert np.max(np.abs(res["time_mask"][bi].cpu().numpy() - final.numpy())) < 1e-3
        half = torch.full((8,), 0.5)
        for mode in ("frame", "sequence"):
            cam, probs = eng.gradcam(xg[bi:bi + 1], [labels[bi]], mask=half[None].cuda(), normalization_mode=mode)
            want, pr = ref.gradcam_frames(xi, w, labels[bi], mask=half, per_frame=(mode == "frame"), **kw)
            g = cam[0].cpu().numpy()
            assert np.array_equal(np.isnan(g), np.isnan(want))
            ok = ~np.isnan(want)
            assert rel_err(probs.cpu().numpy(), pr.numpy()) < 1e-4
            if ok.any():
                assert np.max(np.abs(g[ok] - want[ok])) < 2e-3, mode
    note(f"tf-style clstm search: 6-iteration trajectory and final mask vs the torch restatement within 1e-3 (unpinned vs TF)")
    with pytest.raises(Exception):
        eng.gradcam(xg[:1], [0], normalization_mode="clip")
