"""Visualisation blend on the GPU: package `visualisation` (csrc/viz.hip) against the
arrays the reference's own create_image_arrays / vizualize_results_on_gradcam produced
(tests/golden/viz.npz) -- uint8 outputs, bit-exact."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _viz_inputs(tag, T, H, W):
    import ivf_recipe as R
    x = torch.from_numpy(R.uniform(f'g/viz/{tag}/x', (2, 3, T, H, W), 0, 255)).round()
    cam = R.uniform(f'g/viz/{tag}/cam', (T, H, W), 0, 1).astype(np.float32)
    cam[3] = 0.0
    cam[5, :4] = 1.0
    tm = torch.from_numpy(R.uniform(f'g/viz/{tag}/tm', (T,), 0, 1))
    tm[2] = 0.5
    return x, cam, tm


@pytest.mark.parametrize("tag,shape,kinds", [('a', (8, 14, 224), ('freeze', 'reverse')),
                                             ('b', (32, 12, 160), ('freeze',))])
def test_create_image_arrays_vs_reference(tag, shape, kinds, golden, tmp_path):
    import visualisation as viz
    g = golden('viz')
    T, H, W = shape
    x, cam, tm = _viz_inputs(tag, T, H, W)
    for kind in kinds:
        m = tm.clone().cuda()
        out_dir = tmp_path / kind
        img = viz.create_image_arrays(x.cuda(), cam, m, 1, kind, str(out_dir), "tag", 0, W, H)
        assert img.shape == (3, T, H, 3 * W) and img.dtype == np.uint8
        # the reference snaps its HOST copy of a CUDA mask (visualisation.py:39,77-81): the caller's tensor keeps
        # its values, the MASKVALS file shows the snapped ones
        assert np.array_equal(m.cpu().numpy(), tm.numpy())
        snapped = torch.from_numpy(g[f'{tag}_{kind}_mask_after'])
        assert open(out_dir / f"MASKVALScase{kind}tag.txt").read() == str(snapped)
        if tag == 'a':
            assert np.array_equal(img, g[f'a_{kind}_img'])
        else:
            assert np.array_equal(img[..., 2 * W:], g[f'b_{kind}_panel3'])
            assert int(img[..., :2 * W].astype(np.int64).sum()) == int(g[f'b_{kind}_sum12'])
        names = sorted(os.listdir(out_dir))
        assert "img01.jpg" in names and "mygif.gif" in names and f"MASKVALScase{kind}tag.txt" in names
        assert f"case{kind}tag_{T - 1}.png" in names


def test_viz_guards_and_nan_maps(tmp_path):
    """RESIZE_FLAG != 0 is refused; an all-zero Grad-CAM block (0/0 = NaN after normalisation,
    grad_cam_videos.py:129-132) maps to colour index 0 like numpy's uint8 cast on the reference's host."""
    import ivf_lib as L
    import visualisation as viz
    x = torch.rand(1, 3, 4, 8, 224).cuda() * 255
    cam = np.full((4, 8, 224), np.nan, dtype=np.float32)
    m = torch.tensor([0.2, 0.7, 0.6, 0.1]).cuda()
    with pytest.raises(L.IvfError):
        viz.create_image_arrays(x, cam, m, 0, "freeze", str(tmp_path), "t", 1, 224, 8)
    img = viz.create_image_arrays(x, cam, m, 0, "freeze", str(tmp_path), "t", 0, 224, 8)
    assert img.shape == (3, 4, 8, 672) and torch.equal(m.cpu(), torch.tensor([0.2, 0.7, 0.6, 0.1]))
    lut0 = viz.jet_lut_bgr()[0].astype(np.float32)
    base = np.flip(x[0].cpu().numpy().transpose(1, 2, 3, 0), 3)                    # [T,H,W,3] BGR
    f0 = (lut0 + base[0])
    want = np.uint8(255 * (f0 / f0.max()))
    assert np.array_equal(img[:, 0, :, 224:448].transpose(1, 2, 0), want)
