"""CPU: the C-ABI shared library loads and exports every symbol that
include/ivf_hip.h declares (no compute calls without a GPU), argument checks
return error codes rather than crashing, and the product refuses to run
without a GPU instead of falling back."""
import ctypes
import os
import re

import pytest
import torch

from conftest import ROOT


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "ivf_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ivf_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported():
    import ivf_lib
    lib = ctypes.CDLL(ivf_lib.LIB_PATH)
    names = _declared_symbols()
    assert len(names) >= 40
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/ivf_hip.h but not exported"
    # and the python binding covers exactly the declared set
    assert set(ivf_lib.exported_symbols()) == set(names)


def test_bad_arguments_return_error_codes():
    import ivf_lib as L
    lib = L.lib()
    assert lib.ivf_version() >= 100
    rc = lib.ivf_freeze_fwd(None, None, None, 1, 3, 16, 10, 0, 0, None)
    assert rc == -1 and b"freeze_fwd" in lib.ivf_last_error()
    d = L.ConvDesc()
    assert lib.ivf_conv3d(ctypes.byref(d), None, None, None, None, None, None, None) == -1
    cfg = L.I3DConfig()
    h = ctypes.c_void_p()
    assert lib.ivf_i3d_create(ctypes.byref(cfg), ctypes.byref(h)) == -1


def test_plan_is_host_only_and_sized():
    """Plan construction touches no device: shapes and arena sizes can be checked on CPU."""
    import ivf_lib as L
    lib = L.lib()
    cfg = L.I3DConfig()
    cfg.B, cfg.C, cfg.T, cfg.H, cfg.W = 2, 3, 16, 224, 224
    cfg.num_classes, cfg.stem_stride_t, cfg.pool4a_stride_t, cfg.pool5a_stride_t = 174, 2, 2, 2
    cfg.head_kt, cfg.head_kh, cfg.head_kw, cfg.softmax = 2, 7, 7, 1
    h = ctypes.c_void_p()
    L.check(lib.ivf_i3d_create(ctypes.byref(cfg), ctypes.byref(h)))
    assert lib.ivf_i3d_num_convs(h) == 58                      # 57 BN units + logits
    assert abs(lib.ivf_i3d_conv_flops_per_clip(h) / 2e9 - 27.788) < 0.01   # GMAC, SURVEY Appendix A
    assert lib.ivf_i3d_weights_bytes(h) > 2 * 49e6              # fwd + bwd packs of 12.47 M params
    assert 2 * 200e6 < lib.ivf_i3d_workspace_bytes(h) < 2 * 400e6
    # a head window that does not cover Mixed_5c is refused loudly
    cfg.T = 32
    h2 = ctypes.c_void_p()
    assert lib.ivf_i3d_create(ctypes.byref(cfg), ctypes.byref(h2)) == -3
    lib.ivf_i3d_destroy(h)


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU refusal")
def test_no_cpu_fallback():
    import ivf_lib as L
    import mask
    from models import I3D_doubled
    with pytest.raises(L.IvfError):
        mask.perturb_sequence(torch.zeros(1, 3, 16, 4, 4), torch.zeros(16))
    m = I3D_doubled.Model(10, stride_mod_layers="", softMax=1).eval()
    with pytest.raises(L.IvfError):
        m(torch.zeros(1, 3, 16, 224, 224))


def test_jpeg_folder_listing_follows_the_reference_layouts(tmp_path):
    """Host logic of the clip loader (no device needed): PicDatabase / KTHImLoader folder layouts,
    drop_last batching (smth:71-77, KTH:73-78)."""
    import ivf_ingest
    smth = tmp_path / "smth"
    for cls, clips in ((7, (11, 12, 13)), (9, (21,))):
        for c in clips:
            (smth / str(cls) / str(c)).mkdir(parents=True)
    ld = ivf_ingest.JpegFolderLoader(str(smth), clip_size=16, batch_size=3, layout="smth")
    assert sorted((lab, cid) for _, lab, cid in ld.items) == [(7, "11"), (7, "12"), (7, "13"), (9, "21")]
    assert len(ld) == 1                                   # drop_last
    assert len(ivf_ingest.JpegFolderLoader(str(smth), 16, 3, "smth", drop_last=False)) == 2
    kth = tmp_path / "kth"
    for i in range(5):
        (kth / str(i)).mkdir(parents=True)
    ld = ivf_ingest.JpegFolderLoader(str(kth), clip_size=32, batch_size=2, layout="kth")
    assert [os.path.basename(f) for f, _, _ in ld.items] == ["0", "1", "2", "3", "4"] and len(ld) == 2
    with pytest.raises(FileNotFoundError):
        ivf_ingest.JpegFolderLoader(str(tmp_path / "missing"), 16, 2, "smth")
