"""Clip ingest kernel through the C-ABI: bit-exact against the reference
loaders' output (tests/golden/ingest.npz) and against the oracle on ragged sizes."""
import os
import tempfile

import numpy as np
import pytest
import torch

from oracle import ingest_ref

pytestmark = pytest.mark.gpu


def test_ingest_equals_reference_loader_output(golden):
    import ivf_ingest
    g = golden('ingest')
    T = int(g['shape'][0])
    with tempfile.TemporaryDirectory() as root:
        for i in range(T):
            with open(os.path.join(root, 'frame{:02d}.jpg'.format(i + 1)), 'wb') as f:
                f.write(g[f'jpeg{i}'].tobytes())
        frames = ivf_ingest.decode_clip_u8(root, T)
    x = ivf_ingest.ingest_u8(frames)
    assert x.is_cuda and x.dtype == torch.float32
    assert np.array_equal(x.cpu().numpy(), g['kth_data'])
    assert np.array_equal(x.cpu().numpy(), g['smth_data'])
    cl = ivf_ingest.ingest_u8(frames, layout=ivf_ingest.CHANNELS_LAST, cpad=4)
    assert np.array_equal(cl.cpu().numpy(), ingest_ref.to_channels_last(frames, 4))


@pytest.mark.parametrize("shape", [(2, 3, 7, 9, 3), (1, 16, 224, 224, 3), (3, 2, 5, 6, 1), (2, 4, 6, 8, 4),
                                   (1, 32, 120, 160, 3)])
def test_ingest_ragged_and_full_sizes(shape):
    import ivf_ingest
    rng = np.random.default_rng(3)
    frames = rng.integers(0, 256, size=shape, dtype=np.uint8)
    x = ivf_ingest.ingest_u8(frames)
    want = np.stack([ingest_ref.to_model_input(f) for f in frames])
    assert np.array_equal(x.cpu().numpy(), want)
    for cpad in (4, 8):
        cl = ivf_ingest.ingest_u8(torch.from_numpy(frames).cuda(), layout=ivf_ingest.CHANNELS_LAST, cpad=cpad)
        assert np.array_equal(cl.cpu().numpy(), ingest_ref.to_channels_last(frames, cpad))


def test_ingest_feeds_the_plan_like_the_float_clip():
    """A clip ingested from uint8 gives the same logits as the same clip handed over as the
    reference's float tensor."""
    import ivf_ingest
    import ivf_recipe as R
    from ivf_engine import I3DEngine
    eng = I3DEngine(174, (3, 16, 224, 224), max_batch=1)
    eng.load_state_dict(R.i3d_state_dict(num_classes=174), autotune=False)
    clip = R.clip(5)                                          # integer-valued floats, like decoded JPEG
    u8 = np.ascontiguousarray(clip.transpose(1, 2, 3, 0)).astype(np.uint8)
    a = eng.forward(torch.from_numpy(clip)[None].cuda())
    b = eng.forward(ivf_ingest.ingest_u8(u8)[None])
    assert torch.equal(a[0], b[0])
