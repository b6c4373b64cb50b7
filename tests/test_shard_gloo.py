"""N>1 path on CPU: world_size-2 (and 3) gloo process groups exercise the clip
sharding and the record all_gather; the sharded result must equal the
single-process result bit for bit, for even, uneven and empty shards."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import PKG, ROOT


BIG = 1 << 25      # clip ids above 2^24 are not exact in float32: records carry them as int32


CAM = (4, 7, 7)     # Grad-CAM payload of the optional record form (any fixed [T', h, w])


def _fake_records(ids, T=16, with_cam=False):
    """Deterministic stand-in for a finished search: the record ivf_search.pack_records builds."""
    import ivf_search
    ncam = CAM[0] * CAM[1] * CAM[2] if with_cam else 0
    if not ids:
        return torch.empty(0, 7 + T + ncam, dtype=torch.int32)
    g = torch.Generator().manual_seed(1000 + ids[0])
    n = len(ids)
    res = {"pred_class": torch.randint(0, 174, (n,), generator=g), "target": torch.randint(0, 174, (n,), generator=g),
           "time_mask": torch.rand(n, T, generator=g)}
    for k in ivf_search.RECORD_FLOAT_FIELDS:
        res[k] = torch.rand(n, generator=g)
    if with_cam:
        res["gradcam"] = torch.rand(n, *CAM, generator=g)
    return ivf_search.pack_records([BIG + c for c in ids], res, T, with_cam=with_cam)


def _all_records(n_clips, world, with_cam=False):
    """What a single process holding every shard would gather."""
    import ivf_shard
    parts = [_fake_records(ivf_shard.shard_ids(list(range(n_clips)), r, world), with_cam=with_cam) for r in range(world)]
    rec = torch.cat(parts)
    return rec[torch.argsort(rec[:, 0].to(torch.int64), stable=True)]


def _worker(rank, world, port, n_clips, out_dir, with_cam=False):
    import sys
    for p in (PKG, ROOT):
        if p not in sys.path:
            sys.path.insert(0, p)
    import ivf_shard
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ids = ivf_shard.shard_ids(list(range(n_clips)), rank, world)
    assert all(c % world == rank for c in ids)
    got = ivf_shard.gather_records(_fake_records(ids, with_cam=with_cam), equal_shards=(n_clips % world == 0))
    torch.save(got, os.path.join(out_dir, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world,n_clips", [(2, 8), (2, 7), (2, 1), (3, 10)])
def test_sharded_gather_equals_single_process(world, n_clips, tmp_path):
    mp.spawn(_worker, args=(world, _free_port(), n_clips, str(tmp_path)), nprocs=world, join=True)
    import ivf_search
    want = _all_records(n_clips, world)
    assert want[:, 0].tolist() == [BIG + c for c in range(n_clips)]
    for r in range(world):
        got = torch.load(tmp_path / f"r{r}.pt")
        assert got.dtype == torch.int32 and torch.equal(got, want), f"rank {r}"
    if n_clips:
        d = ivf_search.unpack_record(want[-1], 16)
        assert d["clip_id"] == BIG + n_clips - 1 and d["time_mask"].shape == (16,)
        assert 0.0 <= d["freeze_score"] < 1.0


@pytest.mark.parametrize("world,n_clips", [(2, 6), (2, 5)])
def test_sharded_gather_with_gradcam_payload(world, n_clips, tmp_path):
    """The optional record form: Grad-CAM maps travel with the masks, bit for bit."""
    mp.spawn(_worker, args=(world, _free_port(), n_clips, str(tmp_path), True), nprocs=world, join=True)
    import ivf_search
    want = _all_records(n_clips, world, with_cam=True)
    assert want.shape[1] == 7 + 16 + CAM[0] * CAM[1] * CAM[2]
    for r in range(world):
        assert torch.equal(torch.load(tmp_path / f"r{r}.pt"), want), f"rank {r}"
    # the map of the last clip comes back as it was packed
    last = n_clips - 1
    ids = [c for c in range(n_clips) if c % world == last % world]
    g = torch.Generator().manual_seed(1000 + ids[0])
    n = len(ids)
    torch.randint(0, 174, (n,), generator=g); torch.randint(0, 174, (n,), generator=g); torch.rand(n, 16, generator=g)
    for _ in ivf_search.RECORD_FLOAT_FIELDS:
        torch.rand(n, generator=g)
    cams = torch.rand(n, *CAM, generator=g)
    d = ivf_search.unpack_record(want[-1], 16, cam_shape=CAM)
    assert d["clip_id"] == BIG + last and d["gradcam"].shape == CAM
    assert (d["gradcam"] == cams[ids.index(last)].numpy()).all()
    with pytest.raises(Exception):
        ivf_search.unpack_record(want[-1], 16, cam_shape=(3, 7, 7))


def test_single_process_gather_sorts():
    import ivf_shard
    rec = torch.cat([_fake_records([5]), _fake_records([2]), _fake_records([9])])
    out = ivf_shard.gather_records(rec)
    assert out[:, 0].tolist() == [BIG + 2, BIG + 5, BIG + 9]
    with pytest.raises(TypeError):
        ivf_shard.gather_records(rec.float())
