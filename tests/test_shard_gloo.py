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


def _fake_records(ids, T=16):
    """Deterministic stand-in for a finished search record (clip id + 6 scalars + mask)."""
    rows = []
    for c in ids:
        g = torch.Generator().manual_seed(1000 + c)
        rows.append(torch.cat([torch.tensor([float(c)]), torch.rand(6 + T, generator=g)]))
    return torch.stack(rows) if rows else torch.empty(0, 7 + T)


def _worker(rank, world, port, n_clips, out_dir):
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
    got = ivf_shard.gather_records(_fake_records(ids))
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
    want = _fake_records(list(range(n_clips)))
    for r in range(world):
        got = torch.load(tmp_path / f"r{r}.pt")
        assert torch.equal(got, want), f"rank {r}"


def test_single_process_gather_sorts():
    import ivf_shard
    rec = _fake_records([5, 2, 9])
    out = ivf_shard.gather_records(rec)
    assert out[:, 0].tolist() == [2.0, 5.0, 9.0]
