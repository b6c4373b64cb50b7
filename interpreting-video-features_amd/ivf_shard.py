"""Sharding of independent per-clip searches over the GPUs of a node.

One process per GPU (torchrun); clip c belongs to rank c % world; no collective
on the data path (eval-mode rows are independent).  Finished
fixed-size records are exchanged with ONE all_gather (RCCL over xGMI on the GPU
box, gloo in the CPU tests).  Payload is ~100 B per clip, so the point-to-point
xGMI links are never the limit; load balance is, and every search costs the same
(N fixed, no early exit: SURVEY.md F9).
"""
import torch
import torch.distributed as dist


def shard_ids(clip_ids, rank, world):
    """Round-robin ownership: clip_id % world == rank (ids keep their order)."""
    return [c for c in clip_ids if c % world == rank]


def gather_records(records, clip_id_col=0):
    """records [n_local, R] float32 (column `clip_id_col` = clip id) on this rank's
    device -> [n_total, R] on every rank, sorted by clip id.  Shards may be uneven or
    empty: rows are padded to the largest shard with clip id -1 and dropped after."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        order = torch.argsort(records[:, clip_id_col], stable=True) if records.shape[0] else torch.arange(0)
        return records[order]
    world = dist.get_world_size()
    n = torch.tensor([records.shape[0]], device=records.device, dtype=torch.int64)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n)
    nmax = int(max(int(c) for c in counts))
    R = records.shape[1]
    padded = torch.full((max(nmax, 1), R), -1.0, device=records.device, dtype=records.dtype)
    padded[:records.shape[0]] = records
    out = torch.empty(world * max(nmax, 1), R, device=records.device, dtype=records.dtype)
    dist.all_gather_into_tensor(out, padded) if records.is_cuda else \
        out.copy_(torch.cat(_all_gather_list(padded, world)))
    out = out[out[:, clip_id_col] >= 0]
    return out[torch.argsort(out[:, clip_id_col], stable=True)]


def _all_gather_list(t, world):
    parts = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(parts, t)
    return parts
