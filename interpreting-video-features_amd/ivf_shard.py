"""Sharding of independent per-clip searches over the GPUs of a node.

One process per GPU (torchrun); clip c belongs to rank c % world; no collective
on the data path (eval-mode rows are independent).  Finished
fixed-size records are exchanged by an all_gather (RCCL over xGMI on the GPU box,
gloo in the CPU tests).  Payload is ~100 B per clip, so the point-to-point xGMI
links are never the limit; load balance is, and every search costs the same
(N fixed, no early exit: SURVEY.md F9).

A record row is int32: column 0 the clip id, integer fields as they are, float
fields bit-cast (`ivf_search.pack_records`), so ids and classes are exact at any
magnitude and sorting / filtering run on integers.
"""
import torch
import torch.distributed as dist


def shard_ids(clip_ids, rank, world):
    """Round-robin ownership: clip_id % world == rank (ids keep their order)."""
    return [c for c in clip_ids if c % world == rank]


def _sort_rows(rec, col):
    if rec.shape[0] == 0:
        return rec
    return rec[torch.argsort(rec[:, col].to(torch.int64), stable=True)]


def gather_records(records, clip_id_col=0, equal_shards=False):
    """records [n_local, R] int32 (column `clip_id_col` = clip id >= 0) on this rank's device
    -> [n_total, R] on every rank, sorted by clip id.

    equal_shards=True (every rank holds the same number of rows, e.g. bench.py's fixed clips
    per GPU): ONE all_gather_into_tensor, no host sync.  Otherwise shards may be uneven or
    empty: a first small all_gather exchanges the row counts, rows are padded to the largest
    shard with clip id -1 and dropped afterwards (two collectives)."""
    if records.dtype != torch.int32:
        raise TypeError("records must be int32 rows (ivf_search.pack_records)")
    if not (dist.is_available() and dist.is_initialized()):
        return _sort_rows(records, clip_id_col)
    world = dist.get_world_size()
    R = records.shape[1]
    if equal_shards:
        padded = records.contiguous()
        rows = padded.shape[0]
    else:
        n = torch.tensor([records.shape[0]], device=records.device, dtype=torch.int64)
        counts = [torch.zeros_like(n) for _ in range(world)]
        dist.all_gather(counts, n)
        rows = max(max(int(c) for c in counts), 1)
        padded = torch.full((rows, R), -1, device=records.device, dtype=torch.int32)
        padded[:records.shape[0]] = records
    out = torch.empty(world * rows, R, device=records.device, dtype=torch.int32)
    if records.is_cuda:
        dist.all_gather_into_tensor(out, padded)      # RCCL
    else:
        parts = [torch.empty_like(padded) for _ in range(world)]
        dist.all_gather(parts, padded)                # gloo has no all_gather_into_tensor
        out.copy_(torch.cat(parts))
    if not equal_shards:
        out = out[out[:, clip_id_col] >= 0]
    return _sort_rows(out, clip_id_col)
