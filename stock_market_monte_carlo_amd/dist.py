"""Multi-GPU sharding of the path space and the one collective the path needs.

One process per GPU (torch.distributed; backend "nccl" is RCCL on ROCm, "gloo" in CPU
tests).  Paths are independent, so the data path has no collective: rank r simulates a
contiguous range of global path ids.  The only exchange is the statistics record
(64 + 8 * n_bins bytes): one all_gather of the raw records, merged on every rank in rank
order, so the merged sums do not depend on arrival order.  At ~1 KB the collective is
latency-bound; xGMI bandwidth plays no role and nothing is bucketed or pipelined.

Fixes two defects of the reference's multi-GPU launcher on the way: every GPU there runs
the same thread ids with the same seeds (src/simulations.cu:120,140) and the N mod G
remainder is dropped (:602-603).
"""
import numpy as np

from .engine import merge_stats_bytes, stats_from_bytes


def shard_range(n_total, world, rank):
    """(first_path, count) of `rank`: floor(N/G) paths plus one of the N mod G leftovers."""
    base, extra = divmod(int(n_total), int(world))
    count = base + (1 if rank < extra else 0)
    first = rank * base + min(rank, extra)
    return first, count


def gather_stats_records(record, group=None, force=False):
    """record: uint8 tensor holding this rank's packed statistics record (device tensor
    under nccl, CPU tensor under gloo).  Returns the list of all ranks' records as bytes,
    in rank order.  One all_gather (skipped for a one-rank group unless `force`: a one-GPU
    rehearsal of the collective, bench.py --rehearse-rccl)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    if world == 1 and not force:
        return [record.cpu().numpy().tobytes()]
    if dist.get_backend(group) == "gloo" and record.is_cuda:
        record = record.cpu()  # gloo (CPU rehearsals of the N > 1 path) gathers host tensors
    gathered = torch.empty(world * record.numel(), dtype=torch.uint8, device=record.device)
    dist.all_gather_into_tensor(gathered, record.contiguous(), group=group)
    raw = gathered.cpu().numpy().tobytes()
    n = record.numel()
    return [raw[i * n:(i + 1) * n] for i in range(world)]


def all_gather_merge_stats(record, group=None):
    """Gathers every rank's record and merges them in rank order -> engine.Stats."""
    return stats_from_bytes(merge_stats_bytes(gather_stats_records(record, group)))
