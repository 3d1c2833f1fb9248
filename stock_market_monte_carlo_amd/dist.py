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


def gather_stats_tensor(record, group=None, force=False):
    """record: uint8 tensor holding this rank's packed statistics record (device tensor
    under nccl, CPU tensor under gloo).  ONE all_gather (skipped for a one-rank group unless
    `force`: a one-GPU rehearsal of the collective, bench.py --rehearse-rccl); returns the
    gathered tensor [world x record bytes] WHERE THE COLLECTIVE LEFT IT -- under nccl (RCCL) in
    device memory, enqueued on the current stream, no host synchronisation: a stepping loop
    can keep launching (the per-step `.cpu()` serialised kernel and collective).
    records_from_tensor() brings it to the host when someone wants to read it."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    if world == 1 and not force:
        return record.reshape(1, -1)
    if dist.get_backend(group) == "gloo" and record.is_cuda:
        record = record.cpu()  # gloo (CPU rehearsals of the N > 1 path) gathers host tensors
    gathered = torch.empty(world * record.numel(), dtype=torch.uint8, device=record.device)
    dist.all_gather_into_tensor(gathered, record.contiguous(), group=group)
    return gathered.reshape(world, -1)


def records_from_tensor(gathered):
    """The ranks' records as bytes, in rank order (copies to the host: synchronises)."""
    raw = gathered.cpu().numpy()
    return [raw[i].tobytes() for i in range(raw.shape[0])]


def gather_stats_records(record, group=None, force=False):
    """gather_stats_tensor + records_from_tensor: the list of all ranks' records as bytes."""
    return records_from_tensor(gather_stats_tensor(record, group, force))


def all_gather_merge_stats(record, group=None):
    """Gathers every rank's record and merges them in rank order -> engine.Stats."""
    return stats_from_bytes(merge_stats_bytes(gather_stats_records(record, group)))
