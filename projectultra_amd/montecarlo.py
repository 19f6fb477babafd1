"""Monte-Carlo harness over the batched receive path (shape of tools/test_nvis_mode.cpp:35-114
scaled to millions of independent frames): frames shard embarrassingly across the ranks of
one node (one process per GPU), nothing crosses GPUs on the data path, and ONE all-reduce of
the eight uint64 counters (64 bytes, latency-bound) merges the BER/FER statistics — RCCL over
xGMI when the backend is "nccl", gloo in the CPU tests."""
from __future__ import annotations

from typing import Callable, Tuple

from ._lib import COUNTER_NAMES


def shard_range(n_frames: int, rank: int, world_size: int) -> Tuple[int, int]:
    """Contiguous frame range [lo, hi) owned by `rank`."""
    if world_size < 1 or not (0 <= rank < world_size):
        raise ValueError("bad rank/world_size")
    return n_frames * rank // world_size, n_frames * (rank + 1) // world_size


def allreduce_counters(counters, group=None):
    """Sum the counter vector over all ranks in place (single collective); returns it."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(counters, op=dist.ReduceOp.SUM, group=group)
    return counters


def counters_dict(counters) -> dict:
    vals = [int(v) for v in counters.tolist()]
    d = dict(zip(COUNTER_NAMES, vals))
    d["fer"] = d["frame_errors"] / d["frames"] if d["frames"] else float("nan")
    d["ber"] = d["bit_errors"] / d["info_bits"] if d["info_bits"] else float("nan")
    d["mean_iters"] = d["iters_sum"] / d["frames"] if d["frames"] else float("nan")
    d["undetected_rate"] = d["undetected_errors"] / d["frames"] if d["frames"] else float("nan")
    return d


def run_sharded(n_frames: int, rank: int, world_size: int, run_shard: Callable[[int, int], "object"], group=None):
    """run_shard(lo, hi) -> int64[8] counter tensor for frames [lo, hi); returns the global sums."""
    lo, hi = shard_range(n_frames, rank, world_size)
    counters = run_shard(lo, hi)
    return allreduce_counters(counters, group=group)
