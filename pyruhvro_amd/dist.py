"""Multi-GPU plumbing: the path shards by independent records, so ranks never exchange data.

One process per GPU (torch.distributed; backend "nccl" is RCCL on ROCm, "gloo" in CPU tests).
The only collectives are the timing barrier, a MAX over per-rank wall times and an all-gather of a
small per-rank stats vector (records, bytes, kernel milliseconds) so rank 0 can report whole-job
throughput -- SURVEY.md section 8(e).
"""
from __future__ import annotations

from typing import Dict, List, Tuple

STAT_KEYS = ("records", "input_bytes", "output_bytes", "size_kernel_ms", "scan_kernel_ms", "emit_kernel_ms",
             "step_ms")


def shard_rows(n_per_rank: int, rank: int) -> Tuple[int, int]:
    """Weak scaling: rank r decodes rows [r*n, (r+1)*n) of the synthetic stream."""
    return rank * n_per_rank, (rank + 1) * n_per_rank


def strong_shard(n: int, num_chunks: int, world: int, rank: int) -> Dict[str, int]:
    """One list of n records over `world` GPUs, one process per GPU (BASELINE config 5): this rank's rows and the
    chunk geometry to hand rh_decode_device (rh_opts.chunk_rows) so that its batches are exactly the reference's
    chunks [c0, c1) of the WHOLE list (rh_shard_chunks in the C ABI -- the same deal rh_decode makes for a device
    list inside one process)."""
    from . import cabi
    c0, c1, r0, r1 = cabi.shard_chunks(n, num_chunks, world, rank)
    k = max(1, min(max(num_chunks, 1), max(n, 1)))
    return {"row_lo": r0, "rows": r1 - r0, "chunks": c1 - c0, "chunk_rows": n // k if c1 > c0 else 0,
            "chunk_lo": c0, "chunk_hi": c1}


def partition_chunks(n: int, num_chunks: int, world: int) -> List[List[Tuple[int, int]]]:
    """Strong-scaling form (one list over `world` GPUs): chunk boundaries of the reference
    (deserialize.rs:53-68) are kept, whole chunks are dealt to GPUs in contiguous runs, so every
    returned batch is produced by exactly one GPU.  -> per-rank list of (row_lo, row_hi)."""
    k = max(1, min(max(num_chunks, 1), max(n, 1)))
    sz = n // k
    bounds = [(i * sz, n if i == k - 1 else (i + 1) * sz) for i in range(k)]
    out: List[List[Tuple[int, int]]] = []
    for r in range(world):
        lo, hi = r * k // world, (r + 1) * k // world
        out.append(bounds[lo:hi])
    return out


def init_process_group(backend: str, device=None):
    """`device`: this rank's GPU (backend "nccl" = RCCL): handed to torch so the communicator is bound to it from the
    start and barriers do not have to guess the device."""
    import torch.distributed as dist
    if not dist.is_initialized():
        if device is not None and backend == "nccl":
            dist.init_process_group(backend=backend, device_id=device)
        else:
            dist.init_process_group(backend=backend)
    return dist.get_rank(), dist.get_world_size()


def max_over_ranks(value: float, device) -> float:
    import torch
    import torch.distributed as dist
    t = torch.tensor([value], dtype=torch.float64, device=device)
    if dist.is_initialized():                  # (a one-rank group too: bench.py BENCH_FORCE_DIST runs the real collectives on one GPU)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_stats(local: Dict[str, float], device) -> List[Dict[str, float]]:
    """all_gather of the per-rank stats vector (the only payload that crosses xGMI)."""
    import torch
    import torch.distributed as dist
    v = torch.tensor([float(local.get(k, 0.0)) for k in STAT_KEYS], dtype=torch.float64, device=device)
    if dist.is_initialized():
        outs = [torch.zeros_like(v) for _ in range(dist.get_world_size())]
        dist.all_gather(outs, v)
    else:
        outs = [v]
    return [{k: float(o[i]) for i, k in enumerate(STAT_KEYS)} for o in outs]


def aggregate(per_rank: List[Dict[str, float]], steps: int, wall_s: float) -> Dict[str, float]:
    """Whole-job numbers from the gathered vectors: units all ranks processed / max wall time."""
    recs = sum(r["records"] for r in per_rank) * steps
    return {
        "records_per_s": recs / wall_s if wall_s > 0 else 0.0,
        "records_total": recs,
        "input_bytes": sum(r["input_bytes"] for r in per_rank),
        "output_bytes": sum(r["output_bytes"] for r in per_rank),
        "emit_kernel_ms_max": max(r["emit_kernel_ms"] for r in per_rank),
        "size_kernel_ms_max": max(r["size_kernel_ms"] for r in per_rank),
    }
