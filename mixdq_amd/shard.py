"""Multi-GPU story of the hot path: identical replicas, the batch sharded across ranks.

The reference has no distributed code at all.  The per-image UNet forward
is independent (per-tensor STATIC activation scales: no batch-dependent statistic), so the path
shards naturally on the batch dimension with NO collective in the step loop.  The only exchange
is at load: rank 0's quantized buffers (weight_int, scale, bias0, ... about 2.6 GB for SDXL) are
broadcast once over RCCL/xGMI in a few large buckets (xGMI is point-to-point: a broadcast is
per-link bound, so few big messages beat many small ones), which also guarantees bit-identical
replicas.  Works with any torch.distributed backend ("nccl" = RCCL on ROCm; "gloo" in CPU tests).
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def env_rank_world():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init_distributed(backend: str | None = None):
    """One process per GPU; rendezvous from RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT."""
    rank, local_rank, world = env_rank_world()
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl" and torch.cuda.device_count() > local_rank:
            torch.cuda.set_device(local_rank)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def shard_range(total: int, rank: int, world: int):
    """Contiguous, balanced [lo, hi) of `total` images for `rank` (first ranks get the remainder)."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_batch(batch: dict, rank: int, world: int):
    """Slice every tensor with a leading batch dimension (nested dicts included)."""
    sizes = [v.shape[0] for v in _leaves(batch) if v.dim() > 0]
    B = max(sizes) if sizes else 0
    lo, hi = shard_range(B, rank, world)

    def cut(v):
        if isinstance(v, dict):
            return {k: cut(x) for k, x in v.items()}
        if torch.is_tensor(v) and v.dim() > 0 and v.shape[0] == B:
            return v[lo:hi]
        return v
    return cut(batch)


def _leaves(d):
    for v in d.values():
        if isinstance(v, dict):
            yield from _leaves(v)
        elif torch.is_tensor(v):
            yield v


@torch.no_grad()
def broadcast_module_state(module: torch.nn.Module, src: int = 0, bucket_bytes: int = 256 << 20):
    """Broadcast every parameter and buffer of `module` from `src`, packed per dtype into
    buckets of up to `bucket_bytes`.  Returns the number of bytes broadcast."""
    if not (dist.is_available() and dist.is_initialized()):
        return 0          # (a process group of ONE rank still runs the collective: tests/test_dist_gpu.py)
    tensors = [t for t in list(module.parameters()) + list(module.buffers()) if t.numel() > 0]
    by_dtype = {}
    for t in tensors:
        by_dtype.setdefault((t.dtype, t.device), []).append(t)
    total = 0
    for (dtype, device), group in by_dtype.items():
        bucket, size = [], 0
        for t in group + [None]:
            if t is not None and (not bucket or size + t.numel() * t.element_size() <= bucket_bytes):
                bucket.append(t)
                size += t.numel() * t.element_size()
                continue
            flat = torch.cat([b.reshape(-1) for b in bucket])
            dist.broadcast(flat, src=src)
            off = 0
            for b in bucket:
                b.copy_(flat[off:off + b.numel()].view_as(b))
                off += b.numel()
            total += size
            bucket, size = ([t], t.numel() * t.element_size()) if t is not None else ([], 0)
    # the buffers were written in place: state derived from them (conv border tables, BOS rows of
    # persistent K/V buffers) is re-derived at its existing addresses
    for m in module.modules():
        if hasattr(m, "refresh_derived_"):
            m.refresh_derived_()
            if m is module:
                break          # a root-level refresh (SDXLUNet) already walks its sub-modules
    return total


def max_over_ranks(value: float, device) -> float:
    if not (dist.is_available() and dist.is_initialized()):
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier():
    if dist.is_available() and dist.is_initialized():
        dist.barrier()
