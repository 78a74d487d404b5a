"""Data-parallel scale-out of independent text chunks over the GPUs of one node.

One process per GPU (torchrun), `torch.distributed` over RCCL ("nccl" backend on ROCm).  The
reference is single-process / single-device; its unit of independent work is one text chunk with
its own seed (handler.py:747-759, inference.py:371-385).  Units cost the same (fixed S, fixed step
count), so they are dealt round-robin.  The only collective is ONE broadcast of the frozen weights
from rank 0 at start-up; there is no collective in the sampler's step loop; finished waveforms are
gathered in unit order on rank 0.
"""
from __future__ import annotations

import os
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def init_distributed(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """(rank, world_size, local_rank) from the torchrun environment; initialises the process group when world > 1."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(int(os.environ.get("ECHO_FORCE_DEVICE", local)))
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_units(n_units: int, rank: int, world: int) -> List[int]:
    """Static round-robin: unit i -> rank i % world."""
    return list(range(rank, n_units, world))


def broadcast_state(spec: Sequence[Tuple[str, Tuple[int, ...]]], state: Optional[Dict[str, torch.Tensor]], device,
                    dtype: torch.dtype, src: int = 0, bucket_bytes: int = 1 << 30) -> Dict[str, torch.Tensor]:
    """Broadcast a checkpoint from `src` to every rank in large flat buckets (xGMI links are point-to-point, so a few
    big transfers beat thousands of small ones).  `spec` lists (name, shape) in a fixed order; ranks other than `src`
    pass state=None."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        assert state is not None
        return state
    rank = dist.get_rank()
    out: Dict[str, torch.Tensor] = {}
    esize = torch.empty((), dtype=dtype).element_size()
    bucket: List[Tuple[str, Tuple[int, ...], int]] = []
    pending = 0

    def flush() -> None:
        nonlocal bucket, pending
        if not bucket:
            return
        flat = torch.empty((pending,), dtype=dtype, device=device)
        if rank == src:
            off = 0
            for name, shape, n in bucket:
                flat[off:off + n] = state[name].to(device=device, dtype=dtype).reshape(-1)
                off += n
        dist.broadcast(flat.view(torch.uint8), src=src)   # raw bytes: independent of the backend's dtype support
        off = 0
        for name, shape, n in bucket:
            out[name] = flat[off:off + n].view(shape)
            off += n
        bucket, pending = [], 0

    for name, shape in spec:
        n = 1
        for s in shape:
            n *= s
        if pending and (pending + n) * esize > bucket_bytes:
            flush()
        bucket.append((name, tuple(shape), n))
        pending += n
    flush()
    return out


def gather_ordered(local: Dict[int, torch.Tensor], n_units: int, dst: int = 0) -> Optional[List[torch.Tensor]]:
    """Collect per-unit results (possibly different lengths) on `dst`, ordered by unit index."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return [local[i] for i in range(n_units)]
    payload = {i: t.detach().to("cpu") for i, t in local.items()}
    gathered: List[Optional[dict]] = [None] * dist.get_world_size() if dist.get_rank() == dst else None
    dist.gather_object(payload, gathered, dst=dst)
    if dist.get_rank() != dst:
        return None
    merged: Dict[int, torch.Tensor] = {}
    for part in gathered:
        merged.update(part)
    return [merged[i] for i in range(n_units)]


def run_data_parallel(n_units: int, work: Callable[[int], torch.Tensor]) -> Optional[List[torch.Tensor]]:
    """Run `work(unit)` for this rank's units and gather everything on rank 0 in order."""
    rank = dist.get_rank() if dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    mine = {i: work(i) for i in shard_units(n_units, rank, world)}
    return gather_ordered(mine, n_units)
