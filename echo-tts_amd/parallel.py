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
        if "MASTER_PORT" not in os.environ:      # a launcher (torchrun, bench.py's own spawner) always sets it; there is no safe default
            raise RuntimeError("WORLD_SIZE > 1 but MASTER_PORT is not set: launch the ranks with torch.distributed.run "
                               "(or `python bench.py --gpus N`, which picks a free port)")
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


def _free_port() -> int:
    import socket
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    return port


def gather_ordered(local: Dict[int, torch.Tensor], n_units: int, dst: int = 0) -> Optional[List[torch.Tensor]]:
    """Collect per-unit fp32 results (possibly different lengths / shapes, <= 4 dims) on `dst`, ordered by unit index.
    Plain tensor collectives only (two all_gathers: a small shape table, then the zero-padded payloads) - they run the same over
    RCCL (device tensors, no host staging, no pickling) and over gloo (CPU rehearsals).  Called once per request / job after the
    step loop; nothing here is on the sampler's path."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return [local[i] for i in range(n_units)]
    world, rank = dist.get_world_size(), dist.get_rank()
    on_gpu = dist.get_backend() == "nccl"
    dev = torch.device("cuda", torch.cuda.current_device()) if on_gpu else torch.device("cpu")
    per = (n_units + world - 1) // world
    items = sorted(local.items())
    if len(items) > per:
        raise ValueError(f"rank {rank} holds {len(items)} units, more than ceil({n_units} / {world})")
    meta = torch.full((per, 6), -1, dtype=torch.int64)
    for j, (u, t) in enumerate(items):
        if t.dim() > 4:
            raise ValueError("gather_ordered: at most 4 dims per unit")
        meta[j, 0], meta[j, 1] = u, t.dim()
        for d, n in enumerate(t.shape):
            meta[j, 2 + d] = n
    meta = meta.to(dev)
    metas = [torch.empty_like(meta) for _ in range(world)]
    dist.all_gather(metas, meta)
    metas = [m.cpu() for m in metas]

    def numel(row) -> int:
        n = 1
        for d in range(int(row[1])):
            n *= int(row[2 + d])
        return n
    longest = max([numel(r) for m in metas for r in m if int(r[0]) >= 0] + [1])
    buf = torch.zeros((per, longest), dtype=torch.float32, device=dev)
    for j, (u, t) in enumerate(items):
        buf[j, : t.numel()] = t.detach().to(device=dev, dtype=torch.float32).reshape(-1)
    bufs = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(bufs, buf)
    if rank != dst:
        return None
    merged: Dict[int, torch.Tensor] = {}
    for r in range(world):
        for j in range(per):
            row = metas[r][j]
            if int(row[0]) >= 0:
                shape = [int(row[2 + d]) for d in range(int(row[1]))]
                merged[int(row[0])] = bufs[r][j, : numel(row)].reshape(shape).cpu()
    missing = [i for i in range(n_units) if i not in merged]
    if missing:
        raise RuntimeError(f"gather_ordered: units {missing} were produced by no rank")
    return [merged[i] for i in range(n_units)]


def run_data_parallel(n_units: int, work: Callable[[int], torch.Tensor]) -> Optional[List[torch.Tensor]]:
    """Run `work(unit)` for this rank's units and gather everything on rank 0 in order."""
    rank = dist.get_rank() if dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    mine = {i: work(i) for i in shard_units(n_units, rank, world)}
    return gather_ordered(mine, n_units)


def run_data_parallel_batched(n_units: int, work_many: Callable[[List[int]], Dict[int, torch.Tensor]]) -> Optional[List[torch.Tensor]]:
    """The runner of SURVEY.md 8e: this rank's round-robin share of the units goes through ONE `work_many(units)` call (the
    reference sampler's own batch axis: the chunks of a request / the utterances of a C3 batch as rows of one sampler call), the
    results are gathered in unit order on rank 0 (None elsewhere).  No collective inside `work_many`."""
    rank = dist.get_rank() if dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    mine = shard_units(n_units, rank, world)
    local = work_many(mine) if mine else {}
    if sorted(local) != mine:
        raise RuntimeError(f"work_many returned units {sorted(local)} for {mine}")
    return gather_ordered(local, n_units)
