"""CPU, gloo, world_size 2: the data-parallel plumbing (sharding, weight broadcast, ordered gather)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from echo_tts_amd import parallel as P


def _free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank: int, world: int, port: int, q) -> None:
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r, w, _ = P.init_distributed(backend="gloo")
    spec = [("a.weight", (3, 5)), ("b.bias", (7,)), ("c.weight", (2, 2, 2))]
    state = None
    if r == 0:
        g = torch.Generator().manual_seed(0)
        state = {n: torch.randn(s, generator=g) for n, s in spec}
    got = P.broadcast_state(spec, state, "cpu", torch.float32, bucket_bytes=64)     # tiny buckets: several flushes
    g = torch.Generator().manual_seed(0)
    want = {n: torch.randn(s, generator=g) for n, s in spec}
    ok = all(torch.equal(got[n], want[n]) for n, _ in spec)
    # bf16 checkpoints travel as raw bytes
    st16 = {n: want[n].bfloat16() for n, _ in spec} if r == 0 else None
    got16 = P.broadcast_state(spec, st16, "cpu", torch.bfloat16, bucket_bytes=1 << 20)
    ok = ok and all(torch.equal(got16[n], want[n].bfloat16()) for n, _ in spec)
    res = P.run_data_parallel(5, lambda i: torch.full((i + 1,), float(i)))
    if r == 0:
        ok = ok and [t.tolist() for t in res] == [[float(i)] * (i + 1) for i in range(5)]
    else:
        ok = ok and res is None
    q.put((r, ok, P.shard_units(5, r, w)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_broadcast_shard_gather():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert out[0] == (0, True, [0, 2, 4]) and out[1] == (1, True, [1, 3])


def test_single_process_paths():
    assert P.shard_units(7, 1, 3) == [1, 4]
    st = {"w": torch.ones(2)}
    assert P.broadcast_state([("w", (2,))], st, "cpu", torch.float32) is st
    assert [t.item() for t in P.run_data_parallel(3, lambda i: torch.tensor(float(i)))] == [0.0, 1.0, 2.0]
