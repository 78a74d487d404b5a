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


LONG_TEXT = " ".join(f"Sentence number {i} of a long request, which the handler cuts into chunks." for i in range(12))


def _fake_chunks(model, fish_ae, pca_state, sample_fn, pieces, seed, speaker_latent, speaker_mask, sequence_length, max_batch,
                 speaker_kv=None, indices=None):
    """Stands in for the GPU sampler + decode: chunk i of a request -> a waveform that encodes (seed, i, its text length)."""
    idx = list(range(len(pieces))) if indices is None else list(indices)
    return [torch.full((1, 100 + 7 * i), float(seed + 1000 * i + len(p_) % 13)) for i, p_ in zip(idx, pieces)]


def _handler_job(rank: int):
    """Runs the same request through handler._run_job on this rank; returns True when this rank got the full, correct audio."""
    import types
    from echo_tts_amd import handler as H
    H._sample_chunks_batched = _fake_chunks
    model = types.SimpleNamespace(device=torch.device("cpu"), dtype=torch.float32, config=types.SimpleNamespace(latent_size=80))
    job = {"text": LONG_TEXT, "parameters": {"seed": 5, "max_chars_per_chunk": 120, "normalize_boundaries": False, "enable_crossfade": False}}
    audio, seed, n = H._run_job(job, model, None, None, None, None)
    pieces = H.chunk_text_for_audio(LONG_TEXT, max_chars=120, target_duration_seconds=10.0)
    want = torch.cat(_fake_chunks(None, None, None, None, pieces, 5, None, None, 640, 8), dim=-1)
    if audio is None:
        return False
    return n == len(pieces) and n > 3 and seed == 5 and torch.equal(audio, want)


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
    # the batched runner: each rank's share through ONE call, ragged 2-D results, uneven unit count (5 units on 2 ranks)
    calls = []

    def many(units):
        calls.append(list(units))
        return {u: torch.arange(3 * (u + 2), dtype=torch.float32).reshape(3, u + 2) + u for u in units}
    res = P.run_data_parallel_batched(5, many)
    ok = ok and calls == [P.shard_units(5, r, w)]
    if r == 0:
        ok = ok and all(torch.equal(res[u], torch.arange(3 * (u + 2), dtype=torch.float32).reshape(3, u + 2) + u) for u in range(5))
    else:
        ok = ok and res is None
    # the handler's data-parallel path: every rank gets the same request, rank 0 the assembled audio in chunk order
    ok = ok and _handler_job(r) == (r == 0)
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
