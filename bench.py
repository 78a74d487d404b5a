#!/usr/bin/env python3
"""Benchmark of the Echo-TTS hot path on MI355X: BASELINE.json config C2.

One "step" = one utterance through the whole hot path on one GPU: text + speaker KV encode,
sample_euler_cfg_independent_guidances (S=640, 40 Euler steps, CFG text 3.0 / speaker 8.0 on
t in [0.5, 1] -> 20 three-row + 20 one-row EchoDiT forwards, bf16) and ae_decode (Fish S1-DAC, fp32)
-> 640 * 2048 / 44100 = 29.72 audio-seconds.  Inputs are resident in HBM before the timed region.
Weights are seeded random tensors of the exact architecture (the checkpoints are gated), text is a
synthetic 436-token prompt padded to 768 like sample_pipeline does, the speaker reference is a
synthetic (1, 2560, 80) latent.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python bench.py --gpus N ...          (no torchrun environment: this process spawns the N ranks itself, before it touches the GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A step runs `--batch` (default 24; BASELINE config C3's per-GPU share is 4, `--batch 4`) independent
utterances through ONE sampler call, using the reference sampler's own batch axis (inference.py:448-449: text ids
(B, Tt), speaker latents (B, Ts, 80), one noise draw of (B, S, 80)): the EchoDiT GEMMs then see M = 3*B*640 rows in
the CFG steps and B*640 in the others, which is what fills 256 CUs with 256x256 tiles.  `--batch 1` is the
single-request configuration (C2 proper); its throughput is measured in the same run and reported as
`single_request`.  `--concurrency` (default 2) such calls are kept in flight on separate HIP streams / engine contexts,
as a serving process does with independent requests (handler.py:747-759): the second call's kernels fill the CUs that
the last, partial round of 256x256 tiles of the first leaves idle (+6 % measured; `--concurrency 1` for one call).
Measured sweep (audio-s/s, batch x concurrency, round-1 v7 binaries): 1x1 97, 4x1 134, 4x2 148, 8x2 157, 12x2 160, 16x2 161;
round 2: 8x2 169-173, 12x2 172.6, 16x2 172.9, 8x3 169.6.

Next to the headline the line carries: `single_request` (C2 proper: one utterance per call, with its own roofline fraction), `c3`
(BASELINE config C3: 4 mixed-length text_presets utterances per GPU, seeds = unit index, sharded over the ranks by
`parallel.run_data_parallel_batched` and gathered in order on rank 0; `c3_share` is its N = 1 name), `c5` (BASELINE config C5: fp8
operands, 100 Euler steps, its own roofline against the fp8 peak), `legacy_r1_workload` (round 1's definition of a step: 8 per call,
per-row speaker KV), `cpu_baseline` (the oracle on the host cores, real 24-layer model) and `eager_gpu_baseline` (the oracle's
torch ops run eagerly through PyTorch-ROCm on the same GPU: the reference's own execution model).  The one reference voice is encoded
once per sampler call, inside the timed region, and shared by the call's rows.

Multi-GPU: independent utterances shard data-parallel (weak scaling: every rank runs `steps`
utterances); the only collective in the job is the start-up broadcast of the frozen weights from
rank 0 over RCCL, plus the barriers / max-reduction that bracket the timed region.
Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

S, STEPS, TT, TVALID, TS = 640, 40, 768, 436, 2560
SAMPLER = dict(num_steps=STEPS, cfg_scale_text=3.0, cfg_scale_speaker=8.0, cfg_min_t=0.5, cfg_max_t=1.0, truncation_factor=None,
               rescale_k=None, rescale_sigma=None, speaker_kv_scale=None, speaker_kv_max_layers=None, speaker_kv_min_t=None,
               sequence_length=S)
AUDIO_S = S * 2048 / 44100.0
PEAK_BF16_TFLOPS = 2500.0          # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)
SUSTAINED_BF16_TFLOPS = 1770.0   # measured: pure MFMA issue, random bf16 operands, board power-limited (profiles/r02_mfma_power.log)


PEAK_FP8_TFLOPS = 5000.0           # dense block-scaled fp8 MFMA (same guide): the --c5 line is priced against this


def dit_gemm_flops(batch: int = 1, steps: int = STEPS) -> float:
    """Algorithmic FLOPs of the gemm_nt launches inside one sampler run: 80 row-forwards of
    S * 2 743 730 176 plus the one-time modulation tables (cond MLP + 144 low-rank refinements for 40 timesteps)."""
    D, F, Lz, L, R, E = 2048, 5888, 80, 24, 256, 512
    per_row = S * (2 * L * (5 * D * D + 3 * D * F) + 2 * 2 * Lz * D)
    rows = (steps // 2) * 3 + (steps - steps // 2)
    mod = 2 * steps * (E * D + D * D + D * 3 * D) + 2 * steps * (2 * L * 3) * (2 * D * R)
    return float(batch * rows * per_row + mod)


def pp_algorithmic_bytes_per_launch(batch: int, steps: int = STEPS, fp8: bool = False) -> float:
    """Algorithmic HBM bytes of an average gemm_pp_kernel launch of one sampler call (A and W read once, C written once, the residual of
    wo / w2 read once): QKVG (N 8192, K 2048), wo (2048, 2048), w1||w3 (N 11776 -> 5888 columns out, K 2048), w2 (2048, K 5888) at
    M = 3 * batch * 640 rows in the CFG steps and batch * 640 in the others; bf16 (fp8: 1-byte A / W operands)."""
    D, F = 2048, 5888
    ab = 1 if fp8 else 2

    def layer(M):
        qkvg = M * D * ab + 4 * D * D * ab + M * 4 * D * 2
        wo = M * D * ab + D * D * ab + 2 * M * D * 2
        w13 = M * D * ab + 2 * F * D * ab + M * F * 2
        w2 = M * F * ab + D * F * ab + 2 * M * D * 2
        return qkvg + wo + w13 + w2
    n3, n1 = steps // 2, steps - steps // 2
    return (n3 * layer(3 * batch * S) + n1 * layer(batch * S)) / (4.0 * (n3 + n1))


def build(device, rank: int, world: int, concurrency: int = 1, batch: int = 1, fp8: bool = False, keep_state: bool = False, bcast=None):
    import echo_tts_amd as E
    from echo_tts_amd import parallel as P
    from echo_tts_amd.weights import dac_param_shapes, dit_param_shapes, random_dac_state, random_dit_state
    cfg, dcfg = E.EchoDiTConfig(), E.DACConfig()
    # frozen weights: rank 0 draws them, everyone else receives them over RCCL (one broadcast, bucketed)
    sd = random_dit_state(cfg, device, torch.bfloat16, seed=0) if rank == 0 else None
    if world > 1:
        torch.cuda.synchronize()
        dist.barrier()
        tb = time.perf_counter()
    sd = P.broadcast_state(dit_param_shapes(cfg, with_blockwise=False), sd, device, torch.bfloat16)
    if world > 1 and bcast is not None:
        torch.cuda.synchronize()
        dist.barrier()
        tb = time.perf_counter() - tb
        nbytes = sum(v.numel() * v.element_size() for v in sd.values())
        bcast.update({"what": "EchoDiT checkpoint, bf16, 1 GiB buckets, rank 0 -> all (the job's only data collective)", "bytes": nbytes,
                      "seconds": round(tb, 3), "GB_per_s": round(nbytes / tb / 1e9, 2)})
    # one engine context (packed weights + KV caches + workspaces) per concurrent request slot
    models = [E.EchoDiT(cfg, sd, dtype=torch.bfloat16, device=device, fp8=fp8) for _ in range(concurrency)]
    dsd = random_dac_state(dcfg, device, seed=0) if rank == 0 else None
    dsd = P.broadcast_state(dac_param_shapes(dcfg), dsd, device, torch.float32)
    dacs = [E.DAC(dcfg, dsd, device=device) for _ in range(concurrency)]
    state = {"dit": sd, "dac": dsd} if keep_state else None      # reference-named checkpoints for the baseline legs (rank 0)
    del sd, dsd
    torch.cuda.empty_cache()
    g = torch.Generator().manual_seed(1234)
    q, _ = torch.linalg.qr(torch.randn(dcfg.latent_dim, cfg.latent_size, generator=g))
    pca = E.PCAState(q.T.contiguous().to(device), (0.1 * torch.randn(dcfg.latent_dim, generator=g)).to(device), 1.0)
    ids = torch.zeros((1, TT), dtype=torch.int32)
    ids[0, 1:TVALID] = torch.randint(32, 127, (TVALID - 1,), generator=g, dtype=torch.int32)
    tmask = torch.zeros((1, TT), dtype=torch.bool)
    tmask[0, :TVALID] = True
    spk = torch.randn((1, TS, cfg.latent_size), generator=g).to(device)
    smask = torch.ones((1, TS), dtype=torch.bool)
    if batch > 1:
        # the reference's own batch axis (inference.py:448-449): `batch` utterances through one sampler call, each with its own
        # noise and its own text KV; the ONE reference voice (C3: "same speaker"; the chunks of a handler request,
        # handler.py:747-759) is encoded once per call - inside the timed region - and shared by all rows (stride-0 KV)
        ids, tmask = ids.repeat(batch, 1), tmask.repeat(batch, 1)
    return E, models, dacs, pca, ids.to(device), tmask, spk, smask, state


def _baseline_ops():
    """The checker's ops (oracle/): imported here only, for the two reported BASELINE legs below - never on the measured path."""
    from oracle import echo_ref as R
    return R


def cpu_baseline(threads: int, state, pca, ids, tmask, spk, smask):
    """The CPU oracle (plain PyTorch restatement of the reference: bf16 EchoDiT + fp32 DAC like the GPU run) on the host cores,
    on the REAL 24-layer model with the bench's own weights and inputs: text + speaker KV encode in full, one 3-row
    CFG forward and one 1-row forward at S = 640 (the two step kinds of the schedule, each measured once after a warm-up of the
    1-row kind), DAC decode of 64 of the 640 frames.  Whole-utterance time = encoders + 20 x t3 + 20 x t1 + decode x 10."""
    R = _baseline_ops()
    torch.set_num_threads(threads)
    cfg, dcfg = R.DiTConfig(), R.DacConfig()
    w = {k: v.to("cpu") for k, v in state["dit"].items()}
    dw = {k: v.to("cpu") for k, v in state["dac"].items()}
    ids1, tm1, spk1, sm1 = ids[:1].cpu(), tmask[:1].cpu(), spk[:1].cpu().bfloat16(), smask[:1].cpu()
    g = torch.Generator().manual_seed(0)
    x = torch.randn((1, S, 80), generator=g)
    pc = R.PCA(pca.pca_components.cpu(), pca.pca_mean.cpu(), float(pca.latent_scale))
    with torch.inference_mode():
        t0 = time.perf_counter()
        kvt = R.kv_cache_text(w, cfg, ids1, tm1)
        kvs = R.kv_cache_speaker(w, cfg, spk1)
        tenc = time.perf_counter() - t0
        kvt3, kvs3 = R._cat3(kvt), R._cat3(kvs)
        tm3 = torch.cat([tm1, torch.zeros_like(tm1), tm1], 0)
        sm3 = torch.cat([sm1, sm1, torch.zeros_like(sm1)], 0)

        def fwd(rows):
            xx = torch.cat([x] * rows, 0).bfloat16()
            tt = (torch.ones((rows,)) * 0.7).bfloat16()
            t0 = time.perf_counter()
            if rows == 3:
                R.dit_forward(w, cfg, xx, tt, tm3, sm3, kvt3, kvs3)
            else:
                R.dit_forward(w, cfg, xx, tt, tm1, sm1, kvt, kvs)
            return time.perf_counter() - t0
        fwd(1)
        t3, t1 = fwd(3), fwd(1)
        nf = 64
        lat = torch.randn((1, nf, 80), generator=g)
        t0 = time.perf_counter()
        R.ae_decode(dw, dcfg, pc, lat)
        tdac = time.perf_counter() - t0
    est = tenc + 20 * t3 + 20 * t1 + tdac * (S / nf)
    return {"value": AUDIO_S / est, "unit": "audio-s/s", "cores": threads, "cpu_model": cpu_model(), "kind": "port",
            "sample": f"oracle/echo_ref.py, torch {torch.__version__} CPU eager, bf16 EchoDiT + fp32 DAC, the bench's 24-layer weights and inputs: "
                      f"text + speaker KV encode in full ({tenc:.2f}s), one 3-row CFG forward ({t3:.2f}s) and one 1-row forward ({t1:.2f}s) at S=640 "
                      f"x 20 steps each, DAC decode of {nf} of 640 frames ({tdac:.2f}s) x 10; estimated {est:.0f}s per utterance"}


def eager_gpu_baseline(state, pca, ids, tmask, spk, smask, sampler_kw, device):
    """SURVEY.md 8d / BASELINE.md 4: the reference's own execution model on this GPU - the oracle's torch ops run eagerly
    through PyTorch-ROCm (bf16 EchoDiT sampler, fp32 DAC decode), one utterance end to end (C2 proper), same weights and inputs.
    The oracle is only ever the baseline here, never the product path."""
    R = _baseline_ops()
    cfg, dcfg = R.DiTConfig(), R.DacConfig()
    w, dw = state["dit"], state["dac"]
    pc = R.PCA(pca.pca_components.to(device), pca.pca_mean.to(device), float(pca.latent_scale))
    a = (spk[:1].to(device), smask[:1].to(device), ids[:1].to(device), tmask[:1].to(device))
    times = {}
    with torch.inference_mode():
        for it in range(2):                    # first pass warms code objects / MIOpen solvers
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            lat = R.sample_euler(w, cfg, torch.bfloat16, *a, rng_seed=it, **sampler_kw)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            wav = R.ae_decode(dw, dcfg, pc, lat)
            torch.cuda.synchronize()
            times = {"sampler_ms": 1e3 * (t1 - t0), "dac_ms": 1e3 * (time.perf_counter() - t1)}
    tot = (times["sampler_ms"] + times["dac_ms"]) * 1e-3
    return {"value": round(AUDIO_S / tot, 3), "unit": "audio-s/s", "kind": "port (oracle ops, PyTorch-ROCm eager on the same MI355X)",
            "ms_per_utterance": round(1e3 * tot, 1), "sampler_ms": round(times["sampler_ms"], 1), "dac_decode_ms": round(times["dac_ms"], 1),
            "finite": bool(torch.isfinite(wav).all()),
            "workload": f"one utterance per call, {sampler_kw['num_steps']} steps, bf16 EchoDiT + fp32 DAC, second of two runs"}


def cpu_model() -> str:
    """Model name of the host CPU the baseline ran on (/proc/cpuinfo)."""
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def host_threads() -> int:
    """CPU threads this process may really use: affinity mask, cgroup quota, capped at the box's 16-core share."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, 16))


# token counts of the 20 prompts of the reference's text_presets.txt through its tokenizer (tests/golden/meta.json,
# host.preset_token_lengths, written by tests/make_goldens.py); the texts themselves stay in the reference - the sampler's cost
# depends on the lengths only
PRESET_TOKEN_LENGTHS = [446, 372, 297, 248, 384, 484, 396, 622, 527, 408, 151, 403, 405, 338, 490, 581, 369, 378, 654, 667]


def mixed_units(units, device):
    """BASELINE config C3's inputs for the given unit indices: unit u is preset u % 20 (synthetic token ids of that prompt's length,
    padded to 768 like sample_pipeline) with seed u (handler.py:749: every chunk / utterance its own seed)."""
    n = len(units)
    ids = torch.zeros((n, TT), dtype=torch.int32)
    tmask = torch.zeros((n, TT), dtype=torch.bool)
    for i, u in enumerate(units):
        ln = PRESET_TOKEN_LENGTHS[u % len(PRESET_TOKEN_LENGTHS)]
        g = torch.Generator().manual_seed(9000 + u)
        ids[i, 1:ln] = torch.randint(32, 127, (ln - 1,), generator=g, dtype=torch.int32)
        tmask[i, :ln] = True
    x0 = torch.cat([torch.randn((1, S, 80), device=device, dtype=torch.float32, generator=torch.Generator(device=device).manual_seed(u))
                    for u in units], 0)
    return ids.to(device), tmask, x0


def free_port() -> int:
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def spawn_ranks(n: int) -> int:
    """`python bench.py --gpus N` outside a torchrun environment: start the N ranks as a child job (torch.distributed.run, one process
    per GPU, rendezvous on 127.0.0.1 and a free port) BEFORE this process makes any GPU call, relay rank 0's JSON line and the
    children's stderr, and return the job's exit code.  Nothing here initialises HIP: a process that has touched the GPU must never
    be replaced or forked."""
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env)
    line = None
    for ln in p.stdout:
        t = ln.strip()
        if t.startswith("{") and '"metric"' in t:
            line = t
        elif t:
            print(t, file=sys.stderr)
    rc = p.wait()
    if line:
        print(line)
    return rc if rc else (0 if line else 1)


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-eager-baseline", action="store_true")
    ap.add_argument("--no-c5", action="store_true", help="skip the C5 (fp8, 100 steps) sub-record of the default run")
    ap.add_argument("--no-legs", action="store_true", help="skip single_request / c3 / legacy legs (headline only)")
    ap.add_argument("--concurrency", type=int, default=2,
                    help="independent utterances in flight per GPU (one HIP stream + one engine context each); a step is then "
                         "`concurrency` utterances")
    ap.add_argument("--batch", type=int, default=24,
                    help="utterances per sampler call (the reference's batch axis B): M = 3*B*640 / B*640 GEMM rows; at most 32 (the engine holds 96 rows per call)")
    ap.add_argument("--c5", action="store_true",
                    help="BASELINE config C5 as the HEADLINE instead of C2: fp8 (e4m3) operands for the EchoDiT block GEMMs, 100 Euler steps "
                         "(50 CFG x3 rows + 50 x1 row); the JSON line then says dtype fp8 and names C5 in config.workload")
    ap.add_argument("--c5-dynamic", action="store_true",
                    help="with --c5: per-token-row activation scales for every operand (two quantisation passes per block) instead of the "
                         "calibrated static scales for the attention / SwiGLU outputs")
    ap.add_argument("--dist-backend", default=None, help="testing only: e.g. gloo to rehearse N ranks on one GPU")
    ap.add_argument("--force-device", type=int, default=None, help="testing only: every rank uses this cuda index")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(args.gpus))             # no GPU call has happened in this process

    from echo_tts_amd import parallel as P
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    if args.force_device is not None:
        os.environ["ECHO_FORCE_DEVICE"] = str(args.force_device)
    rank, world, local = P.init_distributed(backend=args.dist_backend)
    if world != max(1, args.gpus):
        raise SystemExit(f"--gpus {args.gpus} but the job has WORLD_SIZE {world}: launch N ranks (python bench.py --gpus N does it itself)")
    if args.force_device is not None:
        local = args.force_device
    device = torch.device(f"cuda:{local}")
    torch.cuda.set_device(device)
    conc = max(1, args.concurrency)
    nb = max(1, args.batch)
    n_steps = 100 if args.c5 else STEPS
    sampler_kw = dict(SAMPLER, num_steps=n_steps)
    want_base = rank == 0 and world == 1 and not (args.no_cpu_baseline and args.no_eager_baseline)
    want_c5_leg = rank == 0 and world == 1 and not args.c5 and not args.no_c5 and not args.no_roofline
    bcast = {}
    E, models, dacs, pca, ids, tmask, spk, smask, state = build(device, rank, world, conc, nb, fp8=args.c5, keep_state=want_base or want_c5_leg, bcast=bcast)
    model, dac = models[0], dacs[0]

    def calibrate_fp8(ms, kw):
        """fp8 activation-scale calibration (outside any timed region; a deployment does it once per checkpoint and keeps the JSON of
        weights.save_fp8_scales): 10 Euler steps of the bench's own request on the dynamic path, maxima x 1.25"""
        ms[0].fp8_calibration_start()
        E.sample_euler_cfg_independent_guidances(ms[0], spk, smask, ids[:min(nb, 4)], tmask[:min(nb, 4)], rng_seed=11, **dict(kw, num_steps=10))
        sc = ms[0].fp8_calibration_finish(margin=1.25)
        for m in ms:
            m.set_fp8_static_scales(sc)
        return sc

    c5_static = calibrate_fp8(models, sampler_kw) if (args.c5 and not args.c5_dynamic) else None
    streams = [torch.cuda.Stream(device=device) for _ in range(conc)] if conc > 1 else [torch.cuda.current_stream(device)]

    last = {}

    def make_step(ms, ds, kw):
        def step(seed: int):
            """One step: `conc` independent sampler calls of `nb` utterances + their decodes, each on its own stream and engine context."""
            wavs = []
            for c in range(len(ms)):
                with torch.cuda.stream(streams[c]):
                    lat = E.sample_euler_cfg_independent_guidances(ms[c], spk, smask, ids, tmask, rng_seed=seed * conc + c, **kw)
                    last["latent"] = lat
                    wavs.append(E.ae_decode(ds[c], pca, lat))
            return wavs[-1]
        return step

    utterance = make_step(models, dacs, sampler_kw)
    for i in range(args.warmup):
        utterance(1000 + i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        wav = utterance(rank * 100000 + i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    if not bool(torch.isfinite(wav).all()):
        raise SystemExit(f"rank {rank}: non-finite waveform (latent finite: {bool(torch.isfinite(last['latent']).all())}, "
                         f"latent rms {float(last['latent'].float().pow(2).mean().sqrt()):.3g})")

    def roofline_of(m, d, kw, fp8: bool):
        """Live HIP-event timing of every GEMM launch of ONE sampler call of `nb` utterances (events on the launch stream)."""
        steps_n = kw["num_steps"]
        m.set_profiling(True)
        lat = E.sample_euler_cfg_independent_guidances(m, spk, smask, ids, tmask, rng_seed=7, **kw)
        pr = m.get_profile()
        m.set_profiling(False)
        flops = dit_gemm_flops(nb, steps_n)
        peak = PEAK_FP8_TFLOPS if fp8 else PEAK_BF16_TFLOPS
        ach_all = flops / (pr.ms_gemm_sum * 1e-3) / 1e12
        # dominant kernel: the gemm_pp_kernel launches of that call (engine-side HIP events on the launch stream)
        ach = pr.flops_pp / (pr.ms_pp_sum * 1e-3) / 1e12 if pr.n_pp else 0.0
        traffic, traffic_src = None, None
        try:   # HBM-side bytes per launch from the committed rocprofv3 --pmc passes (cannot be collected inside this process)
            if fp8:
                raise RuntimeError("the PMC passes were taken on the C2 command")
            pmf = next(f for f in ("r03_pmc_gemm.json", "r02_pmc_gemm.json", "r01_pmc_gemm.json") if os.path.exists(os.path.join(ROOT, "profiles", f)))
            pm = json.load(open(os.path.join(ROOT, "profiles", pmf)))
            traffic, traffic_src = round(pm["traffic_bytes_per_launch"]), f"profiles/{pmf} (2 x FETCH_SIZE + WRITE_SIZE, KiB; rocprofv3 --pmc passes of this command at the default batch)"
        except Exception:
            pass
        pmc_mfma = None
        try:   # MFMA-pipe utilisation and effective clock of the same kernel from the committed rocprofv3 --pmc pass (tools/summarize_pmc_mfma.py)
            pmf = next(f for f in (("r03_pmc_mfma_c5.json", "r02_pmc_mfma_c5.json") if fp8 else ("r03_pmc_mfma.json", "r02_pmc_mfma.json")) if os.path.exists(os.path.join(ROOT, "profiles", f)))
            k = json.load(open(os.path.join(ROOT, "profiles", pmf)))["kernels"]["gemm_pp_kernel"]
            pmc_mfma = {"mfma_utilisation": round(k["mfma_utilisation"], 4), "effective_clock_ghz": round(k["effective_clock_ghz"], 3),
                        "source": f"profiles/{pmf} (SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8); utilisation x clock / 2.4 GHz ~ frac)"}
        except Exception:
            pass
        rf = {"bound": "mfma", "kernel": "gemm_pp_kernel (" + ("fp8-e4m3" if fp8 else "bf16") + " 256x256 ping-pong GEMM: QKVG / wo / SwiGLU / w2 of every EchoDiT block)",
              "achieved": round(ach, 1), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4),
              "traffic": traffic, "traffic_source": traffic_src,
              "algorithmic_bytes_per_launch": round(pp_algorithmic_bytes_per_launch(nb, steps_n, fp8)),
              "traffic_over_algorithmic": round(traffic / pp_algorithmic_bytes_per_launch(nb, steps_n, fp8), 2) if traffic else None,
              "launches": pr.n_pp, "avg_launch_us": round(1e3 * pr.ms_pp_sum / max(pr.n_pp, 1), 2),
              "flops_per_launch": pr.flops_pp / max(pr.n_pp, 1),
              "pmc_mfma": pmc_mfma,
              "power_envelope": None if fp8 else {
                  "sustained_mfma_peak": SUSTAINED_BF16_TFLOPS, "frac_of_sustained": round(ach / SUSTAINED_BF16_TFLOPS, 4),
                  "note": "v_mfma_f32_32x32x16_bf16 issued back to back from registers on every SIMD (no LDS, no memory) sustains 1770 TFLOP/s on "
                          "random operands (board at its power limit, shader clock 1.7-1.8 GHz; 2484 on zeros): tools/micro/mfma_power.hip, "
                          "profiles/r02_mfma_power.log; `peak`/`frac` above stay the dense 2.4 GHz figure of MI355X_MICROARCH.md"},
              "all_linears": {"achieved": round(ach_all, 1), "launches": pr.n_gemm, "ms": round(pr.ms_gemm_sum, 2),
                              "note": "every gemm launch of the call incl. the small ones (in/out projections, modulation tables), algorithmic FLOPs of SURVEY.md 8d"}}
        d.set_profiling(True)
        E.ae_decode(d, pca, lat)
        dp = d.get_profile()
        d.set_profiling(False)
        # joint attention inside the sampler call: algorithmic FLOPs of SURVEY.md 8d (14.57 TFLOP per utterance at C2)
        attn_flops = nb * 640 * 196608.0 * 115760.0 * (steps_n / STEPS)
        attn = {"ms": round(pr.ms_attn_sum, 2), "launches": pr.n_attn,
                "achieved_tflops": round(attn_flops / (pr.ms_attn_sum * 1e-3) / 1e12, 1) if pr.n_attn else None}
        ph = {"utterances_per_call": nb, "sampler_ms": round(pr.ms_total, 2), "attention": attn, "sampler_gemm_ms": round(pr.ms_gemm_sum, 2), "mod_tables_ms": round(pr.ms_mod, 2),
              "dac_decode_ms_per_utterance": round(dp.ms_total, 2), "dac_gemm_ms_per_utterance": round(dp.ms_gemm_sum, 2)}
        return rf, ph

    roofline = phases = None
    if rank == 0 and not args.no_roofline:
        roofline, phases = roofline_of(model, dac, sampler_kw, args.c5)
    legs = rank == 0 and not args.no_roofline and not args.no_legs
    single = None
    if legs and nb * conc > 1:
        # the same engine on one utterance at a time (BASELINE config C2 proper), 1 warm-up + 3 timed; profiled once more for its phases
        i1, t1, s1, m1 = ids[:1], tmask[:1], spk, smask
        ms = []
        for i in range(4):
            torch.cuda.synchronize()
            t0s = time.perf_counter()
            lat1 = E.sample_euler_cfg_independent_guidances(model, s1, m1, i1, t1, rng_seed=50 + i, **sampler_kw)
            E.ae_decode(dac, pca, lat1)
            torch.cuda.synchronize()
            ms.append(1e3 * (time.perf_counter() - t0s))
        best = sorted(ms[1:])[1]
        model.set_profiling(True)
        E.sample_euler_cfg_independent_guidances(model, s1, m1, i1, t1, rng_seed=54, **sampler_kw)
        p1 = model.get_profile()
        model.set_profiling(False)
        flops1 = dit_gemm_flops(1, n_steps) + 14.57e12 * (n_steps / STEPS) + 4.33e12      # linears + joint attention + DAC decode
        peak1 = PEAK_FP8_TFLOPS if args.c5 else PEAK_BF16_TFLOPS
        single = {"value": round(AUDIO_S / (best * 1e-3), 3), "unit": "audio-s/s", "ms_per_utterance": round(best, 2),
                  "workload": f"{'C5' if args.c5 else 'C2'}: one utterance per sampler call (M = 1920 / 640 GEMM rows), {n_steps} steps",
                  "roofline": {"bound": "mfma", "achieved": round(flops1 / (best * 1e-3) / 1e12, 1), "peak": peak1, "unit": "TFLOP/s",
                               "frac": round(flops1 / (best * 1e-3) / 1e12 / peak1, 4),
                               "note": "end to end: algorithmic FLOPs of one utterance (EchoDiT linears + joint attention + DAC decode) / wall time of encode + sampler + decode"},
                  "phases_profiled_call": {"sampler_ms": round(p1.ms_total, 2), "gemm_ms": round(p1.ms_gemm_sum, 2), "gemm_launches": p1.n_gemm,
                                           "attention_ms": round(p1.ms_attn_sum, 2), "attention_launches": p1.n_attn, "mod_tables_ms": round(p1.ms_mod, 2),
                                           "note": "HIP events around every GEMM / attention launch serialise the stream a little: the sum of the parts exceeds the un-profiled sampler time"}}

    # BASELINE config C3: 32 mixed text_presets utterances over 8 GPUs = 4 per GPU.  Every rank takes its round-robin share of the
    # 4 * world units (parallel.shard_units), runs it as ONE sampler call on the reference's batch axis (one shared voice), decodes,
    # and rank 0 gathers the waveforms in unit order.  No collective between the barriers.
    c3 = None
    if not args.no_roofline and not args.no_legs and not args.c5:
        n_units = 4 * world

        def work(units):
            ids_u, tmask_u, x0 = mixed_units(units, device)
            lat = E.sample_euler_cfg_independent_guidances(model, spk, smask, ids_u, tmask_u, rng_seed=0, x_init=x0, **sampler_kw)
            wv = E.ae_decode(dac, pca, lat)
            return {u: wv[i] for i, u in enumerate(units)}

        mine = P.shard_units(n_units, rank, world)
        work(mine)                                   # warm (plans / workspaces of this shape)
        times = []
        res = None
        for rep in range(2):
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            t0s = time.perf_counter()
            local_out = work(mine)
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            tl = time.perf_counter() - t0s
            if world > 1:
                tm = torch.tensor([tl], device=device, dtype=torch.float64)
                dist.all_reduce(tm, op=dist.ReduceOp.MAX)
                tl = float(tm.item())
            times.append(tl)
            if rep == 1:
                t0g = time.perf_counter()
                res = P.gather_ordered(local_out, n_units)
                tg = time.perf_counter() - t0g
        if rank == 0:
            best = min(times)
            ok = res is not None and len(res) == n_units and all(bool(torch.isfinite(w_).all()) and w_.shape[-1] == S * 2048 for w_ in res)
            c3 = {"value": round(n_units * AUDIO_S / best, 3), "unit": "audio-s/s (whole job)", "per_gpu": round(4 * AUDIO_S / best, 3), "ms_per_call": round(1e3 * best, 2),
                  "units": n_units, "ranks": world, "gathered_in_order_on_rank0": bool(ok), "gather_ms": round(1e3 * tg, 1),
                  "workload": f"C3: {n_units} utterances = 4 per GPU (32 on 8 GPUs), presets {0}..{n_units - 1} mod 20 of text_presets.txt by token count "
                              f"({min(PRESET_TOKEN_LENGTHS)}..{max(PRESET_TOKEN_LENGTHS)} tokens), seed = unit index, one shared voice, 40 steps, one sampler call per rank "
                              "(M = 7680 / 2560 GEMM rows), sharded round-robin (parallel.shard_units, as run_data_parallel_batched does), ordered gather on rank 0 (parallel.gather_ordered) outside the timed region"}

    legacy = None
    if legs and not args.c5 and nb >= 8:
        # round 1's definition of a step, for a like-for-like series - 8 utterances per call x 2 streams, every row its own
        # speaker KV (the voice replicated per row instead of one shared, stride-0 voice)
        ids8, tm8 = ids[:8], tmask[:8]
        spk8, sm8 = spk.repeat(8, 1, 1), smask.repeat(8, 1)
        ms = []
        for i in range(3):
            torch.cuda.synchronize()
            t0s = time.perf_counter()
            for c in range(conc):
                with torch.cuda.stream(streams[c]):
                    l8 = E.sample_euler_cfg_independent_guidances(models[c], spk8, sm8, ids8, tm8, rng_seed=90 + i * conc + c, **sampler_kw)
                    E.ae_decode(dacs[c], pca, l8)
            torch.cuda.synchronize()
            ms.append(1e3 * (time.perf_counter() - t0s))
        best = min(ms[1:])
        legacy = {"value": round(8 * conc * AUDIO_S / (best * 1e-3), 3), "unit": "audio-s/s", "ms_per_step": round(best, 2),
                  "workload": f"round 1's step: 8 utterances per sampler call x {conc} streams, one speaker latent PER ROW (8 speaker encodes + 8 speaker KV sets per call)"}

    c5 = None
    if want_c5_leg:
        # BASELINE config C5 in the driver's own line: fp8 (e4m3) operands for the EchoDiT block GEMMs with calibrated static
        # scales, 100 Euler steps, the headline's batch and streams: one warm step + two timed, and its own roofline (fp8 peak)
        try:
            kw5 = dict(SAMPLER, num_steps=100)
            m5 = [E.EchoDiT(E.EchoDiTConfig(), state["dit"], dtype=torch.bfloat16, device=device, fp8=True) for _ in range(conc)]
            calibrate_fp8(m5, kw5)
            step5 = make_step(m5, dacs, kw5)
            step5(2000)
            torch.cuda.synchronize()
            t0s = time.perf_counter()
            for i in range(2):
                w5 = step5(2001 + i)
            torch.cuda.synchronize()
            d5 = time.perf_counter() - t0s
            rf5, ph5 = roofline_of(m5[0], dacs[0], kw5, True)
            c5 = {"value": round(2 * conc * nb * AUDIO_S / d5, 3), "unit": "audio-s/s", "ms_per_step": round(1e3 * d5 / 2, 2), "steps": 2, "warmup": 1,
                  "finite": bool(torch.isfinite(w5).all()),
                  "dtype": "fp8 (e4m3 operands of the EchoDiT block GEMMs, fp32 accumulate; calibrated static scales for the attention / SwiGLU outputs "
                           "written by their producers, per-token-row scales elsewhere; bf16 elsewhere, fp32 DAC)",
                  "workload": f"C5: {conc * nb} utterance(s)/step ({nb} per sampler call, {conc} HIP stream(s)), seq_len=640, 100 Euler steps (50 CFG x3 rows + 50 x1 row), fp8 DiT GEMMs + fp32 DAC decode",
                  "roofline": rf5, "phases": ph5}
            del m5
            torch.cuda.empty_cache()
        except Exception as ex:      # a sub-record must never take the bench line down
            import traceback
            c5 = {"error": f"{type(ex).__name__}: {ex}", "where": traceback.format_exc().strip().splitlines()[-3:]}

    cpu = eager = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:   # the CPU leg belongs to the N=1 line only
        cpu = cpu_baseline(host_threads(), state, pca, ids, tmask, spk, smask)
    if rank == 0 and world == 1 and not args.no_eager_baseline:
        try:
            eager = eager_gpu_baseline(state, pca, ids, tmask, spk, smask, sampler_kw, device)
        except Exception as ex:      # a baseline must never take the bench line down
            import traceback
            eager = {"error": f"{type(ex).__name__}: {ex}", "where": traceback.format_exc().strip().splitlines()[-3:]}
    state = None

    if rank == 0:
        total_audio = AUDIO_S * args.steps * world * conc * nb
        out = {
            "metric": "audio-sec/sec/GPU @ seq_len=640, 40 steps, CFG(text=3, spk=8); 1/2/4/8 GPU",
            "value": round(total_audio / dt, 3), "unit": "audio-s/s (whole job; divide by n_gpus for per-GPU)",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 2),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": ("fp8 (e4m3 operands of the EchoDiT block GEMMs, fp32 accumulate; " + ("per-token-row activation scales" if c5_static is None else
                      "calibrated static scales for the attention / SwiGLU outputs written by their producers, per-token-row scales elsewhere")
                      + "; bf16 elsewhere, fp32 DAC)") if args.c5 else "bf16", "data": "synthetic",
            "per_gpu": round(total_audio / dt / world, 3),
            "config": {"workload": f"{'C5' if args.c5 else 'C2'}: {conc * nb} utterance(s)/step/GPU ({nb} per sampler call, {conc} HIP stream(s)), seq_len=640, "
                                   f"{n_steps} Euler steps ({n_steps // 2} CFG x3 rows + {n_steps - n_steps // 2} x1 row), "
                                   "cfg_text=3.0 cfg_spk=8.0, text 436 tokens padded to 768, one speaker latent (1,2560,80) encoded per call and shared by its rows, "
                                   + ("EchoDiT fp8-e4m3 block GEMMs" if args.c5 else "EchoDiT bf16") + " + Fish S1-DAC decode fp32, random weights",
                       "parallelism": f"dp{world} (independent utterances, weight broadcast only)"},
            "distributed": {"ranks_seen": world, "backend": (dist.get_backend() if world > 1 else None),
                            "weight_broadcast": bcast or None},
            "roofline": roofline, "cpu_baseline": cpu, "eager_gpu_baseline": eager, "single_request": single,
            ("c3_share" if world == 1 else "c3"): c3, "c5": c5, "legacy_r1_workload": legacy, "phases": phases,
        }
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
