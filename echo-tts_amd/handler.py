"""Request surface of the reference handler (handler.py of sruckh/echo-tts) on the HIP engine: `handler(job)` /
`_synthesize(job_input, job_id)` with the reference's validation messages, parameter defaults, seed fallback and response
shape, duration-aware chunking, boundary normalisation and cross-fade.

What differs, and why:
* models come from `configure(model, fish_ae, pca_state, voices=...)` instead of `_load_models()`'s Hugging Face download
  (no network here; `inference.load_*_from_path` keep the checkpoint key layout);
* the response carries the waveform tensor (`"audio"`, on the device) where the reference returns the URL of an Opus file on
  S3 (`_save_and_upload_audio`: ffmpeg + boto3, out of scope) - every other key of `{"status", "metadata": {...}}` is kept;
* the chunks of one request are independent (own seed `seed + 1000 idx`, same voice: handler.py:747-759), so they go through
  ONE sampler call on the reference sampler's batch axis, the reference voice is encoded once and kept in a per-voice cache
  (speaker latents + the speaker KV of all 24 layers: the reference re-encodes it for every chunk), and flattening point,
  trailing-silence scan and cross-fade run as HIP kernels (csrc/postproc.hip) instead of per-sample Python loops.
"""
from __future__ import annotations

import ctypes as C
import traceback
from collections import OrderedDict
from functools import partial
from typing import Callable, Dict, List, Optional, Tuple

import torch

from . import _lib as L
from .inference import chunk_text, sample_euler_cfg_independent_guidances, sample_pipeline

SAMPLE_RATE = 44100

# job "parameters" keys and defaults (reference handler.py:426-443)
SAMPLER_DEFAULTS = dict(num_steps=40, cfg_scale_text=3.0, cfg_scale_speaker=8.0, cfg_min_t=0.5, cfg_max_t=1.0,
                        truncation_factor=None, rescale_k=None, rescale_sigma=None, speaker_kv_scale=None,
                        speaker_kv_max_layers=None, speaker_kv_min_t=None, sequence_length=640)


def _build_sample_fn(params: Dict, request_id: Optional[str] = None) -> Callable:
    """reference handler.py:426-443: `params.get(key, default)` for the four keys with non-None defaults, `params.get(key)`
    for the rest - an explicit None for sequence_length reaches the sampler, which reads it as 640 (inference.py:448)."""
    return partial(sample_euler_cfg_independent_guidances, **{k: params.get(k, v) for k, v in SAMPLER_DEFAULTS.items()})


def chunk_text_for_audio(text: str, max_chars: int = 300, target_duration_seconds: float = 10.0) -> List[str]:
    """~12 characters per second of speech; a last piece shorter than 24 characters joins its predecessor
    (reference handler.py:102-123)."""
    pieces = chunk_text(text, max_chars=min(max_chars, int(target_duration_seconds * 12)))
    if len(pieces) > 1 and len(pieces[-1]) < 24:
        tail = pieces.pop()
        pieces[-1] += " " + tail
    return pieces


# --------------------------------------------------------------------------- post-processing, torch form (any device)
def crossfade_chunks(audio_chunks: List[torch.Tensor], overlap_samples: int = 4410) -> torch.Tensor:
    """Linear cross-fade over min(overlap, len/4 of either side) samples (reference handler.py:126-170)."""
    if len(audio_chunks) <= 1:
        return torch.cat(audio_chunks, dim=-1) if audio_chunks else torch.tensor([])
    out = audio_chunks[0]
    for nxt in audio_chunks[1:]:
        n = min(overlap_samples, nxt.shape[-1] // 4, out.shape[-1] // 4)
        if n <= 0:
            out = torch.cat([out, nxt], dim=-1)
            continue
        down = torch.linspace(1, 0, n, device=nxt.device)
        up = torch.linspace(0, 1, n, device=nxt.device)
        if nxt.dim() == 2:
            down, up = down.view(1, -1), up.view(1, -1)
        mixed = out[..., -n:] * down + nxt[..., :n] * up
        out = torch.cat([out[..., :-n], mixed, nxt[..., n:]], dim=-1)
    return out


def _trailing_quiet(chunk: torch.Tensor, window: int, threshold: float) -> int:
    """Number of trailing samples (of the flattened last `window` samples) whose magnitude is below threshold.
    One vectorised pass instead of the reference's per-sample Python loop (handler.py:205-211)."""
    tail = chunk[..., -window:].abs().flatten()
    loud = torch.nonzero(tail >= threshold).flatten()
    return int(tail.numel() - 1 - int(loud[-1])) if loud.numel() else int(tail.numel())


def _boundary_length(n: int, quiet: int, min_silence_samples: int) -> int:
    """Length of a non-final chunk after handler.py:213-232: excess trailing silence removed, missing silence appended."""
    if quiet > min_silence_samples:
        return n - (quiet - min_silence_samples)
    return n + (min_silence_samples - quiet)     # covers quiet == 0 (append min_silence) and quiet == min (unchanged)


def normalize_chunk_boundaries(audio_chunks: List[torch.Tensor], sample_rate: int = 44100, silence_threshold: float = 0.01,
                               min_silence_samples: int = 22050) -> torch.Tensor:
    """Make every inner boundary end with exactly `min_silence_samples` of silence, then cross-fade
    (reference handler.py:173-240)."""
    if not audio_chunks:
        return torch.tensor([])
    if len(audio_chunks) == 1:
        return audio_chunks[0]
    fixed = []
    for i, chunk in enumerate(audio_chunks):
        if chunk.dim() == 1:
            chunk = chunk.unsqueeze(0)
        if i < len(audio_chunks) - 1:
            quiet = _trailing_quiet(chunk, min(chunk.shape[-1], 2 * min_silence_samples), silence_threshold)
            if quiet > min_silence_samples:
                chunk = chunk[..., : -(quiet - min_silence_samples)]
            elif quiet < min_silence_samples:
                pad = torch.zeros(*chunk.shape[:-1], min_silence_samples - quiet, device=chunk.device)
                chunk = torch.cat([chunk, pad], dim=-1)
        fixed.append(chunk)
    return crossfade_chunks(fixed)


# --------------------------------------------------------------------------- post-processing, HIP form (device chunks)
def _device_rows(audio_chunks: List[torch.Tensor]) -> Optional[List[torch.Tensor]]:
    """The chunks as contiguous 1-D fp32 device rows, or None when the HIP path does not apply (host tensors, multi-channel
    chunks, more than 64 chunks): the callers then use the torch form above."""
    if not audio_chunks or len(audio_chunks) > 64:
        return None
    rows = []
    for c in audio_chunks:
        if not c.is_cuda or c.dtype != torch.float32 or (c.dim() > 1 and c.numel() != c.shape[-1]):
            return None
        rows.append(c.reshape(-1).contiguous())
    return rows


def trailing_quiet_device(rows: List[torch.Tensor], max_window: int, threshold: float) -> List[int]:
    """handler.py:199-211 for all chunks in one launch + one read-back of len(rows) ints."""
    n = len(rows)
    ptrs = (C.c_void_p * n)(*[r.data_ptr() for r in rows])
    lens = (L.c_i64 * n)(*[r.numel() for r in rows])
    out = torch.empty((n,), dtype=torch.int32, device=rows[0].device)
    L.check(L.load_library().echo_op_trailing_quiet(ptrs, lens, n, int(max_window), float(threshold), out.data_ptr(),
                                                    torch.cuda.current_stream(rows[0].device).cuda_stream))
    return [int(v) for v in out.tolist()]


def _assemble_device(rows: List[torch.Tensor], lens: List[int], overlap_samples: int, two_d: bool) -> torch.Tensor:
    """crossfade_chunks (handler.py:126-170) over chunks whose effective lengths `lens` may be shorter (trimmed) or longer
    (zero-extended) than the stored rows: one kernel writes the final waveform."""
    n = len(rows)
    starts, ovs, acc = [0], [], lens[0]
    for i in range(1, n):
        ov = max(0, min(overlap_samples, lens[i] // 4, acc // 4))
        ovs.append(ov)
        starts.append(acc - ov)
        acc = acc - ov + lens[i]
    ovs.append(0)
    valid = [min(lens[i], rows[i].numel()) for i in range(n)]
    out = torch.empty((acc,), dtype=torch.float32, device=rows[0].device)
    L.check(L.load_library().echo_op_assemble_chunks((C.c_void_p * n)(*[r.data_ptr() for r in rows]), (L.c_i64 * n)(*starts),
                                                     (L.c_i64 * n)(*lens), (L.c_i64 * n)(*valid), (C.c_int32 * n)(*ovs), n, out.data_ptr(), acc,
                                                     torch.cuda.current_stream(rows[0].device).cuda_stream))
    return out.view(1, -1) if two_d else out


def crossfade_chunks_device(audio_chunks: List[torch.Tensor], overlap_samples: int = 4410) -> torch.Tensor:
    rows = _device_rows(audio_chunks)
    if rows is None or len(rows) <= 1:
        return crossfade_chunks(audio_chunks, overlap_samples)
    return _assemble_device(rows, [r.numel() for r in rows], overlap_samples, audio_chunks[0].dim() == 2)


def normalize_chunk_boundaries_device(audio_chunks: List[torch.Tensor], sample_rate: int = 44100, silence_threshold: float = 0.01,
                                      min_silence_samples: int = 22050) -> torch.Tensor:
    rows = _device_rows(audio_chunks)
    if rows is None or len(rows) <= 1:
        return normalize_chunk_boundaries(audio_chunks, sample_rate, silence_threshold, min_silence_samples)
    quiet = trailing_quiet_device(rows[:-1], 2 * min_silence_samples, silence_threshold)
    lens = [_boundary_length(rows[i].numel(), quiet[i], min_silence_samples) for i in range(len(rows) - 1)] + [rows[-1].numel()]
    return _assemble_device(rows, lens, 4410, True)        # the reference makes every chunk 2-D here (handler.py:196-197)


# --------------------------------------------------------------------------- per-voice cache
class VoiceCache:
    """voice id -> speaker latents (+ mask) and, per engine context family, the speaker KV of all EchoDiT layers.

    The reference handler loads and DAC-encodes the voice file and runs the speaker encoder + 24 K/V projections for EVERY text
    chunk of every request (handler.py:720,750-758 -> inference.py:333-340 -> model.py:615-621: ~9 TFLOP for a 2-minute
    voice); here both results are computed on first use and reused (LRU, bounded by `max_bytes` of device memory)."""

    def __init__(self, max_bytes: int = 8 << 30):
        self.max_bytes = int(max_bytes)
        self._latents: "OrderedDict[str, Tuple[torch.Tensor, torch.Tensor]]" = OrderedDict()
        self._kv: "OrderedDict[Tuple[str, int], object]" = OrderedDict()
        self.hits = self.misses = 0

    def latent(self, voice_id: str) -> Optional[Tuple[torch.Tensor, torch.Tensor]]:
        return self._latents.get(voice_id)

    def put_latent(self, voice_id: str, speaker_latent: torch.Tensor, speaker_mask: torch.Tensor) -> None:
        self._latents[voice_id] = (speaker_latent, speaker_mask)
        for key in [k for k in self._kv if k[0] == voice_id]:      # a re-registered voice invalidates its KV
            self._kv.pop(key).close()

    def nbytes(self) -> int:
        return sum(v.nbytes for v in self._kv.values()) + sum(l.numel() * l.element_size() for l, _ in self._latents.values())

    @torch.inference_mode()
    def speaker_kv(self, model, voice_id: str):
        """The voice's KV for `model` (captured from a fresh encode on the first request, bound afterwards)."""
        key = (voice_id, id(model))
        h = self._kv.get(key)
        if h is not None:
            self._kv.move_to_end(key)
            self.hits += 1
            return h
        self.misses += 1
        lat, mask = self._latents[voice_id]
        kv = model.get_kv_cache_speaker(lat.to(model.device, model.dtype), mask)
        h = model.capture_voice(kv)
        self._kv[key] = h
        while self.nbytes() > self.max_bytes and len(self._kv) > 1:
            self._kv.popitem(last=False)[1].close()
        return h


# --------------------------------------------------------------------------- sampling the chunks of one request
@torch.inference_mode()
def _sample_chunks_batched(model, fish_ae, pca_state, sample_fn, pieces: List[str], seed: int,
                           speaker_latent: Optional[torch.Tensor], speaker_mask: Optional[torch.Tensor], sequence_length: int,
                           max_batch: int, speaker_kv=None, indices: Optional[List[int]] = None) -> List[torch.Tensor]:
    """Up to `max_batch` chunks per sampler call.  Every row gets the noise the sequential path would draw for it (a (1, S, 80)
    draw from its own seed), text ids keep their 768 columns behind the key mask, the one voice is shared by all rows (encoded
    once with batch 1, or bound from the cache), decode stays per chunk; the flattening points of the whole batch come from one
    kernel launch.  Each chunk's waveform equals the sequential one up to the engine's batch-shape noise (fp32 engine:
    <= 2e-5 RMS on latents; tests/test_gpu_engine.py)."""
    from .inference import ae_decode, find_flattening_points, get_text_input_ids_and_mask
    device, dtype = model.device, model.dtype
    lz = model.config.latent_size
    if speaker_latent is None and speaker_kv is None:      # no reference voice: one masked key, as sample_pipeline does (inference.py:341-346)
        speaker_latent = torch.zeros((1, 4, lz), device=device, dtype=dtype)
        speaker_mask = torch.zeros((1, 4), device=device, dtype=torch.bool)
    out: List[torch.Tensor] = []
    idx = list(range(len(pieces))) if indices is None else list(indices)      # chunk numbers within the request: they set the seeds
    for g0 in range(0, len(pieces), max_batch):
        grp = pieces[g0:g0 + max_batch]
        B = len(grp)
        ids, tmask = get_text_input_ids_and_mask(grp, max_length=768, device=device)     # as sample_pipeline: 768 columns + key mask
        x0 = torch.cat([torch.randn((1, sequence_length, lz), device=device, dtype=torch.float32,
                                    generator=torch.Generator(device=device).manual_seed(seed + idx[g0 + i] * 1000)) for i in range(B)], 0)
        lat = sample_fn(model, speaker_latent, speaker_mask, ids, tmask, seed + idx[g0] * 1000, x_init=x0, speaker_kv=speaker_kv)
        cut = find_flattening_points(lat)
        for i in range(B):
            audio = ae_decode(fish_ae, pca_state, lat[i:i + 1])
            out.append(audio[0][..., : cut[i] * 2048])
    return out


def _run_job(job_input: Dict, model, fish_ae, pca_state, speaker_latent, speaker_mask, speaker_kv=None) -> Tuple[torch.Tensor, int, int]:
    """handler.py:702-767 from the parameters on: (audio (1, n), seed, number of chunks)."""
    text = job_input.get("text")
    parameters = job_input.get("parameters") or {}
    seed = parameters.get("seed", job_input.get("seed", 0))                 # handler.py:702
    seed = int(seed)
    sample_fn = _build_sample_fn(parameters)
    max_chars_raw = parameters.get("max_chars_per_chunk", 300)
    enable_crossfade = parameters.get("enable_crossfade", True)
    normalize_boundaries = parameters.get("normalize_boundaries", True)
    target_duration = parameters.get("target_duration_seconds", 10.0)
    try:
        max_chars = int(max_chars_raw)
    except Exception:
        max_chars = 300                                                      # handler.py:729-732
    pieces = chunk_text_for_audio(text, max_chars=max_chars, target_duration_seconds=target_duration) if max_chars and max_chars > 0 else [text]
    if not pieces:
        raise ValueError("Text is empty after normalization")
    seq = parameters.get("sequence_length", SAMPLER_DEFAULTS["sequence_length"])
    seq = 640 if seq is None else int(seq)
    max_batch = int(parameters.get("max_chunk_batch", 8))
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1 and len(pieces) > 1 and parameters.get("data_parallel", True):
        # the chunks of a request are independent units with their own seeds (handler.py:747-759).  Every rank of the job
        # calls the handler with the SAME request; chunk i goes to rank i % world (parallel.shard_units), each rank runs its share as
        # rows of one sampler call, and rank 0 receives the waveforms in chunk order and assembles them.  The other ranks return
        # (None, seed, n): they did their part.  No collective inside the sampler.
        from . import parallel as P

        def work_many(units: List[int]) -> Dict[int, torch.Tensor]:
            wav = _sample_chunks_batched(model, fish_ae, pca_state, sample_fn, [pieces[u] for u in units], seed, speaker_latent, speaker_mask,
                                         seq, max(1, max_batch), speaker_kv=speaker_kv, indices=units)
            return {u: w_.reshape(-1) for u, w_ in zip(units, wav)}
        gathered = P.run_data_parallel_batched(len(pieces), work_many)
        if gathered is None:
            return None, seed, len(pieces)
        chunks = [w_.to(model.device).reshape(1, -1) for w_ in gathered]
    elif max_batch > 1 and len(pieces) > 1:
        chunks = _sample_chunks_batched(model, fish_ae, pca_state, sample_fn, pieces, seed, speaker_latent, speaker_mask, seq, max_batch,
                                        speaker_kv=speaker_kv)
    else:
        chunks = []
        if speaker_kv is not None:
            sample_fn = partial(sample_fn, speaker_kv=speaker_kv)
        for idx, piece in enumerate(pieces):
            audio, _ = sample_pipeline(model, fish_ae, pca_state, sample_fn, piece, None, seed + idx * 1000,
                                       speaker_latent=speaker_latent, speaker_mask=speaker_mask)
            chunks.append(audio[0])
    if normalize_boundaries and len(chunks) > 1:
        audio = normalize_chunk_boundaries_device(chunks, sample_rate=44100)
    elif enable_crossfade and len(chunks) > 1:
        audio = crossfade_chunks_device(chunks)
    else:
        audio = torch.cat(chunks, dim=-1)
    if audio.dim() == 1:
        audio = audio.unsqueeze(0)
    return audio, seed, len(pieces)


def _is_worker_rank() -> bool:
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1 and dist.get_rank() != 0


def _validate_text(job_input: Dict) -> Optional[Dict]:
    """The reference's input checks with its messages (handler.py:690-698)."""
    text = job_input.get("text")
    if not text or not isinstance(text, str):
        return {"error": "Missing or invalid 'text' field (expected string)"}
    if len(text.strip()) == 0:
        return {"error": "Text cannot be empty"}
    if len(text) > 4000:
        return {"error": f"Text too long: {len(text)} characters (max 4000)"}
    return None


def synthesize(job_input: Dict, model, fish_ae, pca_state, speaker_latent: Optional[torch.Tensor] = None,
               speaker_mask: Optional[torch.Tensor] = None, speaker_audio: Optional[torch.Tensor] = None) -> Dict:
    """The compute part of the reference `_synthesize` with the models passed explicitly (tests, embedding in another server).
    `speaker_audio` (1, length) at 44.1 kHz is encoded ONCE per request on the GPU; `speaker_latent` / `speaker_mask` pass an
    already encoded voice.  parameters["max_chunk_batch"] (extension, default 8): chunks per sampler call; 1 = one call per
    chunk like the reference.  Returns audio + flat metadata, or the reference's error dict (handler.py:797-803)."""
    try:
        bad = _validate_text(job_input)
        if bad:
            raise ValueError(bad["error"])
        if speaker_latent is None and speaker_audio is not None:
            from .inference import get_speaker_latent_and_mask
            speaker_latent, speaker_mask = get_speaker_latent_and_mask(fish_ae, pca_state, speaker_audio.to(model.device))
            speaker_latent = speaker_latent.to(model.dtype)
        audio, seed, n = _run_job(job_input, model, fish_ae, pca_state, speaker_latent, speaker_mask)
        if audio is None:                  # a worker rank of a data-parallel job: rank 0 holds the assembled audio
            return {"status": "worker", "chunks": n, "seed": seed}
        return {"audio": audio, "sample_rate": SAMPLE_RATE, "duration": audio.shape[-1] / SAMPLE_RATE, "chunks": n,
                "seed": seed, "text_length": len(job_input.get("text"))}
    except Exception as e:  # same contract as handler.py:797-803
        return {"error": str(e), "error_type": type(e).__name__, "traceback": traceback.format_exc()}


# --------------------------------------------------------------------------- the reference's entry points
class _State:
    model = None
    fish_ae = None
    pca_state = None
    voices: Dict[str, object] = {}
    voice_loader: Optional[Callable[[str], torch.Tensor]] = None
    cache: Optional[VoiceCache] = None


def configure(model, fish_ae, pca_state, voices: Optional[Dict[str, object]] = None,
              voice_loader: Optional[Callable[[str], torch.Tensor]] = None, voice_cache: Optional[VoiceCache] = None) -> None:
    """Stands in for the reference's `_load_models()` (handler.py:323-417, Hugging Face download): the loaded engine objects and
    the voice registry.  `voices` maps a `speaker_voice` name to 44.1 kHz audio (1, n) or to an already encoded
    (speaker_latent, speaker_mask) pair; `voice_loader(name)` is asked for names the registry lacks (returns audio or None)."""
    _State.model, _State.fish_ae, _State.pca_state = model, fish_ae, pca_state
    _State.voices = dict(voices or {})
    _State.voice_loader = voice_loader
    _State.cache = voice_cache or VoiceCache()


def _load_models():
    if _State.model is None:
        raise RuntimeError("models are not loaded: call echo_tts_amd.handler.configure(model, fish_ae, pca_state) first "
                           "(the reference downloads them from Hugging Face, handler.py:323-417)")
    return _State.model, _State.fish_ae, _State.pca_state


def health_check() -> Dict:
    """Compute-side subset of the reference's health check (handler.py:600-680; its S3 / ffmpeg probes are out of scope)."""
    ok = _State.model is not None and torch.cuda.is_available()
    return {"status": "healthy" if ok else "unhealthy",
            "checks": {"models_loaded": _State.model is not None, "gpu": torch.cuda.get_device_name(0) if torch.cuda.is_available() else None,
                       "voices": sorted(_State.voices), "voice_cache_bytes": _State.cache.nbytes() if _State.cache else 0}}


@torch.inference_mode()
def _resolve_voice(name: str, model, fish_ae, pca_state):
    """speaker_voice -> VoiceHandle through the per-voice cache; None when the name is unknown."""
    cache = _State.cache
    if cache.latent(name) is None:
        src = _State.voices.get(name)
        if src is None and _State.voice_loader is not None:
            src = _State.voice_loader(name)
        if src is None:
            return None
        if isinstance(src, (tuple, list)):
            lat, mask = src
        else:
            from .inference import get_speaker_latent_and_mask
            lat, mask = get_speaker_latent_and_mask(fish_ae, pca_state, src.to(model.device))
        cache.put_latent(name, lat.to(model.device, model.dtype), mask)
    return cache.speaker_kv(model, name)


def _synthesize(job_input: Dict, job_id: Optional[str] = None) -> Dict:
    """reference handler.py:682-803."""
    if job_input.get("action") == "health_check":
        return health_check()
    bad = _validate_text(job_input)
    if bad:
        return bad
    speaker_voice_name = job_input.get("speaker_voice")
    try:
        model, fish_ae, pca_state = _load_models()
        speaker_kv = None
        if speaker_voice_name:
            speaker_kv = _resolve_voice(str(speaker_voice_name), model, fish_ae, pca_state)
            if speaker_kv is None:
                return {"error": f"speaker_voice '{speaker_voice_name}' not found"}
        audio_out, seed, n_chunks = _run_job(job_input, model, fish_ae, pca_state, None, None, speaker_kv=speaker_kv)
        if audio_out is None and _is_worker_rank():
            return {"status": "worker", "metadata": {"seed": seed, "device": str(model.device), "chunks": n_chunks}}
        if audio_out is None or len(audio_out) == 0:
            return {"error": "No audio generated"}
        duration_seconds = len(audio_out[0]) / 44_100
        return {
            "status": "completed",
            "audio": audio_out,                   # the reference uploads an Opus file and returns filename / url / s3_key here
            "metadata": {
                "sample_rate": SAMPLE_RATE,       # of `audio`; the reference reports its 24 kHz Opus re-encode (handler.py:787)
                "duration": duration_seconds,
                "seed": seed,
                "device": str(model.device),
                "chunks": n_chunks,
            },
        }
    except Exception as e:
        return {"error": str(e), "error_type": type(e).__name__, "traceback": traceback.format_exc()}


def handler(job: Dict) -> Dict:
    """reference handler.py:806-817 (RunPod entry point)."""
    try:
        return _synthesize(job.get("input", {}), job.get("id"))
    except Exception as e:
        return {"error": str(e), "error_type": type(e).__name__}
