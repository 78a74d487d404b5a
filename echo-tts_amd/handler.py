"""Request-level helpers with the reference handler's semantics (handler.py of sruckh/echo-tts):
parameter defaults of the job schema, duration-aware chunking, boundary normalisation, cross-fade.
Network services of the reference handler (RunPod queue, S3 upload, ffmpeg/Opus, HF download) are out
of scope; `synthesize` returns the waveform and the metadata dict instead of a presigned URL.
"""
from __future__ import annotations

from functools import partial
from typing import Callable, Dict, List, Optional

import torch

from .inference import chunk_text, sample_euler_cfg_independent_guidances, sample_pipeline

SAMPLE_RATE = 44100

# job "parameters" keys and defaults (reference handler.py:426-443)
SAMPLER_DEFAULTS = dict(num_steps=40, cfg_scale_text=3.0, cfg_scale_speaker=8.0, cfg_min_t=0.5, cfg_max_t=1.0,
                        truncation_factor=None, rescale_k=None, rescale_sigma=None, speaker_kv_scale=None,
                        speaker_kv_max_layers=None, speaker_kv_min_t=None, sequence_length=640)


def _build_sample_fn(params: Dict, request_id: Optional[str] = None) -> Callable:
    return partial(sample_euler_cfg_independent_guidances, **{k: params.get(k, v) for k, v in SAMPLER_DEFAULTS.items()})


def chunk_text_for_audio(text: str, max_chars: int = 300, target_duration_seconds: float = 10.0) -> List[str]:
    """~12 characters per second of speech; a last piece shorter than 24 characters joins its predecessor
    (reference handler.py:102-123)."""
    pieces = chunk_text(text, max_chars=min(max_chars, int(target_duration_seconds * 12)))
    if len(pieces) > 1 and len(pieces[-1]) < 24:
        tail = pieces.pop()
        pieces[-1] += " " + tail
    return pieces


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


@torch.inference_mode()
def _sample_chunks_batched(model, fish_ae, pca_state, sample_fn, pieces: List[str], seed: int,
                           speaker_latent: Optional[torch.Tensor], speaker_mask: Optional[torch.Tensor], sequence_length: int,
                           max_batch: int) -> List[torch.Tensor]:
    """The text chunks of one request are independent (own seed seed + 1000 idx, same voice: handler.py:747-759), so up to
    `max_batch` of them go through ONE sampler call on the reference sampler's batch axis instead of one call each: the
    EchoDiT GEMMs then see 3 B 640 rows and fill the 256 CUs (157 instead of 97 audio-s/s on an MI355X).  Every row gets
    the noise the sequential path would draw for it (a (1, S, 80) draw from its own seed), text ids keep their 768 columns
    behind the key mask, decode and crop stay per chunk: each chunk's waveform equals the sequential one
    up to the engine's batch-shape noise (fp32 engine: <= 2e-5 RMS on latents; tests/test_gpu_engine.py)."""
    from .inference import ae_decode, crop_audio_to_flattening_point, get_text_input_ids_and_mask
    device, dtype = model.device, model.dtype
    lz = model.config.latent_size
    if speaker_latent is None:      # no reference voice: one masked key, as sample_pipeline does (inference.py:341-346)
        speaker_latent = torch.zeros((1, 4, lz), device=device, dtype=dtype)
        speaker_mask = torch.zeros((1, 4), device=device, dtype=torch.bool)
    out: List[torch.Tensor] = []
    for g0 in range(0, len(pieces), max_batch):
        grp = pieces[g0:g0 + max_batch]
        B = len(grp)
        ids, tmask = get_text_input_ids_and_mask(grp, max_length=768, device=device)     # as sample_pipeline: 768 columns + key mask
        x0 = torch.cat([torch.randn((1, sequence_length, lz), device=device, dtype=torch.float32,
                                    generator=torch.Generator(device=device).manual_seed(seed + (g0 + i) * 1000)) for i in range(B)], 0)
        lat = sample_fn(model, speaker_latent.to(device).expand(B, -1, -1).contiguous(), speaker_mask.to(device).expand(B, -1).contiguous(),
                        ids, tmask, seed + g0 * 1000, x_init=x0)
        for i in range(B):
            audio = ae_decode(fish_ae, pca_state, lat[i:i + 1])
            out.append(crop_audio_to_flattening_point(audio, lat[i])[0])
    return out


def synthesize(job_input: Dict, model, fish_ae, pca_state, speaker_latent: Optional[torch.Tensor] = None,
               speaker_mask: Optional[torch.Tensor] = None, speaker_audio: Optional[torch.Tensor] = None) -> Dict:
    """The compute part of the reference `_synthesize` (handler.py:682-803): validate, chunk, one sample_pipeline per
    chunk with seed + 1000*idx, normalise boundaries / cross-fade, return audio + metadata (or an error dict).
    `speaker_audio` (1, length) at 44.1 kHz is encoded ONCE per request on the GPU (the reference re-encodes the voice for
    every text chunk, handler.py:750-758); `speaker_latent` / `speaker_mask` pass an already encoded (cached) voice.
    parameters["max_chunk_batch"] (extension, default 8): chunks per sampler call; 1 = one call per chunk like the reference."""
    try:
        if speaker_latent is None and speaker_audio is not None:
            from .inference import get_speaker_latent_and_mask
            speaker_latent, speaker_mask = get_speaker_latent_and_mask(fish_ae, pca_state, speaker_audio.to(model.device))
            speaker_latent = speaker_latent.to(model.dtype)
        text = job_input.get("text")
        if not text or not str(text).strip():
            raise ValueError("text is required")
        if len(text) > 4000:
            raise ValueError("text must be at most 4000 characters")
        params = dict(job_input.get("parameters") or {})
        seed = int(params.get("seed", 0))
        pieces = chunk_text_for_audio(text, int(params.get("max_chars_per_chunk", 300)),
                                      float(params.get("target_duration_seconds", 10.0)))
        sample_fn = _build_sample_fn(params)
        max_batch = int(params.get("max_chunk_batch", 8))
        if max_batch > 1 and len(pieces) > 1:
            chunks = _sample_chunks_batched(model, fish_ae, pca_state, sample_fn, pieces, seed, speaker_latent, speaker_mask,
                                            int(params.get("sequence_length", SAMPLER_DEFAULTS["sequence_length"])), max_batch)
        else:
            chunks = []
            for idx, piece in enumerate(pieces):
                audio, _ = sample_pipeline(model, fish_ae, pca_state, sample_fn, piece, None, seed + idx * 1000,
                                           speaker_latent=speaker_latent, speaker_mask=speaker_mask)
                chunks.append(audio[0])
        if params.get("normalize_boundaries", True) and len(chunks) > 1:
            audio = normalize_chunk_boundaries(chunks)
        elif params.get("enable_crossfade", True) and len(chunks) > 1:
            audio = crossfade_chunks(chunks)
        else:
            audio = torch.cat(chunks, dim=-1)
        return {"audio": audio, "sample_rate": SAMPLE_RATE, "duration": audio.shape[-1] / SAMPLE_RATE, "chunks": len(pieces),
                "seed": seed, "text_length": len(text)}
    except Exception as e:  # same contract as handler.py:797-803
        import traceback
        return {"error": str(e), "error_type": type(e).__name__, "traceback": traceback.format_exc()}
