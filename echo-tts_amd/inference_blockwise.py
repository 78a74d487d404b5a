"""Blockwise sampler with the reference's signature (inference_blockwise.py:14-123), on libechohip."""
from __future__ import annotations

from typing import Iterator, List, Tuple

import torch

from .inference import _multiply_kv_cache, build_schedule, run_euler
from .model import EchoDiT


def sample_blockwise_stream(
    model: EchoDiT,
    speaker_latent: torch.Tensor,
    speaker_mask: torch.Tensor,
    text_input_ids: torch.Tensor,
    text_mask: torch.Tensor,
    rng_seed: int,
    block_sizes: List[int],
    num_steps: int,
    cfg_scale_text: float,
    cfg_scale_speaker: float,
    cfg_min_t: float,
    cfg_max_t: float,
    truncation_factor: float | None,
    rescale_k: float | None,
    rescale_sigma: float | None,
    speaker_kv_scale: float | None,
    speaker_kv_max_layers: int | None,
    speaker_kv_min_t: float | None,
    continuation_latent: torch.Tensor | None = None,
    x_inits: List[torch.Tensor] | None = None,
    speaker_kv=None,
) -> Iterator[Tuple[int, torch.Tensor, torch.Tensor]]:
    """Generator form of the reference's blockwise sampler (inference_blockwise.py:59-121): yields
    `(start_pos, block_latents (B, block_size, 80), prefix_so_far (B, start_pos + block_size, 80))` as soon as each block's
    Euler run has been enqueued, so a consumer (DACStream) can decode and play block k while block k + 1 is sampled.

    Each block is one Euler run of `block_size` latents at positions start_pos.. that also attends to the latent-prefix KV
    (keys 4i < start_pos).  Text/speaker KV are encoded once (or bound from a cached voice, `speaker_kv`).  The speaker-KV scale
    is re-applied at the start of every block exactly like the reference does (inference_blockwise.py:68-70)."""
    if not model.has_latent_encoder:
        raise RuntimeError("this checkpoint was loaded without the blockwise modules")
    with torch.inference_mode():
        device = model.device
        B = text_input_ids.shape[0]
        Lz = model.config.latent_size
        steps, temb = build_schedule(model, num_steps, cfg_min_t, cfg_max_t, rescale_k, rescale_sigma, speaker_kv_scale, speaker_kv_min_t)
        rng = torch.Generator(device=device).manual_seed(rng_seed)
        model.get_kv_cache_text(text_input_ids, text_mask)
        kv_spk = model.bind_voice(speaker_kv) if speaker_kv is not None else model.get_kv_cache_speaker(speaker_latent, speaker_mask)
        prefix = torch.zeros((B, sum(block_sizes), Lz), device=device, dtype=torch.float32)
        start = 0
        if continuation_latent is not None:
            start = continuation_latent.shape[1]
            prefix = torch.cat([continuation_latent.to(device, torch.float32), prefix], dim=1)
        ps = model.config.speaker_patch_size
    for bi, bs in enumerate(block_sizes):
        with torch.inference_mode():
            if speaker_kv_scale is not None:
                _multiply_kv_cache(kv_spk, speaker_kv_scale, speaker_kv_max_layers)
            model.get_kv_cache_latent(prefix, n_latents=(start + ps - 1) // ps * ps)
            if x_inits is None:
                x0 = torch.randn((B, bs, Lz), device=device, dtype=torch.float32, generator=rng)
            else:
                x0 = x_inits[bi].to(device, torch.float32)
            x = run_euler(model, x0, steps, temb, cfg_scale_text, cfg_scale_speaker, truncation_factor, speaker_kv_scale,
                          speaker_kv_max_layers, start_pos=start, use_latent=True)
            prefix[:, start:start + bs] = x
            start += bs
        yield start - bs, x, prefix[:, :start]


@torch.inference_mode()
def sample_blockwise_euler_cfg_independent_guidances(
    model: EchoDiT,
    speaker_latent: torch.Tensor,
    speaker_mask: torch.Tensor,
    text_input_ids: torch.Tensor,
    text_mask: torch.Tensor,
    rng_seed: int,
    block_sizes: List[int],
    num_steps: int,
    cfg_scale_text: float,
    cfg_scale_speaker: float,
    cfg_min_t: float,
    cfg_max_t: float,
    truncation_factor: float | None,
    rescale_k: float | None,
    rescale_sigma: float | None,
    speaker_kv_scale: float | None,
    speaker_kv_max_layers: int | None,
    speaker_kv_min_t: float | None,
    continuation_latent: torch.Tensor | None = None,
    x_inits: List[torch.Tensor] | None = None,
    speaker_kv=None,
) -> torch.Tensor:
    """Drop-in for reference inference_blockwise.py:14-123: (B, continuation + sum(block_sizes), 80) fp32."""
    full = None
    for _, _, full in sample_blockwise_stream(model, speaker_latent, speaker_mask, text_input_ids, text_mask, rng_seed, block_sizes,
                                              num_steps, cfg_scale_text, cfg_scale_speaker, cfg_min_t, cfg_max_t, truncation_factor,
                                              rescale_k, rescale_sigma, speaker_kv_scale, speaker_kv_max_layers, speaker_kv_min_t,
                                              continuation_latent=continuation_latent, x_inits=x_inits, speaker_kv=speaker_kv):
        pass
    if full is None:      # no blocks: the (possibly empty) continuation alone
        B, Lz = text_input_ids.shape[0], model.config.latent_size
        full = continuation_latent.to(model.device, torch.float32) if continuation_latent is not None else torch.zeros((B, 0, Lz), device=model.device)
    return full


def stream_audio_blockwise(model: EchoDiT, fish_ae, pca_state, speaker_latent, speaker_mask, text_input_ids, text_mask, rng_seed: int,
                           block_sizes: List[int], **sampler_kwargs) -> Iterator[torch.Tensor]:
    """Streaming synthesis of one utterance (batch 1): yields the (1, 1, block_size * 2048) waveform of every block right after
    its latents exist; the concatenation equals `ae_decode` of the whole blockwise latent (causal chunked DAC decode, DACStream).
    Time to first audio = one block's sampler run + one block's decode instead of the whole utterance's."""
    from .autoencoder import DACStream
    if text_input_ids.shape[0] != 1:
        raise ValueError("stream_audio_blockwise streams one utterance; run one generator per request")
    dec = DACStream(fish_ae, pca_state)
    cont = sampler_kwargs.get("continuation_latent")
    if cont is not None and cont.shape[1] > 0:
        dec.push(cont[0])                      # the continuation's audio exists already: only its decoder context is needed
    for _, block, _ in sample_blockwise_stream(model, speaker_latent, speaker_mask, text_input_ids, text_mask, rng_seed, block_sizes,
                                               **sampler_kwargs):
        yield dec.push(block[0])


def sample_blockwise(model, speaker_latent, speaker_mask, text_input_ids, text_mask, rng_seed, chunk_size: int = 160,
                     num_chunks: int = 4, **sampler_kwargs) -> torch.Tensor:
    """The README's `sample_blockwise(chunk_size=...)` (README.md:92-102) as a thin wrapper: block_sizes=[chunk_size]*k."""
    return sample_blockwise_euler_cfg_independent_guidances(model, speaker_latent, speaker_mask, text_input_ids, text_mask,
                                                            rng_seed, block_sizes=[chunk_size] * num_chunks, **sampler_kwargs)
