"""Blockwise sampler with the reference's signature (inference_blockwise.py:14-123), on libechohip."""
from __future__ import annotations

from typing import List

import torch

from .inference import _multiply_kv_cache, build_schedule, run_euler
from .model import EchoDiT


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
) -> torch.Tensor:
    """Each block is one Euler run of `block_size` latents at positions start_pos.. that also attends to the
    latent-prefix KV (keys 4i < start_pos).  Text/speaker KV are encoded once.  The speaker-KV scale is re-applied
    at the start of every block exactly like the reference does (inference_blockwise.py:68-70)."""
    if not model.has_latent_encoder:
        raise RuntimeError("this checkpoint was loaded without the blockwise modules")
    device = model.device
    B = text_input_ids.shape[0]
    Lz = model.config.latent_size
    steps, temb = build_schedule(model, num_steps, cfg_min_t, cfg_max_t, rescale_k, rescale_sigma, speaker_kv_scale, speaker_kv_min_t)
    rng = torch.Generator(device=device).manual_seed(rng_seed)
    model.get_kv_cache_text(text_input_ids, text_mask)
    kv_spk = model.get_kv_cache_speaker(speaker_latent, speaker_mask)
    prefix = torch.zeros((B, sum(block_sizes), Lz), device=device, dtype=torch.float32)
    start = 0
    if continuation_latent is not None:
        start = continuation_latent.shape[1]
        prefix = torch.cat([continuation_latent.to(device, torch.float32), prefix], dim=1)
    ps = model.config.speaker_patch_size
    for bi, bs in enumerate(block_sizes):
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
    return prefix


def sample_blockwise(model, speaker_latent, speaker_mask, text_input_ids, text_mask, rng_seed, chunk_size: int = 160,
                     num_chunks: int = 4, **sampler_kwargs) -> torch.Tensor:
    """The README's `sample_blockwise(chunk_size=...)` (README.md:92-102) as a thin wrapper: block_sizes=[chunk_size]*k."""
    return sample_blockwise_euler_cfg_independent_guidances(model, speaker_latent, speaker_mask, text_input_ids, text_mask,
                                                            rng_seed, block_sizes=[chunk_size] * num_chunks, **sampler_kwargs)
