"""Shared definitions for the golden fixtures (sizes, sampler option sets, seeded inputs).

Used by tests/make_goldens.py (which runs the reference) and by the tests (which run the oracle
and the HIP engine on the same inputs).  Pure data + torch RNG; no reference code.
"""
from __future__ import annotations

import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import echo_ref as R  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")

TINY = R.DiTConfig(
    latent_size=80, model_size=256, num_layers=2, num_heads=2, intermediate_size=512, norm_eps=1e-5,
    text_vocab_size=256, text_model_size=256, text_num_layers=2, text_num_heads=2, text_intermediate_size=384,
    speaker_patch_size=4, speaker_model_size=256, speaker_num_layers=2, speaker_num_heads=2,
    speaker_intermediate_size=384, timestep_embed_size=128, adaln_rank=64)

WIDE1 = R.DiTConfig(num_layers=1, text_num_layers=1, speaker_num_layers=1)  # full widths, one layer each

TINY_DAC = R.DacConfig(latent_dim=128, decoder_dim=512, decoder_rates=(8, 8, 4, 2), post_layers=2, post_heads=2,
                       post_head_dim=64, post_ffn=256, post_window=8, post_block_size=256, upsample_factors=(2, 2),
                       # encode path (the reference hard-codes window 512 for the encoder transformer, autoencoder.py:855; channel counts
                       # are multiples of 32 = the fp32 GEMM K step of the HIP engine: 32 -> 64 -> 128 -> 256 -> 512)
                       encoder_dim=32, encoder_rates=(2, 4, 8, 8), encoder_transformer_layers=(0, 0, 0, 1), encoder_window=512,
                       encoder_block_size=16384, n_codebooks=2, codebook_size=16, codebook_dim=8, semantic_codebook_size=16)
TINY_ENC_SAMPLES = 2048 * 22 - 300        # 22 frames (padded), 88 encoder-transformer positions
FULL_ENC_SAMPLES = 512 * 600              # 600 encoder-transformer positions (> window 512), 150 frames


SAMPLER_CASES = {
    # name: kwargs   (all: tiny model, S=32, text 24 tokens padded to 40, speaker 32 latents)
    "cfg_default": dict(num_steps=6, cfg_scale_text=3.0, cfg_scale_speaker=8.0, cfg_min_t=0.5, cfg_max_t=1.0,
                        truncation_factor=None, rescale_k=None, rescale_sigma=None, speaker_kv_scale=None,
                        speaker_kv_max_layers=None, speaker_kv_min_t=None),
    "cfg_off": dict(num_steps=4, cfg_scale_text=3.0, cfg_scale_speaker=8.0, cfg_min_t=1.1, cfg_max_t=1.0,
                    truncation_factor=None, rescale_k=None, rescale_sigma=None, speaker_kv_scale=None,
                    speaker_kv_max_layers=None, speaker_kv_min_t=None),
    "all_options": dict(num_steps=6, cfg_scale_text=2.5, cfg_scale_speaker=5.0, cfg_min_t=0.4, cfg_max_t=0.9,
                        truncation_factor=0.8, rescale_k=1.2, rescale_sigma=3.0, speaker_kv_scale=1.5,
                        speaker_kv_max_layers=1, speaker_kv_min_t=0.6),
}


def tiny_inputs(cfg: R.DiTConfig, seed: int = 5, batch: int = 1, S: int = 32, tt: int = 40, tv: int = 24, ts: int = 32):
    g = torch.Generator().manual_seed(seed)
    ids = torch.zeros((batch, tt), dtype=torch.int32)
    tmask = torch.zeros((batch, tt), dtype=torch.bool)
    for b in range(batch):
        n = tv - 3 * b
        ids[b, 1:n] = torch.randint(32, 127, (n - 1,), generator=g, dtype=torch.int32)
        tmask[b, :n] = True
    spk = torch.randn((batch, ts, cfg.latent_size), generator=g)
    smask = torch.ones((batch, ts), dtype=torch.bool)
    for b in range(batch):
        smask[b, ts - 4 * b:] = False
    x0 = torch.randn((batch, S, cfg.latent_size), generator=g)
    return ids, tmask, spk, smask, x0


