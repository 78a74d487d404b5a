"""CPU oracle for the Echo-TTS hot path.  TEST INFRASTRUCTURE ONLY.

This file is a plain-PyTorch *restatement* of the algorithm on the reference's hot
path, written from SURVEY.md and from reading the reference as text.  It is the
checker for the HIP engine: only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it.  Nothing under ``echo-tts_amd/``
(the product) imports, links or executes anything in ``oracle/``.

Pinning: the reference ships no tests and no golden vectors, and the
trained checkpoints are gated / absent, so the oracle is pinned against outputs of
the reference itself, imported in the build container on CPU with seeded random
weights (``tests/make_goldens.py`` -> ``tests/golden/*.safetensors``).  On those
fixtures the oracle is bit-identical to the reference (``tests/test_oracle_golden.py``).
Trained-weight parity: unpinned.

Style: functional, weights are a flat ``dict[str, Tensor]`` keyed by the reference's
state-dict names.  Every function cites the reference lines it
follows (paths relative to /root/reference).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
Weights = Dict[str, Tensor]
KV = List[Tuple[Tensor, Tensor]]


# --------------------------------------------------------------------------- config
@dataclass
class DiTConfig:
    """Sizes of EchoDiT; defaults are inference.py:16-24."""
    latent_size: int = 80
    model_size: int = 2048
    num_layers: int = 24
    num_heads: int = 16
    intermediate_size: int = 5888
    norm_eps: float = 1e-5
    text_vocab_size: int = 256
    text_model_size: int = 1280
    text_num_layers: int = 14
    text_num_heads: int = 10
    text_intermediate_size: int = 3328
    speaker_patch_size: int = 4
    speaker_model_size: int = 1280
    speaker_num_layers: int = 14
    speaker_num_heads: int = 10
    speaker_intermediate_size: int = 3328
    timestep_embed_size: int = 512
    adaln_rank: int = 256

    @property
    def head_dim(self) -> int:
        return self.model_size // self.num_heads


@dataclass
class DacConfig:
    """Sizes of the Fish S1-DAC decode path; defaults are autoencoder.py:1144-1192."""
    latent_dim: int = 1024
    decoder_dim: int = 1536
    decoder_rates: Tuple[int, ...] = (8, 8, 4, 2)
    post_layers: int = 8
    post_heads: int = 16
    post_head_dim: int = 64
    post_ffn: int = 3072
    post_window: int = 128
    post_block_size: int = 4096
    upsample_factors: Tuple[int, ...] = (2, 2)
    norm_eps: float = 1e-5
    rope_base: float = 10000.0
    # encode path (speaker reference -> latents): Encoder + downsample + pre_module + RVQ; autoencoder.py:1144-1192
    encoder_dim: int = 64
    encoder_rates: Tuple[int, ...] = (2, 4, 8, 8)
    encoder_transformer_layers: Tuple[int, ...] = (0, 0, 0, 4)
    encoder_window: int = 512
    encoder_block_size: int = 16384
    n_codebooks: int = 9
    codebook_size: int = 1024
    codebook_dim: int = 8
    semantic_codebook_size: int = 4096

    @property
    def encoder_hop(self) -> int:
        h = 1
        for r in self.encoder_rates:
            h *= r
        return h

    @property
    def hop(self) -> int:
        h = 1
        for r in self.decoder_rates:
            h *= r
        for f in self.upsample_factors:
            h *= f
        return h


# ------------------------------------------------------------------- DiT primitives
def rope_table(head_dim: int, end: int, theta: float = 10000.0) -> Tensor:
    """complex64 (end, head_dim/2) table of cis(pos * theta_j).  model.py:9-14."""
    inv = 1.0 / (theta ** (torch.arange(0, head_dim, 2)[: head_dim // 2] / head_dim))
    ang = torch.outer(torch.arange(end), inv)
    return torch.complex(torch.cos(ang), torch.sin(ang))


def rotate_pairs(x: Tensor, fc: Tensor) -> Tensor:
    """Interleaved-pair complex rotation in fp32, cast back.  x (b,s,h,d), fc (s,d/2).  model.py:17-24."""
    xc = torch.view_as_complex(x.float().reshape(*x.shape[:3], -1, 2))
    xc = xc * fc[..., None, :]
    return torch.view_as_real(xc).reshape(x.shape).type_as(x)


def rotate_first_half_of_heads(y: Tensor, fc: Tensor) -> Tensor:
    """RoPE on heads [0, H/2) only (chunk along the HEAD axis).  model.py:199-202."""
    a, b = y.chunk(2, dim=-2)
    return torch.cat([rotate_pairs(a, fc), b], dim=-2)


def timestep_embedding(t: Tensor, size: int) -> Tensor:
    """[cos | sin](t * 1000 * exp(-ln(1e4) j / half)), cast to t.dtype.  model.py:27-43."""
    half = size // 2
    freqs = 1000 * torch.exp(
        -torch.log(torch.tensor(10000.0)) * torch.arange(start=0, end=half, dtype=torch.float32) / half
    ).to(t.device)
    args = t[..., None] * freqs[None]
    return torch.cat([torch.cos(args), torch.sin(args)], dim=-1).to(t.dtype)


def rms_norm(x: Tensor, weight: Tensor, eps: float) -> Tensor:
    """fp32 RMS norm incl. the weight multiply, then cast back.  model.py:99-104."""
    dt = x.dtype
    x = x.float()
    x = x * torch.rsqrt(torch.pow(x.float(), 2).mean(dim=-1, keepdim=True) + eps)
    x = x * weight
    return x.to(dt)


def lowrank_adaln(w: Weights, p: str, x: Tensor, cond: Tensor, eps: float) -> Tuple[Tensor, Tensor]:
    """LowRankAdaLN.forward.  model.py:64-83.  Returns (modulated x, tanh gate)."""
    shift, scale, gate = cond.chunk(3, dim=-1)

    def refine(name: str, c: Tensor) -> Tensor:
        h = F.linear(F.silu(c), w[f"{p}.{name}_down.weight"])
        return F.linear(h, w[f"{p}.{name}_up.weight"], w[f"{p}.{name}_up.bias"]) + c

    shift, scale, gate = refine("shift", shift), refine("scale", scale), refine("gate", gate)
    if _LINEAR_TAPS is not None:
        _LINEAR_TAPS[f"{p}.in"] = x.clone()
        _LINEAR_TAPS[f"{p}.scale1p"] = (scale + 1).clone()
        _LINEAR_TAPS[f"{p}.shift"] = shift.clone()
    dt = x.dtype
    x = x.float()
    x = x * torch.rsqrt(torch.pow(x.float(), 2).mean(dim=-1, keepdim=True) + eps)
    x = x * (scale + 1) + shift
    return x.to(dt), torch.tanh(gate)


# BASELINE config C5 ("fp8 MFMA weight path for DiT GEMMs") has no counterpart in the reference; this is the restatement of the
# algorithm the HIP engine's fp8 mode states: the four large linears of every EchoDiT block take OCP e4m3 operands,
# weights scaled per output row and activations per token row by amax / 448, products accumulated in fp32, result rounded to the
# model dtype.  Off by default; tests switch it on to pin the engine's fp8 path to something other than itself.
_FP8_BLOCK_LINEARS = False
_FP8_ACT_STATIC: Optional[Dict[str, float]] = None


def set_fp8_block_linears(on: bool, act_static: Optional[Dict[str, float]] = None) -> None:
    """C5 restatement switch.  `act_static` (engine option `EchoDiT.set_fp8_static_scales`): {"<block prefix>.wo": s, "<block prefix>.w2": s}
    - the activation operand of that linear is quantised with the one calibrated scale s (saturating at 448 s) instead of per token row."""
    global _FP8_BLOCK_LINEARS, _FP8_ACT_STATIC
    _FP8_BLOCK_LINEARS = bool(on)
    _FP8_ACT_STATIC = dict(act_static) if (on and act_static) else None


def fake_quant_rows_e4m3(x: Tensor) -> Tensor:
    xf = x.float()
    s = xf.abs().amax(dim=-1, keepdim=True).clamp_min(1e-30) / 448.0
    return (xf / s).to(torch.float8_e4m3fn).float() * s


def fake_quant_static_e4m3(x: Tensor, s: float) -> Tensor:
    return (x.float() / s).clamp(-448.0, 448.0).to(torch.float8_e4m3fn).float() * s


# Teacher-forcing taps (tests only): when a dict is installed, every block linear records its input and output under
# "<block prefix>.<linear>.in" / ".out", so that a test can feed exactly these operands through the engine's single kernels.
_LINEAR_TAPS: Optional[Dict[str, Tensor]] = None


def set_linear_taps(taps: Optional[Dict[str, Tensor]]) -> None:
    global _LINEAR_TAPS
    _LINEAR_TAPS = taps


def block_linear(x: Tensor, wt: Tensor, p: str, name: str = "") -> Tensor:
    """nn.Linear of an EchoDiT block (attention wq / wk / wv / gate / wo, mlp w1 / w3 / w2); fp8 operands when C5 mode is on."""
    if _FP8_BLOCK_LINEARS and p.startswith("blocks."):
        st = _FP8_ACT_STATIC.get(f"{p}.{name}") if (_FP8_ACT_STATIC and name) else None
        xq = fake_quant_static_e4m3(x, st) if st is not None else fake_quant_rows_e4m3(x)
        y = (xq @ fake_quant_rows_e4m3(wt).t()).to(x.dtype)
    else:
        y = F.linear(x, wt)
    if _LINEAR_TAPS is not None and name:
        _LINEAR_TAPS[f"{p}.{name}.in"] = x.clone()
        _LINEAR_TAPS[f"{p}.{name}.out"] = y.clone()
    return y


def swiglu(w: Weights, p: str, x: Tensor) -> Tensor:
    """w2(silu(w1 x) * w3 x).  model.py:307-308."""
    return block_linear(F.silu(block_linear(x, w[f"{p}.w1.weight"], p, "w1")) * block_linear(x, w[f"{p}.w3.weight"], p, "w3"), w[f"{p}.w2.weight"], p, "w2")


def encoder_self_attention(w: Weights, p: str, x: Tensor, mask: Optional[Tensor], fc: Tensor,
                           heads: int, causal: bool, eps: float) -> Tensor:
    """SelfAttention.forward (gated, q/k head-norm, full RoPE).  model.py:128-161."""
    b, s = x.shape[:2]
    q = F.linear(x, w[f"{p}.wq.weight"]).reshape(b, s, heads, -1)
    k = F.linear(x, w[f"{p}.wk.weight"]).reshape(b, s, heads, -1)
    v = F.linear(x, w[f"{p}.wv.weight"]).reshape(b, s, heads, -1)
    g = F.linear(x, w[f"{p}.gate.weight"])
    q = rms_norm(q, w[f"{p}.q_norm.weight"], eps)
    k = rms_norm(k, w[f"{p}.k_norm.weight"], eps)
    q = rotate_pairs(q, fc[:s])
    k = rotate_pairs(k, fc[:s])
    am = mask[:, None, None] if mask is not None else None
    o = F.scaled_dot_product_attention(
        query=q.transpose(1, 2), key=k.transpose(1, 2), value=v.transpose(1, 2), attn_mask=am, is_causal=causal
    ).transpose(1, 2)
    o = o.reshape(b, s, -1) * torch.sigmoid(g)
    return block_linear(o, w[f"{p}.wo.weight"], p, "wo")


def encoder_block(w: Weights, p: str, x: Tensor, mask: Optional[Tensor], fc: Tensor,
                  heads: int, causal: bool, eps: float) -> Tensor:
    """EncoderTransformerBlock.forward.  model.py:335-339."""
    x = x + encoder_self_attention(w, f"{p}.attention", rms_norm(x, w[f"{p}.attention_norm.weight"], eps),
                                   mask, fc, heads, causal, eps)
    x = x + swiglu(w, f"{p}.mlp", rms_norm(x, w[f"{p}.mlp_norm.weight"], eps))
    return x


def text_encoder(w: Weights, cfg: DiTConfig, ids: Tensor, mask: Optional[Tensor]) -> Tensor:
    """TextEncoder.forward.  model.py:419-427."""
    x = F.embedding(ids, w["text_encoder.text_embedding.weight"])
    fc = rope_table(cfg.text_model_size // cfg.text_num_heads, ids.shape[1]).to(x.device)
    for i in range(cfg.text_num_layers):
        x = encoder_block(w, f"text_encoder.blocks.{i}", x, mask, fc, cfg.text_num_heads, False, cfg.norm_eps)
    return x


def patch_encoder(w: Weights, cfg: DiTConfig, prefix: str, latent: Tensor) -> Tensor:
    """SpeakerEncoder.forward (also the latent-prefix encoder).  model.py:458-469."""
    ps = cfg.speaker_patch_size
    x = latent.reshape(*latent.shape[:-2], latent.shape[-2] // ps, latent.shape[-1] * ps)
    x = F.linear(x, w[f"{prefix}.in_proj.weight"], w[f"{prefix}.in_proj.bias"])
    x = x / 6.0
    fc = rope_table(cfg.speaker_model_size // cfg.speaker_num_heads, x.shape[1]).to(x.device)
    for i in range(cfg.speaker_num_layers):
        x = encoder_block(w, f"{prefix}.blocks.{i}", x, None, fc, cfg.speaker_num_heads, True, cfg.norm_eps)
    return x


def kv_cache_text(w: Weights, cfg: DiTConfig, ids: Tensor, mask: Optional[Tensor]) -> KV:
    """EchoDiT.get_kv_cache_text.  model.py:606-613, 270-275."""
    s = rms_norm(text_encoder(w, cfg, ids, mask), w["text_norm.weight"], cfg.norm_eps)
    out = []
    for i in range(cfg.num_layers):
        p = f"blocks.{i}.attention"
        k = F.linear(s, w[f"{p}.wk_text.weight"]).reshape(s.shape[0], s.shape[1], cfg.num_heads, -1)
        v = F.linear(s, w[f"{p}.wv_text.weight"]).reshape(s.shape[0], s.shape[1], cfg.num_heads, -1)
        out.append((rms_norm(k, w[f"{p}.k_norm.weight"], cfg.norm_eps), v))
    return out


def kv_cache_speaker(w: Weights, cfg: DiTConfig, speaker_latent: Tensor) -> KV:
    """EchoDiT.get_kv_cache_speaker.  model.py:615-621, 277-282."""
    s = rms_norm(patch_encoder(w, cfg, "speaker_encoder", speaker_latent), w["speaker_norm.weight"], cfg.norm_eps)
    out = []
    for i in range(cfg.num_layers):
        p = f"blocks.{i}.attention"
        k = F.linear(s, w[f"{p}.wk_speaker.weight"]).reshape(s.shape[0], s.shape[1], cfg.num_heads, -1)
        v = F.linear(s, w[f"{p}.wv_speaker.weight"]).reshape(s.shape[0], s.shape[1], cfg.num_heads, -1)
        out.append((rms_norm(k, w[f"{p}.k_norm.weight"], cfg.norm_eps), v))
    return out


def kv_cache_latent(w: Weights, cfg: DiTConfig, prefix_latent: Tensor) -> KV:
    """EchoDiT.get_kv_cache_latent: keys get half-head RoPE at positions 4*i.  model.py:623-636, 284-293."""
    s = rms_norm(patch_encoder(w, cfg, "latent_encoder", prefix_latent), w["latent_norm.weight"], cfg.norm_eps)
    n = s.shape[1]
    fc = rope_table(cfg.head_dim, n * cfg.speaker_patch_size).to(s.device)
    fc = fc[torch.arange(n, device=s.device) * cfg.speaker_patch_size]
    out = []
    for i in range(cfg.num_layers):
        p = f"blocks.{i}.attention"
        k = F.linear(s, w[f"{p}.wk_latent.weight"]).reshape(s.shape[0], n, cfg.num_heads, -1)
        v = F.linear(s, w[f"{p}.wv_latent.weight"]).reshape(s.shape[0], n, cfg.num_heads, -1)
        k = rotate_first_half_of_heads(rms_norm(k, w[f"{p}.k_norm.weight"], cfg.norm_eps), fc)
        out.append((k, v))
    return out


def joint_attention(w: Weights, cfg: DiTConfig, p: str, x: Tensor, text_mask: Tensor, speaker_mask: Tensor,
                    fc: Tensor, kv_text: Tuple[Tensor, Tensor], kv_speaker: Tuple[Tensor, Tensor],
                    start_pos: int, kv_latent: Optional[Tuple[Tensor, Tensor]]) -> Tensor:
    """JointAttention.forward: keys = [self | latent | text | speaker].  model.py:204-268."""
    b, s = x.shape[:2]
    h = cfg.num_heads
    q = block_linear(x, w[f"{p}.wq.weight"], p, "wq").reshape(b, s, h, -1)
    k = block_linear(x, w[f"{p}.wk.weight"], p, "wk").reshape(b, s, h, -1)
    v = block_linear(x, w[f"{p}.wv.weight"], p, "wv").reshape(b, s, h, -1)
    q = rms_norm(q, w[f"{p}.q_norm.weight"], cfg.norm_eps)
    k = rms_norm(k, w[f"{p}.k_norm.weight"], cfg.norm_eps)
    g = block_linear(x, w[f"{p}.gate.weight"], p, "gate")
    fq = fc[start_pos:start_pos + s]
    q = rotate_first_half_of_heads(q, fq)
    k = rotate_first_half_of_heads(k, fq)
    kt, vt = kv_text
    ks, vs = kv_speaker
    if kv_latent is None or kv_latent[0].shape[1] == 0:
        kl = torch.zeros((b, 0, h, q.shape[-1]), device=x.device, dtype=x.dtype)
        vl = torch.zeros((b, 0, h, q.shape[-1]), device=x.device, dtype=x.dtype)
        lmask = torch.zeros((b, 0), dtype=torch.bool, device=x.device)
    else:
        kl, vl = kv_latent
        pos = torch.arange(kl.shape[1], device=x.device, dtype=torch.long) * cfg.speaker_patch_size
        lmask = (pos[None, :] < start_pos).expand(b, kl.shape[1])
    kk = torch.cat([k, kl, kt, ks], dim=1)
    vv = torch.cat([v, vl, vt, vs], dim=1)
    smask = torch.ones((b, s), dtype=torch.bool, device=x.device)
    m = torch.cat([smask, lmask, text_mask, speaker_mask], dim=1)[:, None, None]
    o = F.scaled_dot_product_attention(
        query=q.transpose(1, 2), key=kk.transpose(1, 2), value=vv.transpose(1, 2), attn_mask=m, is_causal=False
    ).transpose(1, 2)
    o = o.reshape(b, s, -1) * torch.sigmoid(g)
    return block_linear(o, w[f"{p}.wo.weight"], p, "wo")


def dit_forward(w: Weights, cfg: DiTConfig, x: Tensor, t: Tensor, text_mask: Tensor, speaker_mask: Tensor,
                kv_text: KV, kv_speaker: KV, start_pos: Optional[int] = None, kv_latent: Optional[KV] = None,
                taps: Optional[Dict[str, Tensor]] = None) -> Tensor:
    """EchoDiT.forward -> fp32 velocity.  model.py:563-604, 371-390."""
    sp = 0 if start_pos is None else start_pos
    fc = rope_table(cfg.head_dim, sp + x.shape[1]).to(x.device)
    speaker_mask = speaker_mask[..., :: cfg.speaker_patch_size]
    c = timestep_embedding(t, cfg.timestep_embed_size)
    c = F.linear(F.silu(F.linear(F.silu(F.linear(c, w["cond_module.0.weight"])), w["cond_module.2.weight"])),
                 w["cond_module.4.weight"])
    c = c[:, None]
    x = F.linear(x, w["in_proj.weight"], w["in_proj.bias"])
    if taps is not None:
        taps["cond"] = c.clone()
        taps["in_proj"] = x.clone()
    for i in range(cfg.num_layers):
        p = f"blocks.{i}"
        xn, ga = lowrank_adaln(w, f"{p}.attention_adaln", x, c, cfg.norm_eps)
        a = joint_attention(w, cfg, f"{p}.attention", xn, text_mask, speaker_mask, fc, kv_text[i], kv_speaker[i],
                            sp, kv_latent[i] if kv_latent is not None else None)
        x = x + ga * a
        xn2, gm = lowrank_adaln(w, f"{p}.mlp_adaln", x, c, cfg.norm_eps)
        x = x + gm * swiglu(w, f"{p}.mlp", xn2)
        if taps is not None:
            taps[f"block{i}.xn"] = xn.clone()
            taps[f"block{i}.attn"] = a.clone()
            taps[f"block{i}.out"] = x.clone()
    x = rms_norm(x, w["out_norm.weight"], cfg.norm_eps)
    x = F.linear(x, w["out_proj.weight"], w["out_proj.bias"])
    return x.float()


# ----------------------------------------------------------------------- samplers
def _cat3(c: KV) -> KV:
    """inference.py:398-406 with three copies of one cache."""
    return [(torch.cat([k, k, k], dim=0), torch.cat([v, v, v], dim=0)) for k, v in c]


def _scale_kv(c: KV, scale: float, max_layers: Optional[int]) -> None:
    """In-place K and V scaling of the first layers.  inference.py:408-414."""
    n = len(c) if max_layers is None else min(max_layers, len(c))
    for i in range(n):
        c[i][0].mul_(scale)
        c[i][1].mul_(scale)


def temporal_score_rescale(v: Tensor, x: Tensor, t, k: float, sigma: float) -> Tensor:
    """inference.py:416-424."""
    if t < 1:
        snr = (1 - t) ** 2 / (t ** 2)
        ratio = (snr * sigma ** 2 + 1) / (snr * sigma ** 2 / k + 1)
        return 1 / (1 - t) * (ratio * ((1 - t) * v + x) - x)
    return v


@torch.inference_mode()
def sample_euler(w: Weights, cfg: DiTConfig, dtype: torch.dtype, speaker_latent: Tensor, speaker_mask: Tensor,
                 text_input_ids: Tensor, text_mask: Tensor, rng_seed: int, num_steps: int, cfg_scale_text: float,
                 cfg_scale_speaker: float, cfg_min_t: float, cfg_max_t: float, truncation_factor: Optional[float],
                 rescale_k: Optional[float], rescale_sigma: Optional[float], speaker_kv_scale: Optional[float],
                 speaker_kv_max_layers: Optional[int], speaker_kv_min_t: Optional[float],
                 sequence_length: Optional[int] = None, x_init: Optional[Tensor] = None,
                 trace: Optional[List[Tensor]] = None, model_t_dtype: Optional[torch.dtype] = None) -> Tensor:
    """sample_euler_cfg_independent_guidances.  inference.py:427-517.

    ``x_init`` (already multiplied by nothing) overrides the RNG draw so that device-specific
    generators do not enter parity tests; truncation is still applied to it.

    ``model_t_dtype`` (tests only; None = the reference): the timestep handed to the MODEL is first rounded to that dtype, everything
    else stays as it is.  The reference's bf16 path rounds t to bf16 before the timestep embedding (inference.py:488-489,
    model.py:40: bf16(0.666) moves embedding phases by up to 2 rad), which alone puts its bf16 run ~1 RMS from its fp32 run;
    an fp32 run at the bf16-ROUNDED timesteps is the reference a bf16 implementation's own rounding noise can be measured against.
    """
    def model_t(n: int, t: Tensor) -> Tensor:
        tt = torch.ones((n,), device=dev) * t
        return (tt.to(model_t_dtype).to(dtype) if model_t_dtype is not None else tt.to(dtype))

    if sequence_length is None:
        sequence_length = 640
    dev = text_input_ids.device
    b = text_input_ids.shape[0]
    ts = torch.linspace(1.0, 0.0, num_steps + 1, device=dev) * 0.999
    kv_t = kv_cache_text(w, cfg, text_input_ids, text_mask)
    kv_s = kv_cache_speaker(w, cfg, speaker_latent.to(dtype))
    if speaker_kv_scale is not None:
        _scale_kv(kv_s, speaker_kv_scale, speaker_kv_max_layers)
    kv_t3, kv_s3 = _cat3(kv_t), _cat3(kv_s)
    tm3 = torch.cat([text_mask, torch.zeros_like(text_mask), text_mask], dim=0)
    sm3 = torch.cat([speaker_mask, speaker_mask, torch.zeros_like(speaker_mask)], dim=0)
    if x_init is None:
        rng = torch.Generator(device=dev).manual_seed(rng_seed)
        x = torch.randn((b, sequence_length, cfg.latent_size), device=dev, dtype=torch.float32, generator=rng)
    else:
        x = x_init.clone().float()
    if truncation_factor is not None:
        x = x * truncation_factor
    for i in range(num_steps):
        t, tn = ts[i], ts[i + 1]
        if ((t >= cfg_min_t) * (t <= cfg_max_t)).item():
            vc, vut, vus = dit_forward(
                w, cfg, torch.cat([x, x, x], dim=0).to(dtype), model_t(b * 3, t),
                tm3, sm3, kv_t3, kv_s3).float().chunk(3, dim=0)
            v = vc + cfg_scale_text * (vc - vut) + cfg_scale_speaker * (vc - vus)
        else:
            v = dit_forward(w, cfg, x.to(dtype), model_t(b, t),
                            text_mask, speaker_mask, kv_t, kv_s).float()
        if rescale_k is not None and rescale_sigma is not None:
            v = temporal_score_rescale(v, x, t, rescale_k, rescale_sigma)
        if speaker_kv_scale is not None and tn < speaker_kv_min_t and t >= speaker_kv_min_t:
            _scale_kv(kv_s, 1.0 / speaker_kv_scale, speaker_kv_max_layers)
            kv_s3 = _cat3(kv_s)
        x = x + v * (tn - t)
        if trace is not None:
            trace.append(x.clone())
    return x


@torch.inference_mode()
def sample_blockwise(w: Weights, cfg: DiTConfig, dtype: torch.dtype, speaker_latent: Tensor, speaker_mask: Tensor,
                     text_input_ids: Tensor, text_mask: Tensor, rng_seed: int, block_sizes: Sequence[int],
                     num_steps: int, cfg_scale_text: float, cfg_scale_speaker: float, cfg_min_t: float,
                     cfg_max_t: float, truncation_factor: Optional[float], rescale_k: Optional[float],
                     rescale_sigma: Optional[float], speaker_kv_scale: Optional[float],
                     speaker_kv_max_layers: Optional[int], speaker_kv_min_t: Optional[float],
                     continuation_latent: Optional[Tensor] = None,
                     x_inits: Optional[Sequence[Tensor]] = None) -> Tensor:
    """sample_blockwise_euler_cfg_independent_guidances.  inference_blockwise.py:14-123."""
    dev = text_input_ids.device
    b = text_input_ids.shape[0]
    rng = torch.Generator(device=dev).manual_seed(rng_seed)
    ts = torch.linspace(1.0, 0.0, num_steps + 1, device=dev) * 0.999
    kv_t = kv_cache_text(w, cfg, text_input_ids, text_mask)
    kv_s = kv_cache_speaker(w, cfg, speaker_latent.to(dtype))
    kv_t3, kv_s3 = _cat3(kv_t), _cat3(kv_s)
    tm3 = torch.cat([text_mask, torch.zeros_like(text_mask), text_mask], dim=0)
    sm3 = torch.cat([speaker_mask, speaker_mask, torch.zeros_like(speaker_mask)], dim=0)
    prefix = torch.zeros((b, sum(block_sizes), cfg.latent_size), device=dev, dtype=torch.float32)
    start = 0
    if continuation_latent is not None:
        start = continuation_latent.shape[1]
        prefix = torch.cat([continuation_latent, prefix], dim=1)
    for bi, bs in enumerate(block_sizes):
        if speaker_kv_scale is not None:
            _scale_kv(kv_s, speaker_kv_scale, speaker_kv_max_layers)
            kv_s3 = _cat3(kv_s)
        kv_l3 = kv_cache_latent(w, cfg, torch.cat([prefix, prefix, prefix], dim=0).to(dtype))
        kv_l = [(k[:b], v[:b]) for k, v in kv_l3]
        if x_inits is None:
            x = torch.randn((b, bs, cfg.latent_size), device=dev, dtype=torch.float32, generator=rng)
        else:
            x = x_inits[bi].clone().float()
        if truncation_factor is not None:
            x = x * truncation_factor
        for i in range(num_steps):
            t, tn = ts[i], ts[i + 1]
            if ((t >= cfg_min_t) * (t <= cfg_max_t)).item():
                vc, vut, vus = dit_forward(
                    w, cfg, torch.cat([x, x, x], dim=0).to(dtype), (torch.ones((b * 3,), device=dev) * t).to(dtype),
                    tm3, sm3, kv_t3, kv_s3, start_pos=start, kv_latent=kv_l3).float().chunk(3, dim=0)
                v = vc + cfg_scale_text * (vc - vut) + cfg_scale_speaker * (vc - vus)
            else:
                v = dit_forward(w, cfg, x.to(dtype), (torch.ones((b,), device=dev) * t).to(dtype),
                                text_mask, speaker_mask, kv_t, kv_s, start_pos=start, kv_latent=kv_l).float()
            if rescale_k is not None and rescale_sigma is not None:
                v = temporal_score_rescale(v, x, t, rescale_k, rescale_sigma)
            if speaker_kv_scale is not None and tn < speaker_kv_min_t and t >= speaker_kv_min_t:
                _scale_kv(kv_s, 1.0 / speaker_kv_scale, speaker_kv_max_layers)
                kv_s3 = _cat3(kv_s)
            x = x + v * (tn - t)
        prefix[:, start:start + bs] = x
        start += bs
    return prefix


# ------------------------------------------------------------------ Fish S1-DAC decode
def fold_weight_norm(w: Weights, p: str) -> Tensor:
    """w = g * v / ||v|| (norm over all dims but 0).  autoencoder.py:90-94,291-293; torch weight_norm, dim=0."""
    g = w[f"{p}.parametrizations.weight.original0"]
    v = w[f"{p}.parametrizations.weight.original1"]
    return torch._weight_norm(v, g, 0)


def _conv_weight(w: Weights, p: str) -> Tensor:
    return w[f"{p}.weight"] if f"{p}.weight" in w else fold_weight_norm(w, p)


def snake(x: Tensor, alpha: Tensor) -> Tensor:
    """x + (alpha+1e-9)^-1 * sin(alpha x)^2.  autoencoder.py:96-102."""
    sh = x.shape
    x = x.reshape(sh[0], sh[1], -1)
    x = x + (alpha + 1e-9).reciprocal() * torch.sin(alpha * x).pow(2)
    return x.reshape(sh)


def causal_conv1d(w: Weights, p: str, x: Tensor, k: int, dilation: int = 1, stride: int = 1, groups: int = 1) -> Tensor:
    """CausalConvNet.forward: left pad (k-1)d+1-stride, right pad to complete the last window.  autoencoder.py:264-289."""
    eff = (k - 1) * dilation + 1
    pad = eff - stride
    length = x.shape[-1]
    n_frames = (length - eff + pad) / stride + 1
    extra = (math.ceil(n_frames) - 1) * stride + (eff - pad) - length
    x = F.pad(x, (pad, extra), "constant", 0.0)
    return F.conv1d(x, _conv_weight(w, f"{p}.conv"), w[f"{p}.conv.bias"], stride=stride, dilation=dilation,
                    groups=groups).contiguous()


def causal_conv_transpose1d(w: Weights, p: str, x: Tensor, k: int, stride: int) -> Tensor:
    """CausalTransConvNet.forward: run unpadded, drop the last k-stride samples.  autoencoder.py:300-316."""
    y = F.conv_transpose1d(x, _conv_weight(w, f"{p}.conv"), w[f"{p}.conv.bias"], stride=stride)
    pad = k - stride
    return y[..., : y.shape[-1] - pad].contiguous()


def ae_rms_norm(x: Tensor, weight: Tensor, eps: float) -> Tensor:
    """AE RMSNorm: normalise in fp32, cast back, THEN multiply by weight.  autoencoder.py:726-731."""
    xf = x.float()
    y = (xf * torch.rsqrt(torch.mean(xf * xf, dim=-1, keepdim=True) + eps)).type_as(x)
    return y * weight


def ae_rope_cache(seq_len: int, n_elem: int, base: float = 10000.0, dtype: torch.dtype = torch.bfloat16) -> Tensor:
    """(seq, n_elem/2, 2) cos/sin cache, stored in bf16 by default.  autoencoder.py:805-813."""
    freqs = 1.0 / (base ** (torch.arange(0, n_elem, 2)[: n_elem // 2].float() / n_elem))
    ang = torch.outer(torch.arange(seq_len, device=freqs.device), freqs)
    cis = torch.polar(torch.ones_like(ang), ang)
    return torch.stack([cis.real, cis.imag], dim=-1).to(dtype=dtype)


def ae_rotate(x: Tensor, cache: Tensor) -> Tensor:
    """autoencoder.py:815-826.  x (b,s,h,d), cache (s,d/2,2)."""
    xs = x.float().reshape(*x.shape[:-1], -1, 2)
    c = cache.view(1, xs.size(1), 1, xs.size(3), 2)
    out = torch.stack([xs[..., 0] * c[..., 0] - xs[..., 1] * c[..., 1],
                       xs[..., 1] * c[..., 0] + xs[..., 0] * c[..., 1]], -1)
    return out.flatten(3).type_as(x)


def window_mask(n: int, window: int) -> Tensor:
    """key j visible to query i iff max(0, i-window+1) <= j <= i.  autoencoder.py:762-773."""
    i = torch.arange(n).view(-1, 1)
    j = torch.arange(n)
    return ((j >= (i - window + 1).clamp(min=0)) & (j <= i))[None, None]


def window_transformer(w: Weights, p: str, z: Tensor, layers: int, h: int, hd: int, window: int, block_size: int,
                       eps: float, rope_base: float) -> Tensor:
    """WindowLimitedTransformer.forward with input_dim == dim, channels-first in/out.  autoencoder.py:786-802, 590-626, 663-717."""
    x = z.transpose(1, 2)
    b, s, _ = x.shape
    if f"{p}.freqs_cis" in w:
        cache = w[f"{p}.freqs_cis"][:s]
    else:
        cache = ae_rope_cache(block_size, hd, rope_base)[:s]
    cache = cache.to(x.device)
    mask = window_mask(s, window).to(x.device)
    for i in range(layers):
        lp = f"{p}.layers.{i}"
        xn = ae_rms_norm(x, w[f"{lp}.attention_norm.weight"], eps)
        q, k, v = F.linear(xn, w[f"{lp}.attention.wqkv.weight"]).split([h * hd, h * hd, h * hd], dim=-1)
        q = ae_rotate(q.view(b, s, h, hd), cache).transpose(1, 2)
        k = ae_rotate(k.view(b, s, h, hd), cache).transpose(1, 2)
        v = v.view(b, s, h, hd).transpose(1, 2)
        y = F.scaled_dot_product_attention(q, k, v, dropout_p=0.0, attn_mask=mask)
        y = F.linear(y.transpose(1, 2).contiguous().view(b, s, h * hd), w[f"{lp}.attention.wo.weight"])
        hmid = x + y.mul_(w[f"{lp}.attention_layer_scale.gamma"])
        hn = ae_rms_norm(hmid, w[f"{lp}.ffn_norm.weight"], eps)
        f = F.linear(F.silu(F.linear(hn, w[f"{lp}.feed_forward.w1.weight"])) *
                     F.linear(hn, w[f"{lp}.feed_forward.w3.weight"]), w[f"{lp}.feed_forward.w2.weight"])
        x = hmid + f.mul_(w[f"{lp}.ffn_layer_scale.gamma"])
    x = ae_rms_norm(x, w[f"{p}.norm.weight"], eps)
    return x.transpose(1, 2)


def post_module(w: Weights, cfg: DacConfig, z: Tensor, p: str = "quantizer.post_module") -> Tensor:
    """quantizer.post_module / pre_module (same configuration).  autoencoder.py:1144-1161."""
    return window_transformer(w, p, z, cfg.post_layers, cfg.post_heads, cfg.post_head_dim, cfg.post_window, cfg.post_block_size,
                              cfg.norm_eps, cfg.rope_base)


def convnext_block(w: Weights, p: str, x: Tensor) -> Tensor:
    """ConvNeXtBlock.forward (dw conv k7, LN 1e-6, Linear x4, erf-GELU, Linear, gamma, residual).  autoencoder.py:360-373."""
    c = x.shape[1]
    y = causal_conv1d(w, f"{p}.dwconv", x, 7, groups=c).permute(0, 2, 1)
    y = F.layer_norm(y, (c,), w[f"{p}.norm.weight"], w[f"{p}.norm.bias"], 1e-6)
    y = F.linear(F.gelu(F.linear(y, w[f"{p}.pwconv1.weight"], w[f"{p}.pwconv1.bias"])),
                 w[f"{p}.pwconv2.weight"], w[f"{p}.pwconv2.bias"])
    y = w[f"{p}.gamma"] * y
    return x + y.permute(0, 2, 1)


def residual_unit(w: Weights, p: str, x: Tensor, dilation: int) -> Tensor:
    """ResidualUnit.forward (causal): snake, conv k7 dil, snake, conv k1, + x.  autoencoder.py:879-900."""
    y = snake(x, w[f"{p}.block.0.alpha"])
    y = causal_conv1d(w, f"{p}.block.1", y, 7, dilation=dilation)
    y = snake(y, w[f"{p}.block.2.alpha"])
    y = causal_conv1d(w, f"{p}.block.3", y, 1)
    return x + y


def dac_decoder(w: Weights, cfg: DacConfig, z: Tensor, p: str = "decoder.model") -> Tensor:
    """Decoder.forward; DecoderBlock never adds its transformer.  autoencoder.py:971-998, 959-968."""
    x = causal_conv1d(w, f"{p}.0", z, 7)
    n = len(cfg.decoder_rates)
    for i, r in enumerate(cfg.decoder_rates):
        bp = f"{p}.{i + 1}.block"
        x = snake(x, w[f"{bp}.0.alpha"])
        x = causal_conv_transpose1d(w, f"{bp}.1", x, 2 * r, r)
        for j, d in enumerate((1, 3, 9)):
            x = residual_unit(w, f"{bp}.{2 + j}", x, d)
    x = snake(x, w[f"{p}.{n + 1}.alpha"])
    x = causal_conv1d(w, f"{p}.{n + 2}", x, 7)
    return torch.tanh(x)


@torch.inference_mode()
def dac_decode_zq(w: Weights, cfg: DacConfig, z_q: Tensor, taps: Optional[Dict[str, Tensor]] = None) -> Tensor:
    """DAC.decode_zq.  autoencoder.py:1128-1132, 427-435."""
    z = post_module(w, cfg, z_q)
    if taps is not None:
        taps["post_module"] = z.clone()
    nf = len(cfg.upsample_factors)
    for i, f in enumerate(reversed(cfg.upsample_factors)):
        z = causal_conv_transpose1d(w, f"quantizer.upsample.{i}.0", z, f, f)
        z = convnext_block(w, f"quantizer.upsample.{i}.1", z)
    if taps is not None:
        taps["upsample"] = z.clone()
    return dac_decoder(w, cfg, z)


@dataclass
class PCA:
    """inference.py:86-90."""
    pca_components: Tensor
    pca_mean: Tensor
    latent_scale: float


@torch.inference_mode()
def ae_decode(w: Weights, cfg: DacConfig, pca: PCA, latent: Tensor, ae_dtype: torch.dtype = torch.float32) -> Tensor:
    """inference.py:226-229."""
    z = (latent / pca.latent_scale) @ pca.pca_components + pca.pca_mean
    return dac_decode_zq(w, cfg, z.transpose(1, 2).to(ae_dtype)).float()


# ------------------------------------------------------------ host-side post-processing
# ------------------------------------------------------------------- DAC encode (speaker reference -> latents)
def dac_encoder(w: Weights, cfg: DacConfig, audio: Tensor, p: str = "encoder.block") -> Tensor:
    """Encoder.forward, causal: conv k7 1->C; per block 3 ResidualUnits (dil 1, 3, 9), Snake, conv k = 2s stride s
    (C/2 -> C), optional window-limited transformer (heads = C / 64, ffn = 3C); Snake; conv k3.  autoencoder.py:839-929."""
    x = causal_conv1d(w, f"{p}.0", audio, 7)
    ch = cfg.encoder_dim
    n = len(cfg.encoder_rates)
    for i, (r, nt) in enumerate(zip(cfg.encoder_rates, cfg.encoder_transformer_layers)):
        ch *= 2
        bp = f"{p}.{i + 1}.block"
        for j, d in enumerate((1, 3, 9)):
            x = residual_unit(w, f"{bp}.{j}", x, d)
        x = snake(x, w[f"{bp}.3.alpha"])
        x = causal_conv1d(w, f"{bp}.4", x, 2 * r, stride=r)
        if nt > 0:
            x = window_transformer(w, f"{bp}.5", x, nt, ch // 64, 64, cfg.encoder_window, cfg.encoder_block_size, cfg.norm_eps,
                                   cfg.rope_base)
    x = snake(x, w[f"{p}.{n + 1}.alpha"])
    return causal_conv1d(w, f"{p}.{n + 2}", x, 3)


def _wn_conv1x1(w: Weights, p: str, x: Tensor) -> Tensor:
    """WNConv1d(kernel_size=1) on (B, C, T).  autoencoder.py:90-94."""
    return F.conv1d(x, fold_weight_norm(w, p), w[f"{p}.bias"])


def vq_forward(w: Weights, p: str, z: Tensor, gaps: Optional[List[Tensor]] = None) -> Tuple[Tensor, Tensor]:
    """VectorQuantize.forward (eval): in_proj, nearest L2-normalised code, straight-through value, out_proj.
    Returns (z_q (B, D, T), indices (B, T)).  autoencoder.py:130-158.  `gaps` (tests): receives per frame the distance margin
    between the chosen code and the runner-up, i.e. how close the argmax is to a tie."""
    z_e = _wn_conv1x1(w, f"{p}.in_proj", z)
    b, d, t = z_e.shape
    enc = F.normalize(z_e.permute(0, 2, 1).reshape(b * t, d))
    cb = F.normalize(w[f"{p}.codebook.weight"])
    dist = enc.pow(2).sum(1, keepdim=True) - 2 * enc @ cb.t() + cb.pow(2).sum(1, keepdim=True).t()
    idx = (-dist).max(1)[1].view(b, t)
    if gaps is not None:
        top2 = (-dist).topk(2, dim=1)[0]
        gaps.append((top2[:, 0] - top2[:, 1]).view(b, t))
    z_q = F.embedding(idx, w[f"{p}.codebook.weight"]).transpose(1, 2)
    z_q = z_e + (z_q - z_e)
    return _wn_conv1x1(w, f"{p}.out_proj", z_q), idx


def vq_from_code(w: Weights, p: str, idx: Tensor) -> Tensor:
    """out_proj(codebook[idx]) for one quantizer.  autoencoder.py:142-143, 223-231."""
    return _wn_conv1x1(w, f"{p}.out_proj", F.embedding(idx, w[f"{p}.codebook.weight"]).transpose(1, 2))


@torch.inference_mode()
def dac_encode_codes(w: Weights, cfg: DacConfig, audio: Tensor, taps: Optional[Dict[str, Tensor]] = None) -> Tensor:
    """DAC.encode: pad to whole frames, Encoder, downsample (conv k = f stride f + ConvNeXt per factor), pre_module,
    semantic VQ, residual VQ stack.  Returns codes (B, 1 + n_codebooks, T).  autoencoder.py:1080-1108, 451-464, 184-220."""
    if audio.ndim == 2:
        audio = audio.unsqueeze(1)
    frame = cfg.encoder_hop
    for f in cfg.upsample_factors:
        frame *= f
    length = audio.shape[-1]
    audio = F.pad(audio, (0, math.ceil(length / frame) * frame - length))
    z = dac_encoder(w, cfg, audio)
    if taps is not None:
        taps["encoder"] = z.clone()
    for i, f in enumerate(cfg.upsample_factors):
        z = causal_conv1d(w, f"quantizer.downsample.{i}.0", z, f, stride=f)
        z = convnext_block(w, f"quantizer.downsample.{i}.1", z)
    z = post_module(w, cfg, z, "quantizer.pre_module")
    if taps is not None:
        taps["pre_module"] = z.clone()
    gaps: Optional[List[Tensor]] = [] if taps is not None else None
    sem_z, sem_idx = vq_forward(w, "quantizer.semantic_quantizer.quantizers.0", z, gaps)
    residual = z - sem_z
    codes = [sem_idx]
    for i in range(cfg.n_codebooks):
        zq_i, idx_i = vq_forward(w, f"quantizer.quantizer.quantizers.{i}", residual, gaps)
        residual = residual - zq_i
        codes.append(idx_i)
    if taps is not None:
        taps["vq_gap"] = torch.stack(gaps, dim=1)          # (B, 1 + n_codebooks, T): argmax margin of every stage
    return torch.stack(codes, dim=1)


@torch.inference_mode()
def dac_encode_zq(w: Weights, cfg: DacConfig, audio: Tensor, taps: Optional[Dict[str, Tensor]] = None) -> Tensor:
    """DAC.encode_zq: codes -> sum of out_proj(codebook[code]) over the semantic and residual quantizers.  autoencoder.py:1117-1126."""
    codes = dac_encode_codes(w, cfg, audio, taps)
    if taps is not None:
        taps["codes"] = codes.clone()
    z_sem = 0.0 + vq_from_code(w, "quantizer.semantic_quantizer.quantizers.0", codes[:, 0].clamp(max=cfg.semantic_codebook_size - 1))
    z_res = 0.0
    for i in range(cfg.n_codebooks):
        z_res = z_res + vq_from_code(w, f"quantizer.quantizer.quantizers.{i}", codes[:, 1 + i].clamp(max=cfg.codebook_size - 1))
    return z_sem + z_res


@torch.inference_mode()
def ae_encode(w: Weights, cfg: DacConfig, pca: PCA, audio: Tensor) -> Tensor:
    """inference.py:218-224: (B, 1, L) audio -> (B, T, 80) latents."""
    z_q = dac_encode_zq(w, cfg, audio).float()
    z_q = (z_q.transpose(1, 2) - pca.pca_mean) @ pca.pca_components.T
    return z_q * pca.latent_scale


@torch.inference_mode()
def get_speaker_latent_and_mask(w: Weights, cfg: DacConfig, pca: PCA, audio: Tensor, max_speaker_latent_length: int = 6400,
                                audio_chunk_size: int = 640 * 2048, pad_to_max: bool = False,
                                divis_by_patch_size: Optional[int] = 4) -> Tuple[Tensor, Tensor]:
    """inference.py:239-283: encode in 30 s chunks (each zero padded to a whole chunk), trim to the true length."""
    factor = 2048
    audio = audio[:, : max_speaker_latent_length * factor]
    lat = []
    for i in range(0, audio.shape[1], audio_chunk_size):
        chunk = audio[:, i:i + audio_chunk_size]
        if chunk.shape[1] < audio_chunk_size:
            chunk = F.pad(chunk, (0, audio_chunk_size - chunk.shape[1]))
        lat.append(ae_encode(w, cfg, pca, chunk.unsqueeze(0)))
    speaker_latent = torch.cat(lat, dim=1)
    actual = audio.shape[1] // factor
    mask = (torch.arange(speaker_latent.shape[1]) < actual).unsqueeze(0)
    if pad_to_max and speaker_latent.shape[1] < max_speaker_latent_length:
        speaker_latent = F.pad(speaker_latent, (0, 0, 0, max_speaker_latent_length - speaker_latent.shape[1]))
        mask = F.pad(mask, (0, max_speaker_latent_length - mask.shape[1]))
    elif not pad_to_max:
        speaker_latent, mask = speaker_latent[:, :actual], mask[:, :actual]
    if divis_by_patch_size is not None:
        n = speaker_latent.shape[1] // divis_by_patch_size * divis_by_patch_size
        speaker_latent, mask = speaker_latent[:, :n], mask[:, :n]
    return speaker_latent, mask


def find_flattening_point(data: Tensor, target_value: float = 0.0, window_size: int = 20,
                          std_threshold: float = 0.05) -> int:
    """inference.py:288-296."""
    padded = torch.cat([data, torch.zeros(window_size, *data.shape[1:], device=data.device, dtype=data.dtype)])
    for i in range(len(padded) - window_size):
        win = padded[i:i + window_size]
        if win.std() < std_threshold and abs(win.mean() - target_value) < 0.1:
            return i
    return len(data)


# ------------------------------------------------------------------ seeded weight recipes
def make_dit_weights(cfg: DiTConfig, seed: int = 0, with_blockwise: bool = True) -> Weights:
    """Seeded random EchoDiT weights (SURVEY.md §8c recipe, bf16-representable fp32).

    Independent of the reference classes so it can be regenerated on the GPU box; the parameter
    names/shapes follow SURVEY.md §A.5.  ndim>=2 and no 'norm' in the name -> N(0, 0.02);
    'norm' in the name -> 1; biases -> small N(0, 0.02) (so that bias paths are exercised).
    """
    g = torch.Generator().manual_seed(seed)
    w: Weights = {}

    def mat(name: str, *shape: int) -> None:
        w[name] = (torch.randn(shape, generator=g) * 0.02).bfloat16().float()

    def ones(name: str, *shape: int) -> None:
        w[name] = (1.0 + 0.1 * torch.randn(shape, generator=g)).bfloat16().float()

    def enc(prefix: str, d: int, heads: int, f: int, layers: int) -> None:
        for i in range(layers):
            p = f"{prefix}.blocks.{i}"
            for n in ("wq", "wk", "wv", "wo", "gate"):
                mat(f"{p}.attention.{n}.weight", d, d)
            ones(f"{p}.attention.q_norm.weight", heads, d // heads)
            ones(f"{p}.attention.k_norm.weight", heads, d // heads)
            mat(f"{p}.mlp.w1.weight", f, d)
            mat(f"{p}.mlp.w3.weight", f, d)
            mat(f"{p}.mlp.w2.weight", d, f)
            ones(f"{p}.attention_norm.weight", d)
            ones(f"{p}.mlp_norm.weight", d)

    mat("text_encoder.text_embedding.weight", cfg.text_vocab_size, cfg.text_model_size)
    w["text_encoder.text_embedding.weight"] *= 50.0  # embeddings ~N(0,1) like nn.Embedding
    w["text_encoder.text_embedding.weight"] = w["text_encoder.text_embedding.weight"].bfloat16().float()
    enc("text_encoder", cfg.text_model_size, cfg.text_num_heads, cfg.text_intermediate_size, cfg.text_num_layers)
    encs = ["speaker_encoder"] + (["latent_encoder"] if with_blockwise else [])
    for e in encs:
        mat(f"{e}.in_proj.weight", cfg.speaker_model_size, cfg.latent_size * cfg.speaker_patch_size)
        w[f"{e}.in_proj.weight"] = (w[f"{e}.in_proj.weight"] * 10).bfloat16().float()
        mat(f"{e}.in_proj.bias", cfg.speaker_model_size)
        enc(e, cfg.speaker_model_size, cfg.speaker_num_heads, cfg.speaker_intermediate_size, cfg.speaker_num_layers)
    ones("text_norm.weight", cfg.text_model_size)
    ones("speaker_norm.weight", cfg.speaker_model_size)
    if with_blockwise:
        ones("latent_norm.weight", cfg.speaker_model_size)
    d = cfg.model_size
    mat("cond_module.0.weight", d, cfg.timestep_embed_size)
    mat("cond_module.2.weight", d, d)
    mat("cond_module.4.weight", 3 * d, d)
    for k in ("cond_module.0.weight", "cond_module.2.weight", "cond_module.4.weight"):
        w[k] = (w[k] * 2.5).bfloat16().float()
    mat("in_proj.weight", d, cfg.latent_size)
    w["in_proj.weight"] = (w["in_proj.weight"] * 5).bfloat16().float()
    mat("in_proj.bias", d)
    for i in range(cfg.num_layers):
        p = f"blocks.{i}"
        for n in ("wq", "wk", "wv", "gate", "wo"):
            mat(f"{p}.attention.{n}.weight", d, d)
        srcs = ["text", "speaker"] + (["latent"] if with_blockwise else [])
        for s in srcs:
            sd = cfg.text_model_size if s == "text" else cfg.speaker_model_size
            mat(f"{p}.attention.wk_{s}.weight", d, sd)
            mat(f"{p}.attention.wv_{s}.weight", d, sd)
        ones(f"{p}.attention.q_norm.weight", cfg.num_heads, cfg.head_dim)
        ones(f"{p}.attention.k_norm.weight", cfg.num_heads, cfg.head_dim)
        mat(f"{p}.mlp.w1.weight", cfg.intermediate_size, d)
        mat(f"{p}.mlp.w3.weight", cfg.intermediate_size, d)
        mat(f"{p}.mlp.w2.weight", d, cfg.intermediate_size)
        for a in ("attention_adaln", "mlp_adaln"):
            for n in ("shift", "scale", "gate"):
                mat(f"{p}.{a}.{n}_down.weight", cfg.adaln_rank, d)
                mat(f"{p}.{a}.{n}_up.weight", d, cfg.adaln_rank)
                mat(f"{p}.{a}.{n}_up.bias", d)
    ones("out_norm.weight", d)
    mat("out_proj.weight", cfg.latent_size, d)
    mat("out_proj.bias", cfg.latent_size)
    return w


def make_dac_weights(cfg: DacConfig, seed: int = 0) -> Weights:
    """Seeded random decode-path DAC weights with the reference's key names (weight-norm kept unfolded).

    Gains are chosen so that the random network is well conditioned (waveform RMS ~1e-2, like the
    reference's own init): unit-variance fan-in init, conv gain 0.5 through the
    weight-norm g, small layer scales.  With gain 1 the tanh saturates and the net amplifies fp32
    rounding noise to ~5e-5, which would make a 1e-4 waveform tolerance meaningless.
    """
    g = torch.Generator().manual_seed(seed)
    w: Weights = {}

    def rnd(*shape: int, std: float = 0.02) -> Tensor:
        return torch.randn(shape, generator=g) * std

    def lin(co: int, ci: int) -> Tensor:
        return rnd(co, ci, std=1.0 / math.sqrt(ci))

    def wn_conv(p: str, co: int, ci: int, k: int, transpose: bool = False) -> None:
        shape = (ci, co, k) if transpose else (co, ci, k)
        v = rnd(*shape, std=1.0 / math.sqrt(ci * k))
        w[f"{p}.conv.parametrizations.weight.original1"] = v
        gshape = (shape[0], 1, 1)
        w[f"{p}.conv.parametrizations.weight.original0"] = 0.5 * v.flatten(1).norm(dim=1).view(gshape) * \
            (1.0 + 0.05 * torch.randn(gshape, generator=g))
        w[f"{p}.conv.bias"] = rnd(co)

    d, hd, nh = cfg.latent_dim, cfg.post_head_dim, cfg.post_heads
    pm = "quantizer.post_module"
    for i in range(cfg.post_layers):
        lp = f"{pm}.layers.{i}"
        w[f"{lp}.attention.wqkv.weight"] = lin(3 * nh * hd, d)
        w[f"{lp}.attention.wo.weight"] = lin(d, nh * hd)
        w[f"{lp}.feed_forward.w1.weight"] = lin(cfg.post_ffn, d)
        w[f"{lp}.feed_forward.w3.weight"] = lin(cfg.post_ffn, d)
        w[f"{lp}.feed_forward.w2.weight"] = lin(d, cfg.post_ffn)
        w[f"{lp}.ffn_norm.weight"] = 1.0 + rnd(d, std=0.1)
        w[f"{lp}.attention_norm.weight"] = 1.0 + rnd(d, std=0.1)
        w[f"{lp}.attention_layer_scale.gamma"] = 0.2 + rnd(d, std=0.05)
        w[f"{lp}.ffn_layer_scale.gamma"] = 0.2 + rnd(d, std=0.05)
    w[f"{pm}.norm.weight"] = 1.0 + rnd(d, std=0.1)
    for i in range(len(cfg.upsample_factors)):
        f = list(reversed(cfg.upsample_factors))[i]
        up = f"quantizer.upsample.{i}"
        w[f"{up}.0.conv.weight"] = rnd(d, d, f, std=1.0 / math.sqrt(d))
        w[f"{up}.0.conv.bias"] = rnd(d)
        w[f"{up}.1.dwconv.conv.weight"] = rnd(d, 1, 7, std=0.3)
        w[f"{up}.1.dwconv.conv.bias"] = rnd(d)
        w[f"{up}.1.norm.weight"] = 1.0 + rnd(d, std=0.1)
        w[f"{up}.1.norm.bias"] = rnd(d)
        w[f"{up}.1.pwconv1.weight"] = lin(4 * d, d)
        w[f"{up}.1.pwconv1.bias"] = rnd(4 * d)
        w[f"{up}.1.pwconv2.weight"] = lin(d, 4 * d)
        w[f"{up}.1.pwconv2.bias"] = rnd(d)
        w[f"{up}.1.gamma"] = 0.3 + rnd(d, std=0.05)
    dm = "decoder.model"
    ch = cfg.decoder_dim
    wn_conv(f"{dm}.0", ch, d, 7)
    for i, r in enumerate(cfg.decoder_rates):
        ci, co = ch // 2 ** i, ch // 2 ** (i + 1)
        bp = f"{dm}.{i + 1}.block"
        w[f"{bp}.0.alpha"] = 1.0 + rnd(1, ci, 1, std=0.2)
        wn_conv(f"{bp}.1", co, ci, 2 * r, transpose=True)
        for j in range(3):
            rp = f"{bp}.{2 + j}.block"
            w[f"{rp}.0.alpha"] = 1.0 + rnd(1, co, 1, std=0.2)
            wn_conv(f"{rp}.1", co, co, 7)
            w[f"{rp}.2.alpha"] = 1.0 + rnd(1, co, 1, std=0.2)
            wn_conv(f"{rp}.3", co, co, 1)
    n = len(cfg.decoder_rates)
    co = ch // 2 ** n
    w[f"{dm}.{n + 1}.alpha"] = 1.0 + rnd(1, co, 1, std=0.2)
    wn_conv(f"{dm}.{n + 2}", 1, co, 7)
    return w


def make_dac_encoder_weights(cfg: DacConfig, seed: int = 0) -> Weights:
    """Seeded random encode-path DAC weights (Encoder, quantizer.downsample, pre_module, the VQ stacks) with the reference's
    key names.  Drawn from their own generator so that the decode-path weights of make_dac_weights (and every fixture made
    from them) stay what they were.  Unit-variance fan-in init, weight-norm gains around 0.7, codebooks of unit-scale rows."""
    g = torch.Generator().manual_seed(seed + 4242)
    w: Weights = {}

    def rnd(*shape: int, std: float = 0.02) -> Tensor:
        return torch.randn(shape, generator=g) * std

    def lin(co: int, ci: int) -> Tensor:
        return rnd(co, ci, std=1.0 / math.sqrt(ci))

    def wn_conv(p: str, co: int, ci: int, k: int, gain: float = 0.7, sub: str = ".conv") -> None:
        v = rnd(co, ci, k, std=1.0 / math.sqrt(ci * k))
        w[f"{p}{sub}.parametrizations.weight.original1"] = v
        w[f"{p}{sub}.parametrizations.weight.original0"] = gain * v.flatten(1).norm(dim=1).view(co, 1, 1) * \
            (1.0 + 0.05 * torch.randn((co, 1, 1), generator=g))
        w[f"{p}{sub}.bias"] = rnd(co)

    def transformer(p: str, layers: int, d: int, nh: int, hd: int, ffn: int) -> None:
        for i in range(layers):
            lp = f"{p}.layers.{i}"
            w[f"{lp}.attention.wqkv.weight"] = lin(3 * nh * hd, d)
            w[f"{lp}.attention.wo.weight"] = lin(d, nh * hd)
            w[f"{lp}.feed_forward.w1.weight"] = lin(ffn, d)
            w[f"{lp}.feed_forward.w3.weight"] = lin(ffn, d)
            w[f"{lp}.feed_forward.w2.weight"] = lin(d, ffn)
            w[f"{lp}.ffn_norm.weight"] = 1.0 + rnd(d, std=0.1)
            w[f"{lp}.attention_norm.weight"] = 1.0 + rnd(d, std=0.1)
            w[f"{lp}.attention_layer_scale.gamma"] = 0.2 + rnd(d, std=0.05)
            w[f"{lp}.ffn_layer_scale.gamma"] = 0.2 + rnd(d, std=0.05)
        w[f"{p}.norm.weight"] = 1.0 + rnd(d, std=0.1)

    ep = "encoder.block"
    ch = cfg.encoder_dim
    wn_conv(f"{ep}.0", ch, 1, 7, gain=1.5)
    n = len(cfg.encoder_rates)
    for i, (r, nt) in enumerate(zip(cfg.encoder_rates, cfg.encoder_transformer_layers)):
        ci, ch = ch, ch * 2
        bp = f"{ep}.{i + 1}.block"
        for j in range(3):
            rp = f"{bp}.{j}.block"
            w[f"{rp}.0.alpha"] = 1.0 + rnd(1, ci, 1, std=0.2)
            wn_conv(f"{rp}.1", ci, ci, 7)
            w[f"{rp}.2.alpha"] = 1.0 + rnd(1, ci, 1, std=0.2)
            wn_conv(f"{rp}.3", ci, ci, 1)
        w[f"{bp}.3.alpha"] = 1.0 + rnd(1, ci, 1, std=0.2)
        wn_conv(f"{bp}.4", ch, ci, 2 * r, gain=1.0)
        if nt > 0:
            transformer(f"{bp}.5", nt, ch, ch // 64, 64, 3 * ch)
    w[f"{ep}.{n + 1}.alpha"] = 1.0 + rnd(1, ch, 1, std=0.2)
    wn_conv(f"{ep}.{n + 2}", cfg.latent_dim, ch, 3, gain=1.0)
    d = cfg.latent_dim
    for i, f in enumerate(cfg.upsample_factors):
        dn = f"quantizer.downsample.{i}"
        w[f"{dn}.0.conv.weight"] = rnd(d, d, f, std=1.0 / math.sqrt(d * f))
        w[f"{dn}.0.conv.bias"] = rnd(d)
        w[f"{dn}.1.dwconv.conv.weight"] = rnd(d, 1, 7, std=0.3)
        w[f"{dn}.1.dwconv.conv.bias"] = rnd(d)
        w[f"{dn}.1.norm.weight"] = 1.0 + rnd(d, std=0.1)
        w[f"{dn}.1.norm.bias"] = rnd(d)
        w[f"{dn}.1.pwconv1.weight"] = lin(4 * d, d)
        w[f"{dn}.1.pwconv1.bias"] = rnd(4 * d)
        w[f"{dn}.1.pwconv2.weight"] = lin(d, 4 * d)
        w[f"{dn}.1.pwconv2.bias"] = rnd(d)
        w[f"{dn}.1.gamma"] = 0.3 + rnd(d, std=0.05)
    transformer("quantizer.pre_module", cfg.post_layers, d, cfg.post_heads, cfg.post_head_dim, cfg.post_ffn)

    def vq(p: str, size: int) -> None:
        wn_conv(p, cfg.codebook_dim, d, 1, gain=1.0, sub=".in_proj")
        wn_conv(p, d, cfg.codebook_dim, 1, gain=0.5, sub=".out_proj")
        w[f"{p}.codebook.weight"] = rnd(size, cfg.codebook_dim, std=1.0)

    vq("quantizer.semantic_quantizer.quantizers.0", cfg.semantic_codebook_size)
    for i in range(cfg.n_codebooks):
        vq(f"quantizer.quantizer.quantizers.{i}", cfg.codebook_size)
    return w


def make_test_audio(n_samples: int, seed: int = 11) -> Tensor:
    """(1, 1, n) seeded test signal: a few gliding sinusoids + noise, amplitude ~0.2 (stands in for a speaker reference)."""
    g = torch.Generator().manual_seed(seed)
    t = torch.arange(n_samples, dtype=torch.float64) / 44100.0
    x = torch.zeros(n_samples, dtype=torch.float64)
    for f0, a in ((110.0, 0.12), (330.0, 0.06), (1250.0, 0.03)):
        x += a * torch.sin(2 * math.pi * (f0 * t + 40.0 * t * t))
    x = x.float() + 0.02 * torch.randn(n_samples, generator=g)
    return x.view(1, 1, n_samples)


def make_pca(cfg: DacConfig, latent_size: int = 80, seed: int = 0) -> PCA:
    """Synthetic PCAState: orthonormal components, small mean, scale 1."""
    g = torch.Generator().manual_seed(seed + 77)
    q, _ = torch.linalg.qr(torch.randn(cfg.latent_dim, latent_size, generator=g))
    return PCA(pca_components=q.T.contiguous(), pca_mean=0.1 * torch.randn(cfg.latent_dim, generator=g),
               latent_scale=1.0)
