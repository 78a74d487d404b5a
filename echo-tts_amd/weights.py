"""Checkpoint layout of the hot path (names and shapes of the reference state dicts)
and seeded synthetic checkpoints of that layout, generated directly on the device.

The trained Echo-TTS / Fish S1-DAC checkpoints are gated and not available offline, so benchmarks
and smoke tests run on random weights of the exact architecture.
"""
from __future__ import annotations

import math
from typing import Dict, List, Tuple

import torch

from .autoencoder import DACConfig
from .model import EchoDiTConfig

Shape = Tuple[int, ...]


def dit_param_shapes(cfg: EchoDiTConfig, with_blockwise: bool = True) -> List[Tuple[str, Shape]]:
    out: List[Tuple[str, Shape]] = []

    def enc(prefix: str, d: int, heads: int, f: int, layers: int) -> None:
        for i in range(layers):
            p = f"{prefix}.blocks.{i}"
            for n in ("wq", "wk", "wv", "wo", "gate"):
                out.append((f"{p}.attention.{n}.weight", (d, d)))
            out.append((f"{p}.attention.q_norm.weight", (heads, d // heads)))
            out.append((f"{p}.attention.k_norm.weight", (heads, d // heads)))
            out.append((f"{p}.mlp.w1.weight", (f, d)))
            out.append((f"{p}.mlp.w3.weight", (f, d)))
            out.append((f"{p}.mlp.w2.weight", (d, f)))
            out.append((f"{p}.attention_norm.weight", (d,)))
            out.append((f"{p}.mlp_norm.weight", (d,)))

    out.append(("text_encoder.text_embedding.weight", (cfg.text_vocab_size, cfg.text_model_size)))
    enc("text_encoder", cfg.text_model_size, cfg.text_num_heads, cfg.text_intermediate_size, cfg.text_num_layers)
    for e in ["speaker_encoder"] + (["latent_encoder"] if with_blockwise else []):
        out.append((f"{e}.in_proj.weight", (cfg.speaker_model_size, cfg.latent_size * cfg.speaker_patch_size)))
        out.append((f"{e}.in_proj.bias", (cfg.speaker_model_size,)))
        enc(e, cfg.speaker_model_size, cfg.speaker_num_heads, cfg.speaker_intermediate_size, cfg.speaker_num_layers)
    out.append(("text_norm.weight", (cfg.text_model_size,)))
    out.append(("speaker_norm.weight", (cfg.speaker_model_size,)))
    if with_blockwise:
        out.append(("latent_norm.weight", (cfg.speaker_model_size,)))
    d = cfg.model_size
    out += [("cond_module.0.weight", (d, cfg.timestep_embed_size)), ("cond_module.2.weight", (d, d)),
            ("cond_module.4.weight", (3 * d, d)), ("in_proj.weight", (d, cfg.latent_size)), ("in_proj.bias", (d,))]
    hd = d // cfg.num_heads
    for i in range(cfg.num_layers):
        p = f"blocks.{i}"
        for n in ("wq", "wk", "wv", "gate", "wo"):
            out.append((f"{p}.attention.{n}.weight", (d, d)))
        for s in ["text", "speaker"] + (["latent"] if with_blockwise else []):
            sd = cfg.text_model_size if s == "text" else cfg.speaker_model_size
            out.append((f"{p}.attention.wk_{s}.weight", (d, sd)))
            out.append((f"{p}.attention.wv_{s}.weight", (d, sd)))
        out.append((f"{p}.attention.q_norm.weight", (cfg.num_heads, hd)))
        out.append((f"{p}.attention.k_norm.weight", (cfg.num_heads, hd)))
        out.append((f"{p}.mlp.w1.weight", (cfg.intermediate_size, d)))
        out.append((f"{p}.mlp.w3.weight", (cfg.intermediate_size, d)))
        out.append((f"{p}.mlp.w2.weight", (d, cfg.intermediate_size)))
        for a in ("attention_adaln", "mlp_adaln"):
            for n in ("shift", "scale", "gate"):
                out.append((f"{p}.{a}.{n}_down.weight", (cfg.adaln_rank, d)))
                out.append((f"{p}.{a}.{n}_up.weight", (d, cfg.adaln_rank)))
                out.append((f"{p}.{a}.{n}_up.bias", (d,)))
    out += [("out_norm.weight", (d,)), ("out_proj.weight", (cfg.latent_size, d)), ("out_proj.bias", (cfg.latent_size,))]
    return out


def random_dit_state(cfg: EchoDiTConfig, device, dtype=torch.bfloat16, seed: int = 0, with_blockwise: bool = False) -> Dict[str, torch.Tensor]:
    """Random EchoDiT checkpoint on `device`: matrices ~ N(0, 1/fan_in) scaled to keep activations O(1), norms ~ 1."""
    g = torch.Generator(device=device).manual_seed(seed)
    sd = {}
    for name, shape in dit_param_shapes(cfg, with_blockwise):
        if "norm" in name:
            t = 1.0 + 0.1 * torch.randn(shape, device=device, generator=g)
        elif len(shape) == 1:
            t = 0.02 * torch.randn(shape, device=device, generator=g)
        elif "embedding" in name:
            t = torch.randn(shape, device=device, generator=g)
        else:
            t = torch.randn(shape, device=device, generator=g) / math.sqrt(shape[-1])
        sd[name] = t.to(dtype)
    return sd


def dac_param_shapes(cfg: DACConfig) -> List[Tuple[str, Shape]]:
    """Decode-path entries of the Fish S1-DAC state dict (weight-norm convs as original0 = g, original1 = v)."""
    out: List[Tuple[str, Shape]] = []
    d, qk = cfg.latent_dim, cfg.post_heads * cfg.post_head_dim
    pm = "quantizer.post_module"
    for i in range(cfg.post_layers):
        lp = f"{pm}.layers.{i}"
        out += [(f"{lp}.attention.wqkv.weight", (3 * qk, d)), (f"{lp}.attention.wo.weight", (d, qk)),
                (f"{lp}.feed_forward.w1.weight", (cfg.post_ffn, d)), (f"{lp}.feed_forward.w3.weight", (cfg.post_ffn, d)),
                (f"{lp}.feed_forward.w2.weight", (d, cfg.post_ffn)), (f"{lp}.ffn_norm.weight", (d,)),
                (f"{lp}.attention_norm.weight", (d,)), (f"{lp}.attention_layer_scale.gamma", (d,)),
                (f"{lp}.ffn_layer_scale.gamma", (d,))]
    out.append((f"{pm}.norm.weight", (d,)))
    for i, f in enumerate(reversed(cfg.upsample_factors)):
        up = f"quantizer.upsample.{i}"
        out += [(f"{up}.0.conv.weight", (d, d, f)), (f"{up}.0.conv.bias", (d,)), (f"{up}.1.dwconv.conv.weight", (d, 1, 7)),
                (f"{up}.1.dwconv.conv.bias", (d,)), (f"{up}.1.norm.weight", (d,)), (f"{up}.1.norm.bias", (d,)),
                (f"{up}.1.pwconv1.weight", (4 * d, d)), (f"{up}.1.pwconv1.bias", (4 * d,)), (f"{up}.1.pwconv2.weight", (d, 4 * d)),
                (f"{up}.1.pwconv2.bias", (d,)), (f"{up}.1.gamma", (d,))]

    def wn(p: str, w_shape: Shape, co: int) -> None:
        out.append((f"{p}.conv.parametrizations.weight.original0", (w_shape[0], 1, 1)))
        out.append((f"{p}.conv.parametrizations.weight.original1", w_shape))
        out.append((f"{p}.conv.bias", (co,)))

    dm, ch = "decoder.model", cfg.decoder_dim
    wn(f"{dm}.0", (ch, d, 7), ch)
    for i, r in enumerate(cfg.decoder_rates):
        ci, co = ch // 2 ** i, ch // 2 ** (i + 1)
        bp = f"{dm}.{i + 1}.block"
        out.append((f"{bp}.0.alpha", (1, ci, 1)))
        wn(f"{bp}.1", (ci, co, 2 * r), co)
        for j in range(3):
            rp = f"{bp}.{2 + j}.block"
            out.append((f"{rp}.0.alpha", (1, co, 1)))
            wn(f"{rp}.1", (co, co, 7), co)
            out.append((f"{rp}.2.alpha", (1, co, 1)))
            wn(f"{rp}.3", (co, co, 1), co)
    n = len(cfg.decoder_rates)
    out.append((f"{dm}.{n + 1}.alpha", (1, ch // 2 ** n, 1)))
    wn(f"{dm}.{n + 2}", (1, ch // 2 ** n, 7), 1)
    return out


_CONVT = __import__("re").compile(r"^decoder\.model\.\d+\.block\.1\.conv\.")


def random_dac_state(cfg: DACConfig, device, seed: int = 0) -> Dict[str, torch.Tensor]:
    """Random, well-conditioned decode-path checkpoint (conv gain 0.5 through the weight-norm g, small layer scales)."""
    g = torch.Generator(device=device).manual_seed(seed)
    sd: Dict[str, torch.Tensor] = {}

    def r(shape: Shape, std: float) -> torch.Tensor:
        return torch.randn(shape, device=device, generator=g) * std

    for name, shape in dac_param_shapes(cfg):
        if name.endswith("original1"):
            fan = (shape[0] if _CONVT.match(name) else shape[1]) * shape[2]
            sd[name] = r(shape, 1.0 / math.sqrt(fan))
        elif name.endswith("original0"):
            sd[name] = torch.empty(shape, device=device)          # set from ||v|| below
        elif name.endswith("alpha"):
            sd[name] = 1.0 + r(shape, 0.2)
        elif "layer_scale.gamma" in name:
            sd[name] = 0.2 + r(shape, 0.05)
        elif name.endswith(".1.gamma"):
            sd[name] = 0.3 + r(shape, 0.05)
        elif "norm.weight" in name:
            sd[name] = 1.0 + r(shape, 0.1)
        elif len(shape) == 1:
            sd[name] = r(shape, 0.02)
        elif "dwconv" in name:
            sd[name] = r(shape, 0.3)
        else:
            sd[name] = r(shape, 1.0 / math.sqrt(shape[1]))
    for name in list(sd):
        if name.endswith("original0"):
            v = sd[name.replace("original0", "original1")]
            sd[name] = 0.5 * v.flatten(1).norm(dim=1).view(-1, 1, 1)
    return sd


def fake_quant_fp8_e4m3(w: torch.Tensor) -> torch.Tensor:
    """Weight-only fp8 storage numerics (BASELINE config C5): per-output-row scale to the OCP e4m3 range (max 448), round to
    `torch.float8_e4m3fn`, scale back.  The result is what an fp8 weight path multiplies with; this repo has no fp8 MFMA kernel
    yet, so the dequantised matrix runs on the bf16 kernels (same numerics as dequantise-on-load, no speed-up)."""
    wf = w.detach().float()
    s = wf.abs().amax(dim=-1, keepdim=True).clamp_min(1e-12) / 448.0
    q = (wf / s).to(torch.float8_e4m3fn).float() * s
    return q.to(w.dtype)


def fp8_weight_state(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """EchoDiT state dict with every transformer-block linear of the DiT (attention, MLP, the conditioning keys' projections)
    passed through fake_quant_fp8_e4m3; norms, embeddings, AdaLN low-rank factors and biases stay as they are."""
    out = {}
    for k, v in sd.items():
        lin = v.dim() == 2 and k.startswith("blocks.") and (".attention.w" in k or ".attention.gate" in k or ".mlp.w" in k)
        out[k] = fake_quant_fp8_e4m3(v) if lin else v
    return out


def save_fp8_scales(path: str, scales: torch.Tensor, meta: Dict[str, object] | None = None) -> None:
    """Static fp8 activation scales of `EchoDiT.fp8_calibration_finish` as a small JSON file next to a checkpoint
    ({"scales": [[attention_out, swiglu_out] per block], "meta": {...}})."""
    import json
    with open(path, "w") as f:
        json.dump({"format": "echo-hip fp8 activation scales v1", "scales": [[float(a), float(b)] for a, b in scales.float().cpu().tolist()],
                   "meta": dict(meta or {})}, f, indent=1)


def load_fp8_scales(path: str) -> torch.Tensor:
    import json
    with open(path) as f:
        d = json.load(f)
    t = torch.tensor(d["scales"], dtype=torch.float32)
    if t.dim() != 2 or t.shape[1] != 2 or not bool((t > 0).all()) or not bool(torch.isfinite(t).all()):
        raise ValueError(f"{path}: not a (num_layers, 2) table of positive scales")
    return t
