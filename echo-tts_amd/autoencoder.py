"""HIP-backed Fish S1-DAC *decode* path with the reference's call surface (autoencoder.py:1128-1138:
`DAC.decode_zq`, `.device`, `.dtype`).  The checkpoint keeps the reference's state-dict names; this
loader folds weight-norm (w = g * v / ||v||, autoencoder.py:90-94) and reshapes each Conv1d /
ConvTranspose1d kernel into the GEMM form the HIP taps-GEMM consumes (channels-last activations).
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Dict, Optional, Tuple

import torch

from . import _lib as L


@dataclass
class DACConfig:
    """Decode-path sizes of build_ae() (autoencoder.py:1144-1192)."""
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
    latent_size: int = 80

    @classmethod
    def from_any(cls, o) -> "DACConfig":
        kw = {}
        for f in cls.__dataclass_fields__:
            if hasattr(o, f):
                kw[f] = getattr(o, f)
        return cls(**kw)

    @property
    def hop(self) -> int:
        h = 1
        for r in tuple(self.decoder_rates) + tuple(self.upsample_factors):
            h *= r
        return h


def _fold(sd: Dict[str, torch.Tensor], p: str) -> torch.Tensor:
    """Effective conv weight: plain `.weight`, or weight-norm parametrization folded once."""
    if f"{p}.weight" in sd:
        return sd[f"{p}.weight"].float()
    g = sd[f"{p}.parametrizations.weight.original0"].float()
    v = sd[f"{p}.parametrizations.weight.original1"].float()
    return torch._weight_norm(v, g, 0)


def _conv_as_gemm(w: torch.Tensor) -> torch.Tensor:
    """Conv1d weight (Co, Ci, k) -> (Co, k*Ci): tap-major K so that tap j multiplies input row t - (k-1-j)*dilation."""
    co, ci, k = w.shape
    return w.permute(0, 2, 1).reshape(co, k * ci).contiguous()


def _convT_as_gemm(w: torch.Tensor, stride: int) -> torch.Tensor:
    """ConvTranspose1d weight (Ci, Co, k) with k in {stride, 2*stride} -> (stride*Co, taps*Ci).
    Output row q*stride + r of the causal transposed conv = x[q]·w[:, :, r] (+ x[q-1]·w[:, :, r+stride] when k = 2*stride);
    tap 0 of the GEMM reads x[q-1], tap 1 reads x[q]."""
    ci, co, k = w.shape
    if k == stride:
        return w.permute(2, 1, 0).reshape(stride * co, ci).contiguous()
    assert k == 2 * stride
    return w.view(ci, co, 2, stride).flip(2).permute(3, 1, 2, 0).reshape(stride * co, 2 * ci).contiguous()


def ae_rope_cache(seq_len: int, n_elem: int, base: float = 10000.0) -> torch.Tensor:
    """bf16 cos/sin cache of autoencoder.py:805-813, returned as fp32 (seq, n_elem/2, 2)."""
    freqs = 1.0 / (base ** (torch.arange(0, n_elem, 2)[: n_elem // 2].float() / n_elem))
    ang = torch.outer(torch.arange(seq_len), freqs)
    cis = torch.polar(torch.ones_like(ang), ang)
    return torch.stack([cis.real, cis.imag], dim=-1).to(torch.bfloat16).float().contiguous()


class DAC:
    """Decode-only Fish S1-DAC on libechohip (fp32 activations and fp32 MFMA, the reference's default AE dtype)."""

    def __init__(self, config, state_dict: Dict[str, torch.Tensor], device: str | torch.device = "cuda:0"):
        self.config = DACConfig.from_any(config)
        self._device = torch.device(device)
        if self._device.type != "cuda":
            raise L.EchoHipError("DAC (HIP) needs a cuda (ROCm) device; there is no CPU path")
        self._lib = L.load_library()
        c = self.config
        cfg = L.EchoConfig()
        cfg.precision = L.ECHO_F32
        cfg.latent_size = c.latent_size
        cfg.dac_latent_dim, cfg.dac_decoder_dim, cfg.dac_n_rates = c.latent_dim, c.decoder_dim, len(c.decoder_rates)
        for i, r in enumerate(c.decoder_rates):
            cfg.dac_rates[i] = r
        cfg.dac_post_layers, cfg.dac_post_heads, cfg.dac_post_head_dim = c.post_layers, c.post_heads, c.post_head_dim
        cfg.dac_post_ffn, cfg.dac_post_window = c.post_ffn, c.post_window
        cfg.dac_n_up = len(c.upsample_factors)
        for i, f in enumerate(c.upsample_factors):
            cfg.dac_up_factors[i] = f
        cfg.dac_norm_eps = c.norm_eps
        ctx = C.c_void_p()
        L.check(self._lib.echo_ctx_create(C.byref(cfg), self._device.index or 0, C.byref(ctx)))
        self._ctx = ctx
        torch.cuda.set_device(self._device)
        self._load(state_dict)
        pm = "quantizer.post_module.freqs_cis"
        if pm in state_dict:  # persistent buffer wins (autoencoder.py:562-569)
            cache = state_dict[pm].float().contiguous()
        else:
            cache = ae_rope_cache(c.post_block_size, c.post_head_dim, c.rope_base)
        self._rope = cache.to(self._device)
        L.check(self._lib.echo_set_ae_rope_table(self._ctx, self._rope.data_ptr(), self._rope.shape[0]), self._ctx)
        self._pca_key = None

    def __del__(self):
        try:
            if getattr(self, "_ctx", None):
                self._lib.echo_ctx_destroy(self._ctx)
                self._ctx = None
        except Exception:
            pass

    @property
    def device(self) -> torch.device:
        return self._device

    @property
    def dtype(self) -> torch.dtype:
        return torch.float32

    def eval(self) -> "DAC":
        return self

    def _stream(self) -> int:
        return torch.cuda.current_stream(self._device).cuda_stream

    def _put(self, name: str, t: torch.Tensor) -> None:
        t = t.detach().float().contiguous()
        shape = (L.c_i64 * max(t.dim(), 1))(*(list(t.shape) or [1]))
        L.check(self._lib.echo_load_tensor(self._ctx, name.encode(), t.data_ptr(), L.ECHO_F32, max(t.dim(), 1), shape,
                                           int(t.is_cuda)), self._ctx)

    def _load(self, sd: Dict[str, torch.Tensor]) -> None:
        c = self.config
        pm = "quantizer.post_module"
        for i in range(c.post_layers):
            lp = f"{pm}.layers.{i}"
            for n in ("attention.wqkv.weight", "attention.wo.weight", "feed_forward.w1.weight", "feed_forward.w3.weight",
                      "feed_forward.w2.weight", "ffn_norm.weight", "attention_norm.weight", "attention_layer_scale.gamma",
                      "ffn_layer_scale.gamma"):
                self._put(f"{lp}.{n}", sd[f"{lp}.{n}"])
        self._put(f"{pm}.norm.weight", sd[f"{pm}.norm.weight"])
        ups = list(reversed(c.upsample_factors))
        for i, f in enumerate(ups):
            up = f"quantizer.upsample.{i}"
            self._put(f"{up}.0.conv.weight", _convT_as_gemm(_fold(sd, f"{up}.0.conv"), f))
            self._put(f"{up}.0.conv.bias", sd[f"{up}.0.conv.bias"])
            self._put(f"{up}.1.dwconv.conv.weight", _fold(sd, f"{up}.1.dwconv.conv"))
            for n in ("dwconv.conv.bias", "norm.weight", "norm.bias", "pwconv1.weight", "pwconv1.bias", "pwconv2.weight",
                      "pwconv2.bias", "gamma"):
                self._put(f"{up}.1.{n}", sd[f"{up}.1.{n}"])
        dm = "decoder.model"
        self._put(f"{dm}.0.conv.weight", _conv_as_gemm(_fold(sd, f"{dm}.0.conv")))
        self._put(f"{dm}.0.conv.bias", sd[f"{dm}.0.conv.bias"])
        for i, r in enumerate(c.decoder_rates):
            bp = f"{dm}.{i + 1}.block"
            self._put(f"{bp}.0.alpha", sd[f"{bp}.0.alpha"].flatten())
            self._put(f"{bp}.1.conv.weight", _convT_as_gemm(_fold(sd, f"{bp}.1.conv"), r))
            self._put(f"{bp}.1.conv.bias", sd[f"{bp}.1.conv.bias"])
            for j in range(3):
                rp = f"{bp}.{2 + j}.block"
                self._put(f"{rp}.0.alpha", sd[f"{rp}.0.alpha"].flatten())
                self._put(f"{rp}.1.conv.weight", _conv_as_gemm(_fold(sd, f"{rp}.1.conv")))
                self._put(f"{rp}.1.conv.bias", sd[f"{rp}.1.conv.bias"])
                self._put(f"{rp}.2.alpha", sd[f"{rp}.2.alpha"].flatten())
                self._put(f"{rp}.3.conv.weight", _conv_as_gemm(_fold(sd, f"{rp}.3.conv")))
                self._put(f"{rp}.3.conv.bias", sd[f"{rp}.3.conv.bias"])
        n = len(c.decoder_rates)
        self._put(f"{dm}.{n + 1}.alpha", sd[f"{dm}.{n + 1}.alpha"].flatten())
        wout = _fold(sd, f"{dm}.{n + 2}.conv")                       # (1, C, 7) -> (7, C)
        self._put(f"{dm}.{n + 2}.conv.weight", wout[0].t().contiguous())
        self._put(f"{dm}.{n + 2}.conv.bias", sd[f"{dm}.{n + 2}.conv.bias"])
        L.check(self._lib.echo_finalize_dac(self._ctx, self._stream()), self._ctx)

    def set_profiling(self, on: bool) -> None:
        L.check(self._lib.echo_set_profiling(self._ctx, int(on)), self._ctx)

    def get_profile(self) -> L.EchoProfile:
        p = L.EchoProfile()
        L.check(self._lib.echo_get_profile(self._ctx, C.byref(p)), self._ctx)
        return p

    # ------------------------------------------------------------------ decode
    def set_pca(self, pca_state) -> None:
        key = (id(pca_state.pca_components), id(pca_state.pca_mean), float(pca_state.latent_scale))
        if key == self._pca_key:
            return
        w = pca_state.pca_components.detach().float().t().contiguous().to(self._device)    # (C, latent)
        m = pca_state.pca_mean.detach().float().contiguous().to(self._device)
        L.check(self._lib.echo_set_pca(self._ctx, w.data_ptr(), m.data_ptr(), 1, self._stream()), self._ctx)
        self._pca_key = key
        self._latent_scale = float(pca_state.latent_scale)

    @torch.no_grad()
    def decode_latent(self, latent: torch.Tensor) -> torch.Tensor:
        """(B, T, latent_size) fp32 sampler output -> (B, 1, T*hop) fp32 waveform (inference.py:226-229)."""
        lat = latent.to(self._device, torch.float32).contiguous()
        B, T, _ = lat.shape
        out = torch.empty((B, 1, T * self.config.hop), dtype=torch.float32, device=self._device)
        for b in range(B):
            L.check(self._lib.echo_dac_decode(self._ctx, lat[b].data_ptr(), T, self._latent_scale, out[b].data_ptr(),
                                              self._stream()), self._ctx)
        return out

    @torch.no_grad()
    def decode_zq(self, z_q: torch.Tensor) -> torch.Tensor:
        """autoencoder.py:1128-1132: (B, latent_dim, T) -> (B, 1, T*hop)."""
        z = z_q.to(self._device, torch.float32).transpose(1, 2).contiguous()    # channels-last
        B, T, _ = z.shape
        out = torch.empty((B, 1, T * self.config.hop), dtype=torch.float32, device=self._device)
        for b in range(B):
            L.check(self._lib.echo_dac_decode_zq(self._ctx, z[b].data_ptr(), T, out[b].data_ptr(), self._stream()), self._ctx)
        return out

    def encode_zq(self, audio_data: torch.Tensor) -> torch.Tensor:
        raise NotImplementedError("DAC encode is the next scope row (SURVEY.md §8f-1)")
