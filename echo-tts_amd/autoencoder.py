"""HIP-backed Fish S1-DAC with the reference's call surface (autoencoder.py:1080-1138: `DAC.decode_zq`, `DAC.encode_zq`,
`DAC.encode`, `.device`, `.dtype`): the decode path of the hot loop and the encode path of the speaker reference.  The checkpoint keeps the reference's state-dict names; this
loader folds weight-norm (w = g * v / ||v||, autoencoder.py:90-94) and reshapes each Conv1d /
ConvTranspose1d kernel into the GEMM form the HIP taps-GEMM consumes (channels-last activations).
"""
from __future__ import annotations

import os
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
    # encode path (speaker reference -> latents), autoencoder.py:1144-1192
    encoder_dim: int = 64
    encoder_rates: Tuple[int, ...] = (2, 4, 8, 8)
    encoder_transformer_layers: Tuple[int, ...] = (0, 0, 0, 4)
    encoder_window: int = 512
    n_codebooks: int = 9
    codebook_size: int = 1024
    codebook_dim: int = 8
    semantic_codebook_size: int = 4096

    @property
    def frame_length(self) -> int:
        h = 1
        for r in tuple(self.encoder_rates) + tuple(self.upsample_factors):
            h *= r
        return h

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
        self._has_encoder = any(k.startswith("encoder.") for k in state_dict)
        if self._has_encoder:
            cfg.dac_enc_dim, cfg.dac_enc_n_rates, cfg.dac_enc_window = c.encoder_dim, len(c.encoder_rates), c.encoder_window
            for i, (r, nt) in enumerate(zip(c.encoder_rates, c.encoder_transformer_layers)):
                cfg.dac_enc_rates[i], cfg.dac_enc_tlayers[i] = r, nt
            cfg.dac_n_codebooks, cfg.dac_codebook_size = c.n_codebooks, c.codebook_size
            cfg.dac_codebook_dim, cfg.dac_semantic_size = c.codebook_dim, c.semantic_codebook_size
        ctx = C.c_void_p()
        L.check(self._lib.echo_ctx_create(C.byref(cfg), self._device.index or 0, C.byref(ctx)))
        self._ctx = ctx
        torch.cuda.set_device(self._device)
        self._load(state_dict)
        pm = "quantizer.post_module.freqs_cis"
        # one table serves post_module / pre_module (block_size 4096) and the encoder transformer (block_size 16384,
        # autoencoder.py:1163-1165): same head_dim and base, so the shorter caches are prefixes of the longer one
        npos = max(c.post_block_size, 16384 if self._has_encoder else 0)
        if pm in state_dict and state_dict[pm].shape[0] >= npos:  # persistent buffer wins (autoencoder.py:562-569)
            cache = state_dict[pm].float().contiguous()
        else:
            cache = ae_rope_cache(npos, c.post_head_dim, c.rope_base)
        self._rope = cache.to(self._device)
        L.check(self._lib.echo_set_ae_rope_table(self._ctx, self._rope.data_ptr(), self._rope.shape[0]), self._ctx)
        self._pca_key = None
        self._pca_enc_key = None
        if self._has_encoder:
            self._load_encoder(state_dict)

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

    def _load_encoder(self, sd: Dict[str, torch.Tensor]) -> None:
        """Encode-path tensors under "enc.*": weight-norm folded, conv kernels in GEMM form (a stride-s conv with k = 2s
        reads rows of s*Ci: its (Co, k*Ci) tap-major matrix is used as a 2-tap GEMM with K = s*Ci), codebooks L2-normalised
        with the reference's own expression (F.normalize, autoencoder.py:148-149), from_codes operands concatenated."""
        c = self.config
        ep = "encoder.block"
        self._put("enc.conv0.w", _fold(sd, f"{ep}.0.conv")[:, 0, :])
        self._put("enc.conv0.b", sd[f"{ep}.0.conv.bias"])
        n = len(c.encoder_rates)
        tkeys = ("attention.wqkv.weight", "attention.wo.weight", "feed_forward.w1.weight", "feed_forward.w3.weight",
                 "feed_forward.w2.weight", "ffn_norm.weight", "attention_norm.weight", "attention_layer_scale.gamma",
                 "ffn_layer_scale.gamma")
        for i, nt in enumerate(c.encoder_transformer_layers):
            bp = f"{ep}.{i + 1}.block"
            for j in range(3):
                rp = f"{bp}.{j}.block"
                self._put(f"enc.b{i}.ru{j}.a0", sd[f"{rp}.0.alpha"].flatten())
                self._put(f"enc.b{i}.ru{j}.w7", _conv_as_gemm(_fold(sd, f"{rp}.1.conv")))
                self._put(f"enc.b{i}.ru{j}.b7", sd[f"{rp}.1.conv.bias"])
                self._put(f"enc.b{i}.ru{j}.a1", sd[f"{rp}.2.alpha"].flatten())
                self._put(f"enc.b{i}.ru{j}.w1", _conv_as_gemm(_fold(sd, f"{rp}.3.conv")))
                self._put(f"enc.b{i}.ru{j}.b1", sd[f"{rp}.3.conv.bias"])
            self._put(f"enc.b{i}.alpha", sd[f"{bp}.3.alpha"].flatten())
            self._put(f"enc.b{i}.wc", _conv_as_gemm(_fold(sd, f"{bp}.4.conv")))
            self._put(f"enc.b{i}.bc", sd[f"{bp}.4.conv.bias"])
            for l in range(nt):
                for k in tkeys:
                    self._put(f"enc.b{i}.t.layers.{l}.{k}", sd[f"{bp}.5.layers.{l}.{k}"])
            if nt:
                self._put(f"enc.b{i}.t.norm.weight", sd[f"{bp}.5.norm.weight"])
        self._put("enc.final.alpha", sd[f"{ep}.{n + 1}.alpha"].flatten())
        self._put("enc.final.w", _conv_as_gemm(_fold(sd, f"{ep}.{n + 2}.conv")))
        self._put("enc.final.b", sd[f"{ep}.{n + 2}.conv.bias"])
        for i in range(len(c.upsample_factors)):
            dn = f"quantizer.downsample.{i}"
            self._put(f"enc.down{i}.w", _conv_as_gemm(_fold(sd, f"{dn}.0.conv")))
            self._put(f"enc.down{i}.b", sd[f"{dn}.0.conv.bias"])
            self._put(f"enc.down{i}.dw", _fold(sd, f"{dn}.1.dwconv.conv"))
            for src, dst in (("dwconv.conv.bias", "db"), ("norm.weight", "lnw"), ("norm.bias", "lnb"), ("pwconv1.weight", "p1w"),
                             ("pwconv1.bias", "p1b"), ("pwconv2.weight", "p2w"), ("pwconv2.bias", "p2b"), ("gamma", "gamma")):
                self._put(f"enc.down{i}.{dst}", sd[f"{dn}.1.{src}"])
        pm = "quantizer.pre_module"
        for l in range(c.post_layers):
            for k in tkeys:
                self._put(f"enc.pre.layers.{l}.{k}", sd[f"{pm}.layers.{l}.{k}"])
        self._put("enc.pre.norm.weight", sd[f"{pm}.norm.weight"])
        names = ["quantizer.semantic_quantizer.quantizers.0"] + [f"quantizer.quantizer.quantizers.{i}" for i in range(c.n_codebooks)]
        fc_w, fc_b = [], 0.0
        for q, p in enumerate(names):
            cb = sd[f"{p}.codebook.weight"].float()
            out_w = _fold(sd, f"{p}.out_proj")[:, :, 0]                      # (C, 8)
            self._put(f"enc.vq{q}.in_w", _fold(sd, f"{p}.in_proj")[:, :, 0])  # (8, C)
            self._put(f"enc.vq{q}.in_b", sd[f"{p}.in_proj.bias"])
            self._put(f"enc.vq{q}.cbn", torch.nn.functional.normalize(cb))
            self._put(f"enc.vq{q}.cb", cb)
            self._put(f"enc.vq{q}.out_w", out_w)
            self._put(f"enc.vq{q}.out_nb", -sd[f"{p}.out_proj.bias"].float())
            fc_w.append(out_w)
            fc_b = fc_b + sd[f"{p}.out_proj.bias"].float()
        self._put("enc.fc.w", torch.cat(fc_w, dim=1))
        self._put("enc.fc.b", fc_b)
        L.check(self._lib.echo_finalize_dac_encoder(self._ctx, self._stream()), self._ctx)

    # ------------------------------------------------------------------ encode (speaker reference)
    def set_pca_encode(self, pca_state) -> None:
        key = (id(pca_state.pca_components), id(pca_state.pca_mean), float(pca_state.latent_scale))
        if key == self._pca_enc_key:
            return
        comp = pca_state.pca_components.detach().float().cpu()                       # (latent, C)
        bias = -(pca_state.pca_mean.detach().double().cpu() @ comp.double().T).float()  # mean folded: (z - m) P^T = z P^T - m P^T
        w = comp.contiguous().to(self._device)
        b = bias.contiguous().to(self._device)
        L.check(self._lib.echo_set_pca_encode(self._ctx, w.data_ptr(), b.data_ptr(), float(pca_state.latent_scale), 1, self._stream()),
                self._ctx)
        self._pca_enc_key = key

    def _encode_raw(self, audio_data: torch.Tensor, want_latent: bool):
        if not self._has_encoder:
            raise L.EchoHipError("this DAC was loaded without encoder weights")
        a = audio_data.to(self._device, torch.float32)
        if a.ndim == 2:
            a = a.unsqueeze(1)
        assert a.ndim == 3 and a.shape[1] == 1, "audio must be (B, 1, length)"
        B, _, n = a.shape
        fl = self.config.frame_length
        npad = -(-n // fl) * fl
        a = torch.nn.functional.pad(a, (0, npad - n)).contiguous()               # autoencoder.py:1090-1092
        T, c = npad // fl, self.config
        nq = 1 + c.n_codebooks
        codes = torch.empty((B, nq, T), dtype=torch.int32, device=self._device)
        zq = torch.empty((B, T, c.latent_dim), dtype=torch.float32, device=self._device)
        lat = torch.empty((B, T, c.latent_size), dtype=torch.float32, device=self._device) if want_latent else None
        for b in range(B):
            L.check(self._lib.echo_dac_encode(self._ctx, a[b].data_ptr(), npad, lat[b].data_ptr() if want_latent else None,
                                              codes[b].data_ptr(), zq[b].data_ptr(), self._stream()), self._ctx)
        return codes, zq, lat

    @torch.no_grad()
    def encode(self, audio_data: torch.Tensor, audio_lengths=None, n_quantizers=None, **kw):
        """autoencoder.py:1080-1108: (B, 1, L) or (B, L) audio -> (codes (B, 1 + n_codebooks, T) int64, lengths)."""
        codes, _, _ = self._encode_raw(audio_data, False)
        n = audio_data.shape[-1]
        fl = self.config.frame_length
        lens = torch.full((codes.shape[0],), -(-n // fl), dtype=torch.long, device=self._device)
        return codes.long(), lens

    @torch.no_grad()
    def encode_zq(self, audio_data: torch.Tensor) -> torch.Tensor:
        """autoencoder.py:1117-1126: (B, 1, L) -> z_q (B, latent_dim, T)."""
        _, zq, _ = self._encode_raw(audio_data, False)
        return zq.transpose(1, 2)

    @torch.no_grad()
    def encode_latent(self, audio_data: torch.Tensor, pca_state) -> torch.Tensor:
        """inference.py:218-224 in one engine call per item: (B, 1, L) -> (B, T, latent_size) fp32."""
        self.set_pca_encode(pca_state)
        _, _, lat = self._encode_raw(audio_data, True)
        return lat

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
        if os.environ.get("ECHO_DAC_BATCH", "1") == "0":      # A/B aid: one decode call per item (the round-1 path)
            for b in range(B):
                L.check(self._lib.echo_dac_decode(self._ctx, lat[b].data_ptr(), T, self._latent_scale, out[b].data_ptr(), self._stream()), self._ctx)
            return out
        # one call for the batch: PCA inverse + post_module transformer on the B * T stacked rows, the convolution stack per item
        L.check(self._lib.echo_dac_decode_batch(self._ctx, lat.data_ptr(), B, T, self._latent_scale, out.data_ptr(), T * self.config.hop,
                                                self._stream()), self._ctx)
        return out

    @torch.no_grad()
    def decode_latent_tail(self, latent: torch.Tensor, first_frame: int) -> torch.Tensor:
        """Streaming building block: `latent` (T, latent_size) are all frames so far; returns the (T - first_frame) * hop samples
        of frames first_frame.. (the post_module transformer sees every frame, the convolutional stack only the tail)."""
        lat = latent.to(self._device, torch.float32).contiguous()
        T = lat.shape[0]
        out = torch.empty(((T - first_frame) * self.config.hop,), dtype=torch.float32, device=self._device)
        L.check(self._lib.echo_dac_decode_tail(self._ctx, lat.data_ptr(), T, int(first_frame), self._latent_scale, out.data_ptr(),
                                               self._stream()), self._ctx)
        return out

    def conv_context_frames(self) -> int:
        """Latent frames of left context after which the causal convolutional stack (quantizer.upsample + Decoder,
        autoencoder.py:427-435, 971-998) no longer sees the start of its input: dwconv k7 per upsample stage, conv k7, per block
        the ConvTranspose (k = 2 stride: one input step) and three ResidualUnits (k7, dilations 1, 3, 9), final conv k7.  Every
        term is (kernel span in samples) / (samples per latent frame at that layer); rounded up, plus two frames of margin."""
        c = self.config
        rate, need = 1.0, 0.0
        for f in c.upsample_factors:
            rate *= f
            need += 6.0 / rate
        need += 6.0 / rate
        for r in c.decoder_rates:
            need += 1.0 / rate
            rate *= r
            need += 6.0 * (1 + 3 + 9) / rate
        need += 6.0 / rate
        return int(need + 0.999999) + 2

    @torch.no_grad()
    def decode_zq(self, z_q: torch.Tensor) -> torch.Tensor:
        """autoencoder.py:1128-1132: (B, latent_dim, T) -> (B, 1, T*hop)."""
        z = z_q.to(self._device, torch.float32).transpose(1, 2).contiguous()    # channels-last
        B, T, _ = z.shape
        out = torch.empty((B, 1, T * self.config.hop), dtype=torch.float32, device=self._device)
        for b in range(B):
            L.check(self._lib.echo_dac_decode_zq(self._ctx, z[b].data_ptr(), T, out[b].data_ptr(), self._stream()), self._ctx)
        return out




class DACStream:
    """Causal chunked decode of ONE utterance: feed the latents of each generated block as they arrive and
    receive exactly the samples of those frames, equal to a whole-utterance `ae_decode` (tests/test_gpu_engine.py: <= 1e-6).
    The reference decodes whole utterances only (inference.py:226-229, gradio_app.py:43) although every convolution of the
    decoder is causal (autoencoder.py:264-331); block k is therefore audible while block k + 1 is still being sampled.

    Each `push` re-runs the cheap window-limited transformer (3 % of the decode FLOPs) over the frames so far and the
    convolutional stack over the new frames plus `context` frames of left context that are decoded again and dropped."""

    def __init__(self, fish_ae: DAC, pca_state, context: Optional[int] = None):
        self.ae = fish_ae
        fish_ae.set_pca(pca_state)
        self.context = fish_ae.conv_context_frames() if context is None else int(context)
        self.latents: Optional[torch.Tensor] = None
        self.emitted = 0

    @torch.no_grad()
    def push(self, block: torch.Tensor) -> torch.Tensor:
        """block: (n, latent_size) or (1, n, latent_size) new latents -> (1, 1, n * hop) samples of exactly those frames."""
        if block.dim() == 3:
            if block.shape[0] != 1:
                raise ValueError("DACStream decodes one utterance; use one stream per batch item")
            block = block[0]
        block = block.to(self.ae.device, torch.float32)
        self.latents = block if self.latents is None else torch.cat([self.latents, block], dim=0)
        f0 = max(0, self.emitted - self.context)
        wav = self.ae.decode_latent_tail(self.latents, f0)
        wav = wav[(self.emitted - f0) * self.ae.config.hop:]
        self.emitted = self.latents.shape[0]
        return wav.view(1, 1, -1)
