"""HIP-backed EchoDiT: same call surface as the reference `model.EchoDiT` (model.py:472-642) for the
inference path (`forward`, `get_kv_cache_text/speaker/latent`, `.device`, `.dtype`), with every
tensor op executed by libechohip on an MI355X.  PyTorch is only used for device memory, streams
and the small host-side tables the reference also builds on the CPU (RoPE, timestep embedding).
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, fields
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from . import _lib as L


@dataclass
class EchoDiTConfig:
    """Constructor arguments of the reference EchoDiT (model.py:473-497); defaults from inference.py:16-24."""
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

    @classmethod
    def from_any(cls, obj) -> "EchoDiTConfig":
        return cls(**{f.name: getattr(obj, f.name) for f in fields(cls)})


def _dtype_code(dt: torch.dtype) -> int:
    if dt == torch.float32:
        return L.ECHO_F32
    if dt == torch.bfloat16:
        return L.ECHO_BF16
    raise ValueError(f"unsupported dtype {dt}")


def rope_table(head_dim: int, npos: int) -> torch.Tensor:
    """(npos, head_dim/2, 2) fp32 [cos, sin]; same arithmetic as model.py:9-14 (CPU, fp32)."""
    inv = 1.0 / (10000.0 ** (torch.arange(0, head_dim, 2)[: head_dim // 2] / head_dim))
    ang = torch.outer(torch.arange(npos), inv)
    return torch.stack([torch.cos(ang), torch.sin(ang)], dim=-1).contiguous()


def timestep_embedding(t: torch.Tensor, size: int) -> torch.Tensor:
    """model.py:27-43 evaluated on the host; `t` already has the model dtype."""
    half = size // 2
    freqs = 1000 * torch.exp(-torch.log(torch.tensor(10000.0)) * torch.arange(start=0, end=half, dtype=torch.float32) / half)
    args = t[..., None] * freqs[None]
    return torch.cat([torch.cos(args), torch.sin(args)], dim=-1).to(t.dtype)


def mask_to_keys(mask: torch.Tensor) -> Tuple[List[int], Optional[torch.Tensor]]:
    """Boolean key mask (B, T) -> (#leading keys to visit per row, additive 0/-inf bias or None if all rows are prefixes)."""
    m = mask.detach().to("cpu", torch.bool)
    nk, prefix = [], True
    for row in m:
        idx = torch.nonzero(row).flatten()
        n = int(idx[-1]) + 1 if idx.numel() else 0
        nk.append(n)
        if n and not bool(row[:n].all()):
            prefix = False
    if prefix:
        return nk, None
    bias = torch.zeros(m.shape, dtype=torch.float32)
    bias[~m] = float("-inf")
    return nk, bias


class KVHandle:
    """Stands in for the reference's List[Tuple[K, V]] caches: the tensors live inside the HIP context."""

    def __init__(self, model: "EchoDiT", kind: str, mask: Optional[torch.Tensor], batch: int, repeat: int = 1):
        self.model, self.kind, self.mask, self.batch, self.repeat = model, kind, mask, batch, repeat

    def repeated(self, n: int) -> "KVHandle":
        return KVHandle(self.model, self.kind, self.mask, self.batch, self.repeat * n)

    def layer(self, i: int) -> Tuple[torch.Tensor, torch.Tensor]:
        """Read one layer back as fp32 (B, T, H, 128) tensors (tests only)."""
        which = {"text": 0, "speaker": 1, "latent": 2}[self.kind]
        return self.model._read_kv(which, i)


class VoiceHandle:
    """A reference voice as the sampler consumes it: the speaker encoder output projected to the K / V of all EchoDiT layers
    (model.py:615-621), held on the device outside any engine context.  `EchoDiT.capture_voice()` makes one after
    `get_kv_cache_speaker`; `EchoDiT.bind_voice()` installs it into a context (same model / precision / device) without
    running the encoder again — bit-identical to a fresh encode (tests/test_gpu_engine.py)."""

    def __init__(self, lib, ptr, mask: Optional[torch.Tensor], batch: int):
        self._lib, self._ptr, self.mask, self.batch = lib, ptr, mask, batch

    @property
    def nbytes(self) -> int:
        return int(self._lib.echo_voice_bytes(self._ptr)) if self._ptr else 0

    def close(self) -> None:
        if getattr(self, "_ptr", None):
            self._lib.echo_voice_destroy(self._ptr)
            self._ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class EchoDiT:
    """model.py:472 `EchoDiT`, inference only, executed by libechohip."""

    MAX_POS = 4096

    def __init__(self, config, state_dict: Dict[str, torch.Tensor], dtype: torch.dtype = torch.bfloat16,
                 device: str | torch.device = "cuda:0", fp8: bool = False):
        """`fp8=True` (BASELINE config C5; bf16 engine only): the four large linears of every block run on e4m3 operands
        (weights quantised once per output row, activations per token row) at twice the bf16 MFMA rate."""
        self.config = EchoDiTConfig.from_any(config)
        self.fp8 = bool(fp8)
        self._dtype = dtype
        self._device = torch.device(device)
        if self._device.type != "cuda":
            raise L.EchoHipError("EchoDiT (HIP) needs a cuda (ROCm) device; there is no CPU path")
        self._lib = L.load_library()
        has_latent = any(k.startswith("latent_encoder.") for k in state_dict)
        cfg = L.EchoConfig()
        cfg.precision = _dtype_code(dtype)
        for f in fields(EchoDiTConfig):
            setattr(cfg, f.name, getattr(self.config, f.name))
        cfg.has_latent_encoder = int(has_latent)
        cfg.dit_fp8 = int(self.fp8)
        self.has_latent_encoder = has_latent
        ctx = C.c_void_p()
        L.check(self._lib.echo_ctx_create(C.byref(cfg), self._device.index or 0, C.byref(ctx)))
        self._ctx = ctx
        torch.cuda.set_device(self._device)
        self._load(state_dict)
        hd = self.config.model_size // self.config.num_heads
        self._rope = rope_table(hd, self.MAX_POS).to(self._device)
        L.check(self._lib.echo_set_rope_table(self._ctx, self._rope.data_ptr(), self.MAX_POS), self._ctx)
        self._text: Optional[Tuple[torch.Tensor, List[int]]] = None
        self._kvB = 0
        self._keep: List[torch.Tensor] = []

    # ------------------------------------------------------------------ plumbing
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
        return self._dtype

    def eval(self) -> "EchoDiT":
        return self

    def _stream(self) -> int:
        return torch.cuda.current_stream(self._device).cuda_stream

    def _load(self, sd: Dict[str, torch.Tensor]) -> None:
        for name, t in sd.items():
            if t.dtype not in (torch.float32, torch.bfloat16):
                t = t.float()
            t = t.detach().contiguous()
            shape = (L.c_i64 * max(t.dim(), 1))(*(list(t.shape) or [1]))
            L.check(self._lib.echo_load_tensor(self._ctx, name.encode(), t.data_ptr(), _dtype_code(t.dtype), max(t.dim(), 1),
                                               shape, int(t.is_cuda)), self._ctx)
        L.check(self._lib.echo_finalize_dit(self._ctx, self._stream()), self._ctx)

    # ------------------------------------------------------------------ fp8 activation-scale calibration (BASELINE C5)
    def fp8_calibration_start(self) -> None:
        """Clears the per-block maxima and records, during every forward / sampler call until `fp8_calibration_finish`, the largest
        dynamic row scale (amax / 448) of the attention output and of the SwiGLU output of every block (static scales are ignored
        meanwhile).  Run representative requests in between."""
        L.check(self._lib.echo_fp8_calibrate(self._ctx, 1), self._ctx)

    def fp8_calibration_finish(self, margin: float = 1.25) -> torch.Tensor:
        """Stops recording and returns the (num_layers, 2) fp32 tensor of static scales = recorded maximum x `margin` ([:, 0] attention
        output -> wo, [:, 1] SwiGLU output -> w2).  Blocks that saw no data get the scale 1 / 448 (amax 1).  Install them with
        `set_fp8_static_scales`; `weights.save_fp8_scales` / `load_fp8_scales` keep them next to a checkpoint."""
        n = 2 * self.config.num_layers
        buf = (C.c_float * n)()
        L.check(self._lib.echo_fp8_calibration(self._ctx, buf, n), self._ctx)
        L.check(self._lib.echo_fp8_calibrate(self._ctx, 0), self._ctx)
        t = torch.tensor(list(buf), dtype=torch.float32).reshape(self.config.num_layers, 2)
        return torch.where(t > 0, t * float(margin), torch.full_like(t, 1.0 / 448.0))

    def set_fp8_static_scales(self, scales: Optional[torch.Tensor]) -> None:
        """`scales` (num_layers, 2) as returned by `fp8_calibration_finish`: the attention epilogue and the SwiGLU tail then write the
        e4m3 operands of wo / w2 themselves (values beyond 448 x scale saturate) and the two quantisation passes of a block disappear.
        None: back to dynamic per-token-row scales."""
        if scales is None:
            L.check(self._lib.echo_fp8_set_static_scales(self._ctx, None, 0), self._ctx)
            return
        flat = [float(v) for v in scales.detach().float().cpu().reshape(-1)]
        if len(flat) != 2 * self.config.num_layers:
            raise L.EchoHipError(f"fp8 static scales: expected {self.config.num_layers} x 2 values, got {len(flat)}")
        buf = (C.c_float * len(flat))(*flat)
        L.check(self._lib.echo_fp8_set_static_scales(self._ctx, buf, len(flat)), self._ctx)

    def set_profiling(self, on: bool) -> None:
        L.check(self._lib.echo_set_profiling(self._ctx, int(on)), self._ctx)

    def get_profile(self) -> L.EchoProfile:
        p = L.EchoProfile()
        L.check(self._lib.echo_get_profile(self._ctx, C.byref(p)), self._ctx)
        return p

    def _read_kv(self, which: int, layer: int) -> Tuple[torch.Tensor, torch.Tensor]:
        b, t = C.c_int(), C.c_int()
        L.check(self._lib.echo_debug_get_kv(self._ctx, which, layer, None, None, C.byref(b), C.byref(t)), self._ctx)
        D = self.config.model_size
        k = torch.empty((b.value, t.value, D), dtype=torch.float32, device=self._device)
        v = torch.empty_like(k)
        if t.value:
            L.check(self._lib.echo_debug_get_kv(self._ctx, which, layer, k.data_ptr(), v.data_ptr(), C.byref(b), C.byref(t)), self._ctx)
        H = self.config.num_heads
        return k.view(b.value, t.value, H, -1), v.view(b.value, t.value, H, -1)

    # ------------------------------------------------------------------ KV caches (model.py:606-636)
    def get_kv_cache_text(self, text_input_ids: torch.Tensor, text_mask: Optional[torch.Tensor]) -> KVHandle:
        ids = text_input_ids.to(self._device, torch.int32).contiguous()
        B, Tt = ids.shape
        if text_mask is None:
            text_mask = torch.ones((B, Tt), dtype=torch.bool)
        nk, bias = mask_to_keys(text_mask)
        bias_dev = bias.to(self._device).contiguous() if bias is not None else None
        nk_arr = (C.c_int32 * B)(*nk)
        L.check(self._lib.echo_encode_text(self._ctx, ids.data_ptr(), bias_dev.data_ptr() if bias_dev is not None else None,
                                           nk_arr, B, Tt, self._stream()), self._ctx)
        self._kvB = B
        return KVHandle(self, "text", text_mask.detach().to("cpu", torch.bool), B)

    def get_kv_cache_speaker(self, speaker_latent: torch.Tensor, speaker_mask: Optional[torch.Tensor] = None) -> KVHandle:
        """`speaker_mask` (B, Ts) is an extension: the reference passes it to forward() only; giving it here lets the
        engine skip fully masked trailing patches.  Without it every patch is encoded."""
        lat = speaker_latent.to(self._device, self._dtype).contiguous()
        B, Ts, _ = lat.shape
        ps = self.config.speaker_patch_size
        if speaker_mask is None:
            kmask = torch.ones((B, Ts // ps), dtype=torch.bool)
        else:
            kmask = speaker_mask.detach().to("cpu", torch.bool)[..., ::ps]
        nk, bias = mask_to_keys(kmask)
        bias_dev = bias.to(self._device).contiguous() if bias is not None else None
        nk_arr = (C.c_int32 * B)(*nk)
        L.check(self._lib.echo_encode_speaker(self._ctx, lat.data_ptr(), bias_dev.data_ptr() if bias_dev is not None else None,
                                              nk_arr, B, Ts, self._stream()), self._ctx)
        if not self._kvB:
            self._kvB = B
        return KVHandle(self, "speaker", kmask, B)

    # ------------------------------------------------------------------ per-voice cache
    def capture_voice(self, kv_cache_speaker: KVHandle) -> VoiceHandle:
        """Snapshot of the speaker cache `get_kv_cache_speaker` just built (call before any speaker-KV scaling)."""
        ptr = C.c_void_p()
        L.check(self._lib.echo_voice_capture(self._ctx, C.byref(ptr), self._stream()), self._ctx)
        return VoiceHandle(self._lib, ptr, kv_cache_speaker.mask, kv_cache_speaker.batch)

    def bind_voice(self, voice: VoiceHandle) -> KVHandle:
        """Install a captured voice as this context's speaker cache; the returned handle is what `get_kv_cache_speaker` returns."""
        if not voice._ptr:
            raise L.EchoHipError("voice handle was closed")
        L.check(self._lib.echo_voice_bind(self._ctx, voice._ptr, self._stream()), self._ctx)
        return KVHandle(self, "speaker", voice.mask, voice.batch)

    def workspace_bytes(self) -> int:
        return int(self._lib.echo_workspace_bytes(self._ctx))

    def reserve_workspace(self, batch: int, sequence_length: int = 640, text_tokens: int = 768, speaker_latents: int = 6400) -> None:
        """Grow KV caches and workspaces for this request geometry now, so that the first request allocates nothing."""
        L.check(self._lib.echo_reserve_workspace(self._ctx, int(batch), int(sequence_length), int(text_tokens), int(speaker_latents), 0), self._ctx)

    def get_kv_cache_latent(self, prefix_latent: torch.Tensor, n_latents: Optional[int] = None) -> KVHandle:
        """`n_latents`: how many leading prefix latents can ever be attended (start_pos rounded up to the patch);
        the reference encodes the whole zero-padded buffer, which is causal and therefore equivalent."""
        lat = prefix_latent.to(self._device, self._dtype).contiguous()
        B, P, Lz = lat.shape
        ps = self.config.speaker_patch_size
        n = P if n_latents is None else min(P, n_latents)
        n = n // ps * ps
        L.check(self._lib.echo_encode_latent_prefix(self._ctx, lat.data_ptr(), B, n, P * Lz, self._stream()), self._ctx)
        return KVHandle(self, "latent", None, B)

    # ------------------------------------------------------------------ forward (model.py:563-604)
    def forward(self, x: torch.Tensor, t: torch.Tensor, text_mask: torch.Tensor, speaker_mask: torch.Tensor,
                kv_cache_text: KVHandle, kv_cache_speaker: KVHandle, start_pos: Optional[int] = None,
                kv_cache_latent: Optional[KVHandle] = None) -> torch.Tensor:
        rows, S, Lz = x.shape
        B = self._kvB
        if rows % B:
            raise ValueError("batch rows must be a multiple of the KV batch")
        # t (rows,) - any values (model.py:563-604): the distinct timesteps get one modulation table each, row_t maps rows to them
        tt = t.detach().to("cpu").reshape(-1)
        if tt.numel() == 1:
            tt = tt.expand(rows)
        if tt.numel() != rows:
            raise ValueError(f"t must have one entry per row ({rows}), got {tt.numel()}")
        tt = tt.to(self._dtype)                              # the model sees t in its own dtype (inference.py:489, model.py:40)
        uniq: List[float] = []
        row_t: List[int] = []
        for v in tt.float().tolist():
            if v not in uniq:
                uniq.append(v)
            row_t.append(uniq.index(v))
        temb = timestep_embedding(torch.tensor(uniq, dtype=torch.float32).to(self._dtype), self.config.timestep_embed_size)
        temb = temb.to(self._device).contiguous()
        ton = self._row_switch(text_mask, kv_cache_text.mask, rows, B, 1)
        son = self._row_switch(speaker_mask, kv_cache_speaker.mask, rows, kv_cache_speaker.batch, self.config.speaker_patch_size)
        xin = x.to(self._device, self._dtype).contiguous()
        out = torch.empty((rows, S, Lz), dtype=torch.float32, device=self._device)
        L.check(self._lib.echo_dit_forward_t(self._ctx, xin.data_ptr(), temb.data_ptr(), len(uniq), (C.c_int32 * rows)(*row_t), rows, B, S,
                                             int(start_pos or 0), int(kv_cache_latent is not None), (C.c_int32 * rows)(*ton),
                                             (C.c_int32 * rows)(*son), out.data_ptr(), self._stream()), self._ctx)
        return out

    __call__ = forward

    @staticmethod
    def _row_switch(mask: torch.Tensor, cached: Optional[torch.Tensor], rows: int, B: int, stride: int) -> List[int]:
        """Each row's mask must equal the mask the cache was built with (segment on) or be all False (segment off)."""
        m = mask.detach().to("cpu", torch.bool)[..., ::stride]
        on = []
        for r in range(rows):
            if not bool(m[r].any()):
                on.append(0)
            elif cached is None or torch.equal(m[r], cached[r % B]):
                on.append(1)
            else:
                raise NotImplementedError("per-row key masks other than the cached mask or all-False are not supported")
        return on
