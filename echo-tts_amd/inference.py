"""Host side of the Echo-TTS pipeline with the reference's call signatures (inference.py of
sruckh/echo-tts); the sampler loop, the EchoDiT and the Fish S1-DAC decoder run in libechohip.

Kept verbatim from the reference API (names, argument order, defaults, return values):
`tokenizer_encode`, `chunk_text`, `get_text_input_ids_and_mask`, `PCAState`, `ae_decode`,
`find_flattening_point`, `crop_audio_to_flattening_point`, `sample_pipeline`,
`sample_pipeline_chunked`, `sample_euler_cfg_independent_guidances`, `_temporal_score_rescale`.
The string/host helpers are re-implemented here (they are not on the GPU hot path) and pinned by
known-answer tests generated from the reference (tests/golden/meta.json).
"""
from __future__ import annotations

import ctypes as C
import re
from dataclasses import dataclass
from typing import Callable, List, Optional, Sequence, Tuple

import torch

from . import _lib as L
from .autoencoder import DAC
from .model import EchoDiT, KVHandle, timestep_embedding

# --------------------------------------------------------------------------- text front end
_REPLACEMENTS = (("…", "..."), ("’", "'"), ("”", '"'), ("\n", " "), (":", ","), (";", ","), ("—", ", "))


def tokenizer_encode(text: str, append_bos: bool = True, normalize: bool = True, return_normalized_text: bool = False):
    """UTF-8 bytes with BOS=0 (reference inference.py:115-136).  Normalisation maps a few punctuation marks and
    prefixes "[S1] " unless the text opens with '[' / '(' or mentions S1/S2 anywhere."""
    if normalize:
        for src, dst in _REPLACEMENTS:
            text = text.replace(src, dst)
        opens_tagged = text.startswith("[") or text.startswith("(")
        if not opens_tagged and "S1" not in text and "S2" not in text:
            text = "[S1] " + text
    data = list(text.encode("utf-8"))
    if append_bos:
        data = [0] + data
    ids = torch.tensor(data)
    return (ids, text) if return_normalized_text else ids


_WS = re.compile(r"\s+")
_SENT_END = frozenset(".!?")
_CLAUSE_END = frozenset(",;:")
_CLOSERS = frozenset("\"')]}”’")


def chunk_text(text: str, max_chars: int = 300) -> List[str]:
    """Greedy split into pieces of at most `max_chars`, cutting at the LAST sentence end inside the window, else the
    last clause end, else the last whitespace, else hard at max_chars (reference inference.py:140-190)."""
    if max_chars <= 0:
        raise ValueError("max_chars must be > 0")
    rest = _WS.sub(" ", text or "").strip()
    if not rest:
        return []
    out: List[str] = []
    while rest:
        if len(rest) <= max_chars:
            out.append(rest)
            break
        win = rest[: max_chars + 1]
        cut_sentence = cut_clause = cut_space = None
        for pos in range(1, len(win)):
            if not win[pos].isspace():
                continue
            cut_space = pos
            before = win[pos - 1]
            before2 = win[pos - 2] if pos >= 2 else ""
            closes = before in _CLOSERS
            if before in _SENT_END or (closes and before2 in _SENT_END):
                cut_sentence = pos
            elif before in _CLAUSE_END or (closes and before2 in _CLAUSE_END):
                cut_clause = pos
        cut = cut_sentence or cut_clause or cut_space or max_chars
        piece = rest[:cut].strip()
        if piece:
            out.append(piece)
        rest = rest[cut:].strip()
    return out


def get_text_input_ids_and_mask(text_arr: List[str], max_length: Optional[int], device=None, normalize: bool = True,
                                return_normalized_text: bool = False, pad_to_max: bool = True):
    """(B, max_length) int32 ids + bool mask of the valid prefix (reference inference.py:192-214).  As in the reference,
    `pad_to_max=False` does not trim: with max_length given the result always has max_length columns."""
    enc = [tokenizer_encode(t, normalize=normalize, return_normalized_text=True) for t in text_arr]
    if max_length is None:
        max_length = max(len(e) for e, _ in enc)
    ids = torch.zeros((len(text_arr), max_length), dtype=torch.int32)
    mask = torch.zeros((len(text_arr), max_length), dtype=torch.bool)
    for row, (e, _) in enumerate(enc):
        n = min(len(e), max_length)
        ids[row, :n] = e[:n]
        mask[row, :n] = True
    if device is not None:
        ids, mask = ids.to(device), mask.to(device)
    if return_normalized_text:
        return ids, mask, [t for _, t in enc]
    return ids, mask


# --------------------------------------------------------------------------- audio input (reference inference.py:104-113)
from .audio_io import load_audio  # noqa: E402,F401  (same name and arguments as the reference)


# --------------------------------------------------------------------------- autoencoder glue
@dataclass
class PCAState:
    pca_components: torch.Tensor
    pca_mean: torch.Tensor
    latent_scale: float


@torch.inference_mode()
def ae_decode(fish_ae: DAC, pca_state: PCAState, z_q: torch.Tensor) -> torch.Tensor:
    """reference inference.py:226-229: PCA inverse + DAC.decode_zq, (B, T, 80) fp32 -> (B, 1, T*2048) fp32."""
    fish_ae.set_pca(pca_state)
    return fish_ae.decode_latent(z_q)


@torch.inference_mode()
def ae_encode(fish_ae: DAC, pca_state: PCAState, audio: torch.Tensor) -> torch.Tensor:
    """reference inference.py:218-224: (B, 1, length) audio -> (B, T, 80) latents (DAC.encode_zq + PCA projection)."""
    assert audio.ndim == 3 and audio.shape[1] == 1
    return fish_ae.encode_latent(audio, pca_state)


@torch.inference_mode()
def ae_reconstruct(fish_ae: DAC, pca_state: PCAState, audio: torch.Tensor) -> torch.Tensor:
    """reference inference.py:231-235."""
    return ae_decode(fish_ae, pca_state, ae_encode(fish_ae, pca_state, audio))


@torch.inference_mode()
def get_speaker_latent_and_mask(fish_ae: DAC, pca_state: PCAState, audio: torch.Tensor, max_speaker_latent_length: int = 6400,
                                audio_chunk_size: int = 640 * 2048, pad_to_max: bool = False,
                                divis_by_patch_size: int | None = 4) -> Tuple[torch.Tensor, torch.Tensor]:
    """reference inference.py:239-283: (1, length) audio -> (speaker_latent (1, T, 80), mask (1, T)); the audio is encoded in
    chunks of `audio_chunk_size` samples (each zero padded to a whole chunk, as in training), then trimmed to its true length."""
    factor = 2048
    assert audio.ndim == 2 and audio.shape[0] == 1
    audio = audio[:, : max_speaker_latent_length * factor]
    lat = []
    for i in range(0, audio.shape[1], audio_chunk_size):
        chunk = audio[:, i:i + audio_chunk_size]
        if chunk.shape[1] < audio_chunk_size:
            chunk = torch.nn.functional.pad(chunk, (0, audio_chunk_size - chunk.shape[1]))
        lat.append(ae_encode(fish_ae, pca_state, chunk.unsqueeze(0)))
    speaker_latent = torch.cat(lat, dim=1)
    actual = audio.shape[1] // factor
    mask = (torch.arange(speaker_latent.shape[1], device=speaker_latent.device) < actual).unsqueeze(0)
    if pad_to_max and speaker_latent.shape[1] < max_speaker_latent_length:
        speaker_latent = torch.nn.functional.pad(speaker_latent, (0, 0, 0, max_speaker_latent_length - speaker_latent.shape[1]))
        mask = torch.nn.functional.pad(mask, (0, max_speaker_latent_length - mask.shape[1]))
    elif not pad_to_max:
        speaker_latent, mask = speaker_latent[:, :actual], mask[:, :actual]
    if divis_by_patch_size is not None:
        n = speaker_latent.shape[1] // divis_by_patch_size * divis_by_patch_size
        speaker_latent, mask = speaker_latent[:, :n], mask[:, :n]
    return speaker_latent, mask


def find_flattening_points(latents: torch.Tensor, target_value: float = 0.0, window_size: int = 20, std_threshold: float = 0.05) -> List[int]:
    """reference inference.py:288-296 for a batch (B, T, W) of device latents: ONE kernel launch (csrc/postproc.hip) and one
    B-int read-back instead of up to 640 iterations x 2 host syncs per utterance."""
    if latents.dim() == 2:
        latents = latents.unsqueeze(0)
    x = latents.detach().to(torch.float32).contiguous()
    if not x.is_cuda:
        raise L.EchoHipError("find_flattening_points needs device latents (the host version is find_flattening_point)")
    B, T = x.shape[0], x.shape[1]
    W = x[0, 0].numel()
    out = torch.empty((B,), dtype=torch.int32, device=x.device)
    L.check(L.load_library().echo_op_find_flattening_point(x.data_ptr(), T * W, B, T, W, int(window_size), float(target_value),
                                                           float(std_threshold), out.data_ptr(), torch.cuda.current_stream(x.device).cuda_stream))
    return [int(v) for v in out.tolist()]


def find_flattening_point(data: torch.Tensor, target_value: float = 0.0, window_size: int = 20, std_threshold: float = 0.05) -> int:
    """First frame i whose next `window_size` frames (zero padded) have std < threshold and |mean - target| < 0.1
    (reference inference.py:288-296).  Device latents go through the HIP kernel (find_flattening_points); host tensors
    (tests, tools) are evaluated with one cumulative-sum pass instead of the reference's per-frame loop."""
    if data.is_cuda:
        return find_flattening_points(data.unsqueeze(0), target_value, window_size, std_threshold)[0]
    x = data.detach().to("cpu", torch.float64)
    n, width = x.shape[0], x[0].numel()
    flat = torch.cat([x.reshape(n, -1), torch.zeros(window_size, width, dtype=torch.float64)])
    rs = torch.cat([torch.zeros(1, dtype=torch.float64), flat.sum(1).cumsum(0)])
    rq = torch.cat([torch.zeros(1, dtype=torch.float64), (flat * flat).sum(1).cumsum(0)])
    cnt = window_size * width
    s = rs[window_size:] - rs[:-window_size]
    q = rq[window_size:] - rq[:-window_size]
    mean = s / cnt
    var = (q - s * s / cnt) / (cnt - 1)
    ok = (var.clamp_min(0).sqrt() < std_threshold) & ((mean - target_value).abs() < 0.1)
    idx = torch.nonzero(ok[:n]).flatten()
    return int(idx[0]) if idx.numel() else n


def crop_audio_to_flattening_point(audio: torch.Tensor, latent: torch.Tensor) -> torch.Tensor:
    return audio[..., : find_flattening_point(latent) * 2048]


# --------------------------------------------------------------------------- sampler
SampleFn = Callable[[EchoDiT, torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor, int], torch.Tensor]
KVCache = KVHandle


def _concat_kv_caches(*caches: KVHandle) -> KVHandle:
    """reference inference.py:398-406; the HIP engine shares one copy (stride 0), so this only records the repeat."""
    return caches[0].repeated(len(caches))


def _multiply_kv_cache(cache: KVHandle, scale: float, max_layers: Optional[int] = None) -> None:
    """reference inference.py:408-414 (speaker cache only)."""
    m = cache.model
    L.check(m._lib.echo_scale_speaker_kv(m._ctx, float(scale), -1 if max_layers is None else int(max_layers), m._stream()), m._ctx)


def _temporal_score_rescale(v_pred, x_t, t, rescale_k: float, rescale_sigma: float):
    """reference inference.py:416-424 (https://arxiv.org/pdf/2510.01184)."""
    if t < 1:
        snr = (1 - t) ** 2 / (t ** 2)
        ratio = (snr * rescale_sigma ** 2 + 1) / (snr * rescale_sigma ** 2 / rescale_k + 1)
        return 1 / (1 - t) * (ratio * ((1 - t) * v_pred + x_t) - x_t)
    return v_pred


def build_schedule(model: EchoDiT, num_steps: int, cfg_min_t: float, cfg_max_t: float, rescale_k, rescale_sigma,
                   speaker_kv_scale, speaker_kv_min_t):
    """Everything the step loop needs that depends only on the schedule, evaluated with the
    reference's own fp32 tensor expressions on the host: t_i, has_cfg_i, dt_i, rescale coefficients, kv un-scale step,
    and the timestep embeddings of bf16/fp32 t_i."""
    ts = torch.linspace(1.0, 0.0, num_steps + 1) * 0.999          # inference.py:452,459
    steps = (L.EchoStep * num_steps)()
    tvals = []
    for i in range(num_steps):
        t, tn = ts[i], ts[i + 1]
        st = steps[i]
        st.has_cfg = int(((t >= cfg_min_t) * (t <= cfg_max_t)).item())
        st.dt = float((tn - t).item())
        if rescale_k is not None and rescale_sigma is not None and bool(t < 1):
            snr = (1 - t) ** 2 / (t ** 2)
            ratio = (snr * rescale_sigma ** 2 + 1) / (snr * rescale_sigma ** 2 / rescale_k + 1)
            st.rescale, st.r_inv1mt, st.r_ratio, st.r_1mt = 1, float((1 / (1 - t)).item()), float(ratio.item()), float((1 - t).item())
        st.kv_unscale_after = int(speaker_kv_scale is not None and bool(tn < speaker_kv_min_t) and bool(t >= speaker_kv_min_t))
        tvals.append((torch.ones((1,)) * t).to(model.dtype))          # inference.py:489,499
    temb = timestep_embedding(torch.cat(tvals), model.config.timestep_embed_size)
    return steps, temb


def run_euler(model: EchoDiT, x_init: torch.Tensor, steps, temb: torch.Tensor, cfg_scale_text: float, cfg_scale_speaker: float,
              truncation_factor, speaker_kv_scale, speaker_kv_max_layers, start_pos: int = 0, use_latent: bool = False) -> torch.Tensor:
    """One echo_sample_euler call: x_init (B,S,L) fp32 on the model device -> latents (B,S,L) fp32."""
    B, S, _ = x_init.shape
    p = L.EchoSamplerParams()
    p.B, p.S, p.num_steps = B, S, len(steps)
    p.start_pos, p.use_latent = int(start_pos), int(use_latent)
    p.cfg_scale_text, p.cfg_scale_speaker = float(cfg_scale_text), float(cfg_scale_speaker)
    p.has_truncation = int(truncation_factor is not None)        # inference.py:478 `if truncation_factor is not None`
    p.init_scale = 1.0 if truncation_factor is None else float(truncation_factor)
    p.kv_scale = 1.0 if speaker_kv_scale is None else float(speaker_kv_scale)
    p.kv_max_layers = -1 if speaker_kv_max_layers is None else int(speaker_kv_max_layers)
    p.steps = steps
    temb_dev = temb.to(model.device).contiguous()
    p.temb = temb_dev.data_ptr()
    x0 = x_init.to(model.device, torch.float32).contiguous()
    out = torch.empty_like(x0)
    L.check(model._lib.echo_sample_euler(model._ctx, C.byref(p), x0.data_ptr(), out.data_ptr(), model._stream()), model._ctx)
    return out


@torch.inference_mode()
def sample_euler_cfg_independent_guidances(
    model: EchoDiT,
    speaker_latent: torch.Tensor,
    speaker_mask: torch.Tensor,
    text_input_ids: torch.Tensor,
    text_mask: torch.Tensor,
    rng_seed: int,
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
    sequence_length: int | None = None,
    x_init: torch.Tensor | None = None,
    speaker_kv=None,
) -> torch.Tensor:
    """Drop-in for reference inference.py:427-517.  Extensions: `x_init` replaces the RNG draw (fixtures); `speaker_kv`
    (a model.VoiceHandle from the per-voice cache) is bound instead of re-running get_kv_cache_speaker, `speaker_latent` is
    then ignored.  A speaker latent with batch 1 next to B text rows is one voice shared by every row (the handler's chunks of
    one request, handler.py:747-759): its KV is encoded once and addressed with stride 0."""
    if sequence_length is None:
        sequence_length = 640
    device = model.device
    B = text_input_ids.shape[0]
    steps, temb = build_schedule(model, num_steps, cfg_min_t, cfg_max_t, rescale_k, rescale_sigma, speaker_kv_scale, speaker_kv_min_t)
    model.get_kv_cache_text(text_input_ids, text_mask)
    kv_spk = model.bind_voice(speaker_kv) if speaker_kv is not None else model.get_kv_cache_speaker(speaker_latent, speaker_mask)
    if speaker_kv_scale is not None:
        _multiply_kv_cache(kv_spk, speaker_kv_scale, speaker_kv_max_layers)
    if x_init is None:
        rng = torch.Generator(device=device).manual_seed(rng_seed)                                   # inference.py:457
        x_init = torch.randn((B, sequence_length, model.config.latent_size), device=device, dtype=torch.float32, generator=rng)
    return run_euler(model, x_init, steps, temb, cfg_scale_text, cfg_scale_speaker, truncation_factor, speaker_kv_scale,
                     speaker_kv_max_layers)


# --------------------------------------------------------------------------- pipelines (reference inference.py:308-388)
@torch.inference_mode()
def sample_pipeline(
    model: EchoDiT,
    fish_ae: DAC,
    pca_state: PCAState,
    sample_fn: SampleFn,
    text_prompt: str,
    speaker_audio: torch.Tensor | None,
    rng_seed: int,
    pad_to_max_speaker_latent_length: int | None = None,
    pad_to_max_text_length: int | None = None,
    normalize_text: bool = True,
    speaker_latent: torch.Tensor | None = None,
    speaker_mask: torch.Tensor | None = None,
) -> Tuple[torch.Tensor, str]:
    """`speaker_audio` (1, length) goes through the HIP DAC encoder (get_speaker_latent_and_mask) like the reference;
    `speaker_latent`/`speaker_mask` (extension) pass precomputed reference-voice latents instead (per-voice cache)."""
    MAX_TEXT_LENGTH = 768
    device, dtype = model.device, model.dtype
    ids, tmask, norm = get_text_input_ids_and_mask(
        [text_prompt], max_length=min(pad_to_max_text_length or MAX_TEXT_LENGTH, MAX_TEXT_LENGTH), device=device,
        normalize=normalize_text, return_normalized_text=True, pad_to_max=(pad_to_max_text_length is not None))
    if speaker_latent is None:
        if speaker_audio is not None:       # reference inference.py:333-340
            speaker_latent, speaker_mask = get_speaker_latent_and_mask(
                fish_ae, pca_state, speaker_audio.to(device), max_speaker_latent_length=pad_to_max_speaker_latent_length or 6400,
                pad_to_max=pad_to_max_speaker_latent_length is not None)
            speaker_latent = speaker_latent.to(dtype)
        else:
            n = pad_to_max_speaker_latent_length or 4
            speaker_latent = torch.zeros((1, n, model.config.latent_size), device=device, dtype=dtype)
            speaker_mask = torch.zeros((1, n), device=device, dtype=torch.bool)
    latent_out = sample_fn(model, speaker_latent, speaker_mask, ids, tmask, rng_seed)
    audio_out = ae_decode(fish_ae, pca_state, latent_out)
    audio_out = crop_audio_to_flattening_point(audio_out, latent_out[0])
    return audio_out, norm[0]


@torch.inference_mode()
def sample_pipeline_chunked(
    model: EchoDiT,
    fish_ae: DAC,
    pca_state: PCAState,
    sample_fn: SampleFn,
    text_prompt: str,
    speaker_audio: torch.Tensor | None,
    rng_seed: int,
    *,
    max_chars_per_chunk: int = 300,
    pad_to_max_speaker_latent_length: int | None = None,
    pad_to_max_text_length: int | None = None,
    normalize_text: bool = True,
    speaker_latent: torch.Tensor | None = None,
    speaker_mask: torch.Tensor | None = None,
) -> Tuple[torch.Tensor, str]:
    pieces = chunk_text(text_prompt, max_chars=max_chars_per_chunk)
    if not pieces:
        raise ValueError("text_prompt is empty after normalization")
    audio, texts = [], []
    for k, piece in enumerate(pieces):
        a, t = sample_pipeline(model, fish_ae, pca_state, sample_fn, piece, speaker_audio, rng_seed + k,
                               pad_to_max_speaker_latent_length=pad_to_max_speaker_latent_length,
                               pad_to_max_text_length=pad_to_max_text_length, normalize_text=normalize_text,
                               speaker_latent=speaker_latent, speaker_mask=speaker_mask)
        audio.append(a)
        texts.append(t)
    return torch.cat(audio, dim=-1), "\n".join(texts)


# --------------------------------------------------------------------------- loading from LOCAL files
def load_model_from_path(path: str, device: str = "cuda", dtype: torch.dtype | None = torch.bfloat16,
                         delete_blockwise_modules: bool = False, config=None, fp8: bool = False) -> EchoDiT:
    """Local-file counterpart of load_model_from_hf (reference inference.py:14-47): same safetensors key layout."""
    import safetensors.torch as st
    from .model import EchoDiTConfig
    state = st.load_file(path, device="cpu")
    if delete_blockwise_modules:
        state = {k: v for k, v in state.items() if not (k.startswith("latent_encoder.") or k.startswith("latent_norm")
                                                        or ".wk_latent" in k or ".wv_latent" in k)}
    return EchoDiT(config or EchoDiTConfig(), state, dtype=dtype or torch.bfloat16, device="cuda:0" if device == "cuda" else device,
                   fp8=fp8)


def load_fish_ae_from_path(path: str, device: str = "cuda", dtype: torch.dtype | None = torch.float32, config=None) -> DAC:
    import safetensors.torch as st
    from .autoencoder import DACConfig
    return DAC(config or DACConfig(), st.load_file(path, device="cpu"), device="cuda:0" if device == "cuda" else device)


def load_pca_state_from_path(path: str, device: str = "cuda") -> PCAState:
    import safetensors.torch as st
    t = st.load_file(path, device="cpu")
    return PCAState(pca_components=t["pca_components"], pca_mean=t["pca_mean"], latent_scale=float(t["latent_scale"].item()))
