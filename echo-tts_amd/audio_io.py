"""`load_audio` of the reference (inference.py:104-113) and the WAV leg of its output path (handler.py:482-535), without its codec stack.

The reference decodes any container with torchcodec, averages the channels, resamples to 44.1 kHz with
`torchaudio.functional.resample` and scales the peak down to 1.  Neither library exists in this image, so:

* decoding: RIFF/WAVE files (PCM 8 / 16 / 24 / 32 bit, IEEE float 32 / 64, WAVE_FORMAT_EXTENSIBLE) are parsed here; for any other
  container `load_audio` uses torchcodec when it is importable and otherwise fails loudly - there is no silent fallback;
* resampling: torchaudio's published `sinc_interp_hann` algorithm (its documentation and `_get_sinc_resample_kernel`: a bank of
  `new_freq / gcd` windowed-sinc filters of half-width ceil(lowpass_filter_width * orig / (min(orig, new) * rolloff)), applied with
  stride `orig_freq / gcd`) is restated in `sinc_resample_bank`; the filter bank is evaluated on the host in float64 and applied on
  the MI355X by `echo_op_resample` (csrc/postproc.hip).  PARITY UNPINNED: torchaudio is absent, so no reference output exists to
  compare with; tests check the kernel against a float64 evaluation of the same formula and against analytic sine waves;
* output: `wav_bytes` writes the lossless PCM-16 WAV the reference produces first (handler.py:509); its Opus re-encode through
  ffmpeg (handler.py:517-535) needs a codec this image does not have and stays out of scope.
"""
from __future__ import annotations

import math
import struct
from typing import Optional, Tuple

import torch

from . import _lib as L

TARGET_SR = 44_100


def read_wav(path: str, max_duration: Optional[float] = None) -> Tuple[torch.Tensor, int]:
    """(channels, samples) fp32 in [-1, 1] and the sample rate of a RIFF/WAVE file."""
    with open(path, "rb") as f:
        data = f.read()
    if len(data) < 12 or data[:4] != b"RIFF" or data[8:12] != b"WAVE":
        raise ValueError(f"{path}: not a RIFF/WAVE file")
    pos, fmt, pcm = 12, None, None
    while pos + 8 <= len(data):
        cid, size = data[pos:pos + 4], struct.unpack("<I", data[pos + 4:pos + 8])[0]
        body = data[pos + 8:pos + 8 + size]
        if cid == b"fmt ":
            tag, ch, sr, _, _, bits = struct.unpack("<HHIIHH", body[:16])
            if tag == 0xFFFE and len(body) >= 26:          # WAVE_FORMAT_EXTENSIBLE: the real tag is the first 2 bytes of the sub-format GUID
                tag = struct.unpack("<H", body[24:26])[0]
            fmt = (tag, ch, sr, bits)
        elif cid == b"data":
            pcm = body
        pos += 8 + size + (size & 1)
    if fmt is None or pcm is None:
        raise ValueError(f"{path}: missing fmt / data chunk")
    tag, ch, sr, bits = fmt
    if ch < 1 or sr < 1:
        raise ValueError(f"{path}: bad channel count / sample rate")
    frame = ch * bits // 8
    n = len(pcm) // frame
    if max_duration is not None:
        n = min(n, int(max_duration * sr))
    raw = torch.frombuffer(bytearray(pcm[: n * frame]), dtype=torch.uint8)
    if tag == 1 and bits == 8:
        x = (raw.float() - 128.0) / 128.0
    elif tag == 1 and bits == 16:
        x = raw.view(torch.int16).float() / 32768.0
    elif tag == 1 and bits == 24:
        b = raw.view(-1, 3).to(torch.int32)
        v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
        x = torch.where(v >= 1 << 23, v - (1 << 24), v).float() / float(1 << 23)
    elif tag == 1 and bits == 32:
        x = raw.view(torch.int32).double().div(2147483648.0).float()
    elif tag == 3 and bits == 32:
        x = raw.view(torch.float32).clone()
    elif tag == 3 and bits == 64:
        x = raw.view(torch.float64).float()
    else:
        raise ValueError(f"{path}: unsupported WAVE format tag {tag} with {bits} bits")
    return x.view(n, ch).t().contiguous(), sr


def sinc_resample_bank(orig_freq: int, new_freq: int, lowpass_filter_width: int = 6, rolloff: float = 0.99) -> Tuple[torch.Tensor, int, int, int]:
    """The `sinc_interp_hann` filter bank of torchaudio.functional.resample (its defaults), restated from the published algorithm:
    returns (bank (up, taps) fp32, up = new / gcd, down = orig / gcd, width).  Output phase p of frame f is
    sum_k bank[p][k] * x[f * down + k - width]."""
    if orig_freq <= 0 or new_freq <= 0:
        raise ValueError("sample rates must be positive")
    g = math.gcd(int(orig_freq), int(new_freq))
    down, up = int(orig_freq) // g, int(new_freq) // g
    base = min(down, up) * rolloff                                     # cut-off below the lower Nyquist frequency
    width = int(math.ceil(lowpass_filter_width * down / base))
    idx = torch.arange(-width, width + down, dtype=torch.float64) / down       # input sample times around the frame, in input periods
    t = torch.arange(0, -up, -1, dtype=torch.float64)[:, None] / up + idx[None, :]
    t = (t * base).clamp(-lowpass_filter_width, lowpass_filter_width)
    window = torch.cos(t * math.pi / lowpass_filter_width / 2) ** 2    # Hann window over the clamped span
    t = t * math.pi
    sinc = torch.where(t == 0, torch.ones_like(t), torch.sin(t) / t)
    bank = sinc * window * (base / down)
    return bank.float().contiguous(), up, down, width


@torch.inference_mode()
def resample(audio: torch.Tensor, orig_freq: int, new_freq: int, device="cuda:0") -> torch.Tensor:
    """(..., n) -> (..., ceil(new * n / orig)) like torchaudio.functional.resample with its default arguments; identity when the
    rates agree.  Runs on the MI355X (one launch per signal row)."""
    if int(orig_freq) == int(new_freq):
        return audio
    dev = torch.device(device)
    if dev.type != "cuda":
        raise L.EchoHipError("resample runs on the MI355X; there is no CPU path")
    lib = L.load_library()
    bank, up, down, width = sinc_resample_bank(orig_freq, new_freq)
    bank = bank.to(dev)
    shape = audio.shape
    x = audio.reshape(-1, shape[-1]).to(dev, torch.float32).contiguous()
    n = x.shape[1]
    frames = n // down + 1                                              # the zero-extended signal holds this many whole input frames
    target = int(math.ceil(up * n / down))
    out = torch.empty((x.shape[0], frames * up), dtype=torch.float32, device=dev)
    st = torch.cuda.current_stream(dev).cuda_stream
    for r in range(x.shape[0]):
        L.check(lib.echo_op_resample(x[r].data_ptr(), n, bank.data_ptr(), bank.shape[1], up, down, width, out[r].data_ptr(), out.shape[1], st))
    return out[:, :target].reshape(*shape[:-1], target)


@torch.inference_mode()
def load_audio(path: str, max_duration: int = 300, device="cuda:0") -> torch.Tensor:
    """reference inference.py:104-113: decode (at most `max_duration` seconds), average the channels, resample to 44.1 kHz, divide
    by max(peak, 1).  Returns (1, n) fp32 on `device`."""
    try:
        audio, sr = read_wav(path, max_duration)
    except ValueError:
        try:
            from torchcodec.decoders import AudioDecoder          # the reference's decoder, when the deployment has it
        except Exception as e:
            raise L.EchoHipError(f"{path}: only RIFF/WAVE files can be decoded without torchcodec ({type(e).__name__}: {e})")
        dec = AudioDecoder(path)
        sr = dec.metadata.sample_rate
        audio = dec.get_samples_played_in_range(0, max_duration).data
    audio = audio.float().mean(dim=0).unsqueeze(0)
    audio = resample(audio, sr, TARGET_SR, device=device).to(device)
    return audio / torch.maximum(audio.abs().max(), torch.tensor(1.0, device=audio.device))


def wav_bytes(audio: torch.Tensor, sample_rate: int = TARGET_SR) -> bytes:
    """(channels, n) or (n,) fp32 in [-1, 1] -> a PCM-16 RIFF/WAVE file image: the lossless intermediate of handler.py:509."""
    a = audio.detach().float().cpu()
    if a.dim() == 1:
        a = a.unsqueeze(0)
    ch, n = a.shape
    pcm = (a.clamp(-1.0, 1.0) * 32767.0).round().to(torch.int16).t().contiguous().numpy().tobytes()
    hdr = b"RIFF" + struct.pack("<I", 36 + len(pcm)) + b"WAVE" + b"fmt " + struct.pack("<IHHIIHH", 16, 1, ch, sample_rate, sample_rate * ch * 2, ch * 2, 16)
    return hdr + b"data" + struct.pack("<I", len(pcm)) + pcm
