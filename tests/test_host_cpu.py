"""CPU: host-side logic of the product package (no GPU, no compute through the HIP library)."""
import ctypes
import json
import os
import re

import pytest
import torch

import echo_tts_amd as E
from echo_tts_amd import _lib as L
from echo_tts_amd import inference as inf

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_loads_and_exports_every_declared_symbol():
    lib = L.load_library()
    header = open(os.path.join(ROOT, "include", "echo_hip.h"), encoding="utf-8").read()
    declared = set(re.findall(r"\b(echo_[a-z0-9_]+)\s*\(", header)) - {"echo_ctx"}
    assert declared, "no declarations parsed"
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} is declared in include/echo_hip.h but not exported"
        assert name in L.SIGNATURES, f"{name} has no ctypes signature"
    assert lib.echo_abi_version() == L.ABI_VERSION


def test_struct_sizes_match_the_header():
    # sizes computed from the C declarations (ints / floats / pointers / int64), guarding field drift
    assert ctypes.sizeof(L.EchoStep) == 7 * 4
    assert ctypes.sizeof(L.EchoConfig) == 4 * (1 + 5 + 1 + 5 + 5 + 2 + 1 + 3 + 8 + 5 + 1 + 4 + 1 + (2 + 8 + 8 + 1) + 4 + 1)   # ... + dit_fp8
    assert ctypes.sizeof(L.EchoSamplerParams) == 11 * 4 + 4 + 2 * 8      # 11 ints / floats (ABI 7: + has_truncation), padding to 8, 2 pointers


def test_product_fails_loudly_without_gpu():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(Exception):
        E.EchoDiT(E.EchoDiTConfig(), {}, device="cpu")
    cfg = L.EchoConfig()
    ctx = ctypes.c_void_p()
    assert L.load_library().echo_ctx_create(ctypes.byref(cfg), 0, ctypes.byref(ctx)) != 0
    assert b"no HIP device" in L.load_library().echo_last_error(None)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "echo-tts_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, fn), encoding="utf-8").read()
                assert "oracle" not in src.replace("# oracle", ""), f"{fn} mentions the oracle"
    # tools/ are product-side utilities too; scripts that need the checker live under tests/ (make_goldens.py, bench_encode.py)
    for fn in os.listdir(os.path.join(ROOT, "tools")):
        if fn.endswith((".py", ".sh")):
            src = open(os.path.join(ROOT, "tools", fn), encoding="utf-8").read()
            assert "from oracle" not in src and "import oracle" not in src, f"tools/{fn} imports the oracle"
    # bench.py may use it in its two reported baseline legs only (cpu_baseline, eager_gpu_baseline): a single import site, _baseline_ops()
    bsrc = open(os.path.join(ROOT, "bench.py"), encoding="utf-8").read()
    assert bsrc.count("from oracle") + bsrc.count("import oracle") == 1
    assert bsrc.index("def _baseline_ops") < bsrc.index("from oracle") < bsrc.index("def cpu_baseline")
    assert bsrc.count("_baseline_ops()") == 3          # its definition and the two baseline legs


def test_tokenizer_and_masks_kat(golden):
    host = golden["__meta__"]["host"]
    for case in host["tokenizer"]:
        ids, norm = inf.tokenizer_encode(case["text"], return_normalized_text=True)
        assert ids.tolist() == case["ids"]
        assert norm == case["normalized"]
    c0, c1 = host["ids_mask"]
    ids, mask, norm = inf.get_text_input_ids_and_mask(c0["texts"], max_length=768, return_normalized_text=True, pad_to_max=False)
    assert list(ids.shape) == c0["shape"] == [2, 768] and ids.dtype == torch.int32 and mask.dtype == torch.bool
    assert mask.sum(1).tolist() == c0["valid"] and ids[:, :24].tolist() == c0["first"] and norm == c0["normalized"]
    ids, mask = inf.get_text_input_ids_and_mask(c1["texts"], max_length=None)
    assert list(ids.shape) == c1["shape"] and mask.sum(1).tolist() == c1["valid"] and ids[:, :24].tolist() == c1["first"]


def test_chunk_text_kat(golden):
    for case in golden["__meta__"]["host"]["chunk_text"]:
        assert inf.chunk_text(case["text"], case["max_chars"]) == case["chunks"]
    with pytest.raises(ValueError):
        inf.chunk_text("abc", 0)


def test_flattening_point_kat(golden):
    want = golden["__meta__"]["host"]["flattening"]
    assert [inf.find_flattening_point(golden[f"flat.{i}"]) for i in range(3)] == want
    lat = torch.zeros((30, 80))
    assert inf.find_flattening_point(lat) == 0
    audio = torch.zeros((1, 1, 30 * 2048))
    assert inf.crop_audio_to_flattening_point(audio, torch.ones((30, 80)) * 3).shape[-1] == 30 * 2048


def test_schedule_matches_reference_facts():
    class M:  # build_schedule only needs dtype and the embed size
        dtype = torch.bfloat16
        config = E.EchoDiTConfig()
    steps, temb = inf.build_schedule(M, 40, 0.5, 1.0, None, None, None, None)
    assert sum(s.has_cfg for s in steps) == 20                      # 20 CFG steps of 40
    assert temb.shape == (40, 512) and temb.dtype == torch.bfloat16
    assert abs(sum(s.dt for s in steps) + 0.999) < 1e-6
    # bf16(0.999) == 1.0: the first embedding equals the embedding of t = 1
    from echo_tts_amd.model import timestep_embedding
    assert torch.equal(temb[0], timestep_embedding(torch.ones(1).bfloat16(), 512)[0])
    steps, _ = inf.build_schedule(M, 6, 0.4, 0.9, 1.2, 3.0, 1.5, 0.6)
    assert [s.kv_unscale_after for s in steps].count(1) == 1 and all(s.rescale for s in steps)


def test_conv_weight_reshapes_are_consistent():
    """The GEMM forms used by the HIP taps loop equal the convolutions they replace (checked with torch on CPU)."""
    from echo_tts_amd.autoencoder import _conv_as_gemm, _convT_as_gemm
    g = torch.Generator().manual_seed(0)
    x = torch.randn((1, 6, 20), generator=g, dtype=torch.float64)
    w = torch.randn((5, 6, 7), generator=g, dtype=torch.float64)
    for dil in (1, 3):
        ref = torch.nn.functional.conv1d(torch.nn.functional.pad(x, (6 * dil, 0)), w, dilation=dil)[0].T
        xp = torch.cat([torch.zeros((6 * dil, 6), dtype=torch.float64), x[0].T])
        rows = torch.stack([torch.cat([xp[t + j * dil] for j in range(7)]) for t in range(20)])
        assert torch.allclose(rows @ _conv_as_gemm(w).T, ref)
    wt = torch.randn((6, 4, 8), generator=g, dtype=torch.float64)
    ref = torch.nn.functional.conv_transpose1d(x, wt, stride=4)[0, :, :80].T
    xp = torch.cat([torch.zeros((1, 6), dtype=torch.float64), x[0].T])
    rows = torch.stack([torch.cat([xp[q], xp[q + 1]]) for q in range(20)])
    out = (rows @ _convT_as_gemm(wt, 4).T).reshape(80, 4)
    assert torch.allclose(out, ref)


def test_handler_helpers_kat(golden):
    from echo_tts_amd import handler as H
    host = golden["__meta__"]["host"]
    for case in host["chunk_text_for_audio"]:
        assert H.chunk_text_for_audio(case["text"], case["max_chars"], case["dur"]) == case["chunks"]
    a, b, c = golden["post.a"], golden["post.b"], golden["post.c"]
    assert torch.equal(H.crossfade_chunks([a, b, c], 4410), golden["post.crossfade"])
    assert torch.equal(H.normalize_chunk_boundaries([a, b, c], min_silence_samples=2000), golden["post.normalize"])
    assert H.crossfade_chunks([]).numel() == 0 and torch.equal(H.crossfade_chunks([a]), a)
    fn = H._build_sample_fn({"num_steps": 12})
    assert fn.keywords["num_steps"] == 12 and fn.keywords["cfg_scale_speaker"] == 8.0 and fn.keywords["sequence_length"] == 640
    out = H.synthesize({"text": "   "}, None, None, None)
    assert out["error_type"] == "ValueError"


def _build_c_caller(tmp_path):
    """gcc -std=c11 on tests/c_abi/abi_smoke.c against include/echo_hip.h + libechohip.so (the header must be plain C)."""
    import shutil
    import subprocess
    gcc = shutil.which("gcc")
    if gcc is None or not os.path.isdir("/opt/rocm/include"):
        pytest.skip("gcc / ROCm headers not available")
    libdir = os.path.join(ROOT, "echo-tts_amd")
    exe = str(tmp_path / "abi_smoke")
    cmd = [gcc, "-std=c11", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"), "-I", "/opt/rocm/include",
           os.path.join(ROOT, "tests", "c_abi", "abi_smoke.c"), "-o", exe, "-L", libdir, "-lechohip", "-L", "/opt/rocm/lib", "-lamdhip64",
           f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


def test_header_is_plain_c_and_links(tmp_path):
    _build_c_caller(tmp_path)


def test_fp8_static_scale_helpers(tmp_path):
    """the calibrated fp8 activation scales travel as JSON; the oracle's restatement of the static mode quantises the named
    operand with the one scale (saturating at 448 s) and leaves every other block linear on per-row scales."""
    import torch
    from echo_tts_amd import weights as W
    from oracle import echo_ref as R
    t = torch.tensor([[0.01, 0.5], [0.02, 0.25], [3.0, 1e-3]])
    W.save_fp8_scales(str(tmp_path / "s.json"), t, {"model": "unit"})
    assert torch.equal(W.load_fp8_scales(str(tmp_path / "s.json")), t)
    (tmp_path / "bad.json").write_text('{"scales": [[1.0, -2.0]]}')
    import pytest
    with pytest.raises(ValueError):
        W.load_fp8_scales(str(tmp_path / "bad.json"))
    g = torch.Generator().manual_seed(0)
    x, w = torch.randn((5, 64), generator=g) * 3, torch.randn((8, 64), generator=g)
    q = R.fake_quant_static_e4m3(x, 0.01)
    assert abs(float(q.abs().max()) - 4.48) < 1e-6                  # saturates at 448 s
    assert torch.equal(R.fake_quant_static_e4m3(x, 1.0), x.to(torch.float8_e4m3fn).float())
    R.set_fp8_block_linears(True, act_static={"blocks.0.mlp.w2": 0.01})
    try:
        a, b = R.block_linear(x, w, "blocks.0.mlp", "w2"), R.block_linear(x, w, "blocks.0.mlp", "w1")
        c = R.block_linear(x, w, "blocks.1.mlp", "w2")
    finally:
        R.set_fp8_block_linears(False)
    assert not torch.equal(a, b) and torch.equal(b, c)
    assert torch.equal(R.block_linear(x, w, "blocks.0.mlp", "w2"), torch.nn.functional.linear(x, w))   # switched off again


def test_bench_host_logic(golden, monkeypatch):
    """bench.py without a GPU: C3's preset lengths are the reference tokenizer's (tests/golden/meta.json), and `--gpus N` outside a
    torchrun environment spawns the ranks BEFORE any GPU call (torch.cuda is never asked) and relays the child job's exit code."""
    import importlib.util
    import sys
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    assert bench.PRESET_TOKEN_LENGTHS == golden["__meta__"]["host"]["preset_token_lengths"]
    seen = {}

    class FakePopen:
        def __init__(self, cmd, **kw):
            seen["cmd"], seen["env"] = cmd, kw.get("env", {})
            self.stdout = iter(["noise\n", '{"metric": "m", "value": 1.0}\n'])

        def wait(self):
            return 0
    import subprocess
    monkeypatch.setattr(subprocess, "Popen", FakePopen)
    monkeypatch.setattr(torch.cuda, "is_available", lambda: (_ for _ in ()).throw(AssertionError("the spawning parent touched the GPU")))
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "2"])
    with pytest.raises(SystemExit) as ex:
        bench.main()
    assert ex.value.code == 0
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd and "127.0.0.1" in cmd
    assert cmd[-4:] == ["--gpus", "4", "--steps", "2"] and seen["env"].get("HSA_ENABLE_IPC_MODE_LEGACY") == "0"
    ids, tmask, x0 = None, None, None
    assert bench.dit_gemm_flops(1) > 1.3e14 and bench.free_port() > 0


def test_audio_io_host_side(tmp_path):
    """`load_audio`'s host half (reference inference.py:104-113; torchcodec / torchaudio are absent here, so this leg is PARITY UNPINNED and
    checked against its own published formula): the RIFF/WAVE parser round-trips every sample format through `wav_bytes`-style files,
    and the sinc / Hann filter bank has the properties the algorithm states (size, unit DC gain per phase, symmetry of phase 0)."""
    import math
    import struct
    from echo_tts_amd import audio_io as A
    g = torch.Generator().manual_seed(0)
    x = (torch.rand((2, 1000), generator=g) * 2 - 1) * 0.9
    p16 = tmp_path / "a16.wav"
    p16.write_bytes(A.wav_bytes(x, 22050))
    y, sr = A.read_wav(str(p16))
    assert sr == 22050 and y.shape == (2, 1000) and float((y - x).abs().max()) <= 2.0 / 32768
    y2, _ = A.read_wav(str(p16), max_duration=0.01)
    assert y2.shape == (2, 220)
    # float32 and 24-bit PCM files written by hand
    def riff(tag, bits, payload, ch=2, rate=48000):
        fmt = struct.pack("<HHIIHH", tag, ch, rate, rate * ch * bits // 8, ch * bits // 8, bits)
        return b"RIFF" + struct.pack("<I", 4 + 8 + len(fmt) + 8 + len(payload)) + b"WAVE" + b"fmt " + struct.pack("<I", len(fmt)) + fmt + b"data" + struct.pack("<I", len(payload)) + payload
    pf = tmp_path / "f32.wav"
    pf.write_bytes(riff(3, 32, x.t().contiguous().numpy().tobytes()))
    yf, srf = A.read_wav(str(pf))
    assert srf == 48000 and torch.equal(yf, x)
    v24 = (x.t().contiguous() * (1 << 23)).round().clamp(-(1 << 23), (1 << 23) - 1).to(torch.int32).reshape(-1)
    b24 = b"".join(int(v).to_bytes(3, "little", signed=True) for v in v24.tolist())
    p24 = tmp_path / "p24.wav"
    p24.write_bytes(riff(1, 24, b24))
    y24, _ = A.read_wav(str(p24))
    assert float((y24 - x).abs().max()) <= 2.0 ** -23 + 1e-7
    with pytest.raises(ValueError):
        (tmp_path / "bad.wav").write_bytes(b"OggS" + bytes(64))
        A.read_wav(str(tmp_path / "bad.wav"))
    # the filter bank (48 kHz -> 44.1 kHz: up 147, down 160; 24 kHz output leg: 44.1 -> 24 kHz: up 80, down 147)
    for o, n in ((48000, 44100), (44100, 24000), (16000, 44100)):
        bank, up, down, width = A.sinc_resample_bank(o, n)
        gg = math.gcd(o, n)
        assert (up, down) == (n // gg, o // gg) and bank.shape == (up, 2 * width + down)
        assert width == math.ceil(6 * down / (min(up, down) * 0.99))
        assert float((bank.double().sum(dim=1) - 1.0).abs().max()) < 2e-3          # a constant input stays (almost exactly) constant
        c = bank[0, : 2 * width + 1]                                                  # phase 0 is centred on input sample `width`
        assert torch.allclose(c, c.flip(0), atol=1e-7)
    assert A.resample(x, 44100, 44100) is x                                           # equal rates: untouched, no GPU needed
