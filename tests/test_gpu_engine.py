"""GPU: the HIP engine, through the reference-shaped Python API, against the golden vectors the
reference produced (tests/golden) and against the CPU oracle on the same seeded inputs.

Tolerances (BASELINE.json north_star): latents <= 1e-3 RMS, waveform <= 1e-4 RMS, asserted for the
fp32 engine against the fp32 reference.  The bf16 engine cannot meet 1e-3 against a bf16 PyTorch
run any more than PyTorch meets it against itself, so it is held to: no farther
from the fp32 reference than 1.5x the reference's own bf16 run (+1e-3)."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import echo_ref as R  # noqa: E402  (checker only)
from tests import gpu_util as U  # noqa: E402
from tests.golden_defs import SAMPLER_CASES, TINY, TINY_DAC, WIDE1  # noqa: E402

import echo_tts_amd as E  # noqa: E402
from echo_tts_amd.inference import _concat_kv_caches  # noqa: E402
from echo_tts_amd import _lib as L  # noqa: E402

DEV = U.DEV
LAT_TOL = 1e-3
WAV_TOL = 1e-4


def rms(a, b):
    return float((a.float().cpu() - b.float().cpu()).pow(2).mean().sqrt())


@pytest.fixture(scope="module")
def tiny_models():
    w = R.make_dit_weights(TINY, seed=0)
    return {"f32": E.EchoDiT(TINY, w, dtype=torch.float32, device=DEV),
            "bf16": E.EchoDiT(TINY, {k: v.bfloat16() for k, v in w.items()}, dtype=torch.bfloat16, device=DEV)}


def _bf16_budget(golden, key_bf16, key_f32):
    return 1.5 * rms(golden[key_bf16], golden[key_f32]) + 1e-3


@pytest.mark.parametrize("tag,batch", [("tiny", 1), ("tinyb2", 2)])
def test_kv_caches_and_forward_f32(golden, tiny_models, tag, batch):
    g, m = golden, tiny_models["f32"]
    ids, tm = g[f"{tag}.ids"], g[f"{tag}.tmask"].bool()
    spk, sm, x0 = g[f"{tag}.spk"], g[f"{tag}.smask"].bool(), g[f"{tag}.x0"]
    kvt = m.get_kv_cache_text(ids, tm)
    kvs = m.get_kv_cache_speaker(spk, sm)
    last = TINY.num_layers - 1
    k, v = kvt.layer(last)
    n = k.shape[1]
    valid = tm[:, :n, None, None].expand(-1, -1, k.shape[2], k.shape[3])
    for got, want in ((k, g[f"{tag}.f32.kvt_k_last"]), (v, g[f"{tag}.f32.kvt_v_last"])):
        assert rms(got.cpu()[valid], want[:, :n][valid]) < 1e-4       # padded text rows are never attended: not compared
    k, v = kvs.layer(last)
    n = k.shape[1]
    assert rms(k, g[f"{tag}.f32.kvs_k_last"][:, :n]) < 1e-4
    assert rms(v, g[f"{tag}.f32.kvs_v_last"][:, :n]) < 1e-4
    t = torch.full((batch,), 0.7)
    out = m(x0, t, tm, sm, kvt, kvs)
    assert rms(out, g[f"{tag}.f32.forward_v"]) < 2e-4
    tm3 = torch.cat([tm, torch.zeros_like(tm), tm], 0)
    sm3 = torch.cat([sm, sm, torch.zeros_like(sm)], 0)
    out3 = m(torch.cat([x0, x0, x0], 0), torch.full((3 * batch,), 0.7), tm3, sm3, _concat_kv_caches(kvt, kvt, kvt),
             _concat_kv_caches(kvs, kvs, kvs))
    assert rms(out3, g[f"{tag}.f32.forward_v3"]) < 2e-4


@pytest.mark.parametrize("tag,batch", [("tiny", 1), ("tinyb2", 2)])
def test_forward_bf16_within_reference_noise(golden, tiny_models, tag, batch):
    g, m = golden, tiny_models["bf16"]
    ids, tm = g[f"{tag}.ids"], g[f"{tag}.tmask"].bool()
    spk, sm, x0 = g[f"{tag}.spk"], g[f"{tag}.smask"].bool(), g[f"{tag}.x0"]
    kvt = m.get_kv_cache_text(ids, tm)
    kvs = m.get_kv_cache_speaker(spk, sm)
    out = m(x0.bfloat16(), torch.full((batch,), 0.7).bfloat16(), tm, sm, kvt, kvs)
    e = rms(out, g[f"{tag}.f32.forward_v"])
    assert e < _bf16_budget(g, f"{tag}.bf16.forward_v", f"{tag}.f32.forward_v"), e
    tm3 = torch.cat([tm, torch.zeros_like(tm), tm], 0)
    sm3 = torch.cat([sm, sm, torch.zeros_like(sm)], 0)
    out3 = m(torch.cat([x0, x0, x0], 0).bfloat16(), torch.full((3 * batch,), 0.7).bfloat16(), tm3, sm3,
             _concat_kv_caches(kvt, kvt, kvt), _concat_kv_caches(kvs, kvs, kvs))
    e3 = rms(out3, g[f"{tag}.f32.forward_v3"])
    assert e3 < _bf16_budget(g, f"{tag}.bf16.forward_v3", f"{tag}.f32.forward_v3"), e3


@pytest.mark.parametrize("tag", ["tiny", "tinyb2"])
@pytest.mark.parametrize("case", list(SAMPLER_CASES))
def test_euler_sampler_f32_meets_latent_tolerance(golden, tiny_models, tag, case):
    g, m = golden, tiny_models["f32"]
    lat = E.sample_euler_cfg_independent_guidances(m, g[f"{tag}.spk"], g[f"{tag}.smask"].bool(), g[f"{tag}.ids"],
                                                   g[f"{tag}.tmask"].bool(), rng_seed=0, sequence_length=32, x_init=g[f"{tag}.x0"],
                                                   **SAMPLER_CASES[case])
    e = rms(lat, g[f"{tag}.f32.euler.{case}"])
    assert e < LAT_TOL, e


@pytest.mark.parametrize("tag", ["tiny", "tinyb2"])
@pytest.mark.parametrize("case", list(SAMPLER_CASES))
def test_euler_sampler_bf16_within_reference_noise(golden, tiny_models, tag, case):
    g, m = golden, tiny_models["bf16"]
    lat = E.sample_euler_cfg_independent_guidances(m, g[f"{tag}.spk"], g[f"{tag}.smask"].bool(), g[f"{tag}.ids"],
                                                   g[f"{tag}.tmask"].bool(), rng_seed=0, sequence_length=32, x_init=g[f"{tag}.x0"],
                                                   **SAMPLER_CASES[case])
    e = rms(lat, g[f"{tag}.f32.euler.{case}"])
    assert e < _bf16_budget(g, f"{tag}.bf16.euler.{case}", f"{tag}.f32.euler.{case}"), e


@pytest.mark.parametrize("force", ["5,1", "2,1", "0,2"])
def test_euler_sampler_bf16_with_forced_gemm_plans(golden, monkeypatch, force):
    """The same sampler with every GEMM forced onto one tile kernel (5 = ping-pong 256x256 incl. its fused QKV / SwiGLU /
    gate-residual epilogues; 2 = 256x256 one-barrier pipeline; 0 with split-K): all within the bf16 budget."""
    monkeypatch.setenv("ECHO_GEMM_FORCE", force)
    w = R.make_dit_weights(TINY, seed=0)
    m = E.EchoDiT(TINY, {k: v.bfloat16() for k, v in w.items()}, dtype=torch.bfloat16, device=DEV)
    g, tag, case = golden, "tinyb2", "cfg_default"
    lat = E.sample_euler_cfg_independent_guidances(m, g[f"{tag}.spk"], g[f"{tag}.smask"].bool(), g[f"{tag}.ids"],
                                                   g[f"{tag}.tmask"].bool(), rng_seed=0, sequence_length=32, x_init=g[f"{tag}.x0"],
                                                   **SAMPLER_CASES[case])
    e = rms(lat, g[f"{tag}.f32.euler.{case}"])
    assert e < _bf16_budget(g, f"{tag}.bf16.euler.{case}", f"{tag}.f32.euler.{case}"), e


@pytest.mark.parametrize("force", ["5,1", None])
def test_ragged_sequence_length_bf16(golden, monkeypatch, force):
    """sequence_length 20 (not a multiple of 8, not of 16): the transposed-V stores of the fused QKV epilogue take their
    element-wise tail path and the last 128-query attention block is ragged.  bf16 engine (ping-pong GEMM forced / plans
    as tuned) against the fp32 oracle on the same inputs, within the bf16 budget of the tiny model."""
    if force:
        monkeypatch.setenv("ECHO_GEMM_FORCE", force)
    w = R.make_dit_weights(TINY, seed=0)
    m = E.EchoDiT(TINY, {k: v.bfloat16() for k, v in w.items()}, dtype=torch.bfloat16, device=DEV)
    g, tag = golden, "tinyb2"
    x0 = g[f"{tag}.x0"][:, :20].contiguous()
    args = (g[f"{tag}.spk"], g[f"{tag}.smask"].bool(), g[f"{tag}.ids"], g[f"{tag}.tmask"].bool())
    lat = E.sample_euler_cfg_independent_guidances(m, *args, rng_seed=0, sequence_length=20, x_init=x0, **SAMPLER_CASES["cfg_default"])
    ref = R.sample_euler(w, TINY, torch.float32, *args, rng_seed=0, sequence_length=20, x_init=x0, **SAMPLER_CASES["cfg_default"])
    budget = _bf16_budget(g, f"{tag}.bf16.euler.cfg_default", f"{tag}.f32.euler.cfg_default")
    assert rms(lat, ref) < 1.5 * budget, (rms(lat, ref), budget)


@pytest.mark.parametrize("case,opts,cont", [("plain", "cfg_default", False), ("cont_opts", "all_options", True)])
def test_blockwise_sampler_f32(golden, tiny_models, case, opts, cont):
    g, m = golden, tiny_models["f32"]
    xi = [g[f"tiny.blk_x{j}"] for j in range(3)]
    lat = E.sample_blockwise_euler_cfg_independent_guidances(
        m, g["tiny.spk"], g["tiny.smask"].bool(), g["tiny.ids"], g["tiny.tmask"].bool(), rng_seed=0, block_sizes=[16, 8, 8],
        continuation_latent=g["tiny.blk_cont"] if cont else None, x_inits=xi, **SAMPLER_CASES[opts])
    e = rms(lat, g[f"tiny.f32.blockwise.{case}"])
    assert e < LAT_TOL, e


def test_invalid_arguments_fail_loudly_and_the_context_survives(golden, tiny_models):
    """Error behaviour at the boundary (SURVEY.md §8b "Errors"): a bad call returns a status, the shim raises with
    echo_last_error's text, nothing is left half-done - the same context then reproduces the golden sampler run."""
    g, m = golden, tiny_models["f32"]
    ids, tm = g["tiny.ids"], g["tiny.tmask"].bool()
    spk, sm = g["tiny.spk"], g["tiny.smask"].bool()
    with pytest.raises(RuntimeError, match="multiple of the patch size"):
        m.get_kv_cache_speaker(spk[:, :-1], sm[:, :-1])           # the reference's reshape (model.py:455) raises here too
    kvt, kvs = m.get_kv_cache_text(ids, tm), m.get_kv_cache_speaker(spk, sm)
    x0 = g["tiny.x0"]
    with pytest.raises(ValueError, match="one entry per row"):    # t is per row (or one value for all rows), nothing in between
        m(torch.cat([x0, x0, x0], 0), torch.tensor([0.7, 0.6]), torch.cat([tm, tm, tm], 0), torch.cat([sm, sm, sm], 0),
          _concat_kv_caches(kvt, kvt, kvt), _concat_kv_caches(kvs, kvs, kvs))
    with pytest.raises(RuntimeError, match="rope table too short"):
        E.sample_euler_cfg_independent_guidances(m, spk, sm, ids, tm, rng_seed=0, sequence_length=100000, **SAMPLER_CASES["cfg_default"])
    b2 = (g["tinyb2.ids"], g["tinyb2.tmask"].bool(), g["tinyb2.spk"], g["tinyb2.smask"].bool())
    kvt2, kvs2 = m.get_kv_cache_text(b2[0], b2[1]), m.get_kv_cache_speaker(b2[2], b2[3])
    with pytest.raises(ValueError, match="multiple of the KV batch"):
        m(torch.cat([x0, x0, x0], 0), torch.full((3,), 0.7), torch.cat([b2[1], b2[1][:1]], 0), torch.cat([b2[3], b2[3][:1]], 0), kvt2, kvs2)
    lat = E.sample_euler_cfg_independent_guidances(m, spk, sm, ids, tm, rng_seed=0, sequence_length=32, x_init=x0, **SAMPLER_CASES["cfg_default"])
    assert rms(lat, g["tiny.f32.euler.cfg_default"]) <= LAT_TOL


def test_sampler_is_deterministic(golden, tiny_models):
    g, m = golden, tiny_models["bf16"]
    args = (m, g["tiny.spk"], g["tiny.smask"].bool(), g["tiny.ids"], g["tiny.tmask"].bool())
    a = E.sample_euler_cfg_independent_guidances(*args, rng_seed=3, sequence_length=32, **SAMPLER_CASES["cfg_default"])
    b = E.sample_euler_cfg_independent_guidances(*args, rng_seed=3, sequence_length=32, **SAMPLER_CASES["cfg_default"])
    assert torch.equal(a, b)


@pytest.mark.parametrize("dname,dt", [("f32", torch.float32), ("bf16", torch.bfloat16)])
def test_full_width_single_layer(golden, dname, dt):
    """Real head counts / widths (d=2048x16 heads, encoders 1280x10, F=5888/3328), one layer each."""
    g = golden
    w = R.make_dit_weights(WIDE1, seed=0)
    m = E.EchoDiT(WIDE1, {k: v.to(dt) for k, v in w.items()}, dtype=dt, device=DEV)
    ids, tm = g["wide1.ids"], g["wide1.tmask"].bool()
    spk, sm, x0 = g["wide1.spk"], g["wide1.smask"].bool(), g["wide1.x0"]
    kvt = m.get_kv_cache_text(ids, tm)
    kvs = m.get_kv_cache_speaker(spk, sm)
    out = m(x0.to(dt), torch.full((1,), 0.7).to(dt), tm, sm, kvt, kvs)
    e = rms(out, g["wide1.f32.forward_v"])
    if dt == torch.float32:
        assert e < 2e-4, e
    else:
        assert e < _bf16_budget(g, "wide1.bf16.forward_v", "wide1.f32.forward_v"), e


def test_dac_tiny_matches_reference(golden):
    g = golden
    w = R.make_dac_weights(TINY_DAC, 0)
    dac = E.DAC(TINY_DAC, w, device=DEV)
    wav = dac.decode_zq(g["dac_tiny.z"])
    e = rms(wav, g["dac_tiny.wav"])
    assert e < WAV_TOL and e < 1e-3 * U.rms(g["dac_tiny.wav"]), (e, U.rms(g["dac_tiny.wav"]))
    pca = R.make_pca(TINY_DAC, 80, 0)
    st = E.PCAState(pca.pca_components, pca.pca_mean, pca.latent_scale)
    out = E.ae_decode(dac, st, g["dac_tiny.latent"])
    e2 = rms(out, g["dac_tiny.ae_decode"])
    assert e2 < WAV_TOL, e2


def test_dac_full_size_matches_reference(golden):
    g = golden
    cfg = R.DacConfig()
    dac = E.DAC(cfg, R.make_dac_weights(cfg, 0), device=DEV)
    wav = dac.decode_zq(g["dac_full.z"])
    assert wav.shape == (1, 1, 8 * 2048)
    e = rms(wav, g["dac_full.wav"])
    assert e < WAV_TOL and e < 1e-3 * U.rms(g["dac_full.wav"]), (e, U.rms(g["dac_full.wav"]))
    pca = R.make_pca(cfg, 80, 0)
    out = E.ae_decode(dac, E.PCAState(pca.pca_components, pca.pca_mean, pca.latent_scale), g["dac_full.latent"])
    assert rms(out, g["dac_full.ae_decode"]) < WAV_TOL


# ----------------------------------------------------------------------- speaker-reference encode path
def _enc_weights(cfg):
    w = R.make_dac_weights(cfg, 0)
    w.update(R.make_dac_encoder_weights(cfg, 0))
    return w


def _check_encode(g, tag, cfg, n_samples, min_code_match):
    """codes: index work must be bit-exact (measured: 100 % on both fixtures).  The only legitimate difference is a near-tie of
    the argmax: the split3 fp32 GEMM sums in a different order than the CPU reference, so a frame whose two best codes are closer
    than fp32 noise may flip, and its later residual stages then see another input.  So for every frame that differs, the FIRST
    stage that differs must be such a near-tie in the oracle (distance margin < 1e-5 on unit vectors); anything else fails.
    z_q / latents: compared on the frames whose codes all agree."""
    dac = E.DAC(cfg, _enc_weights(cfg), device=DEV)
    audio = R.make_test_audio(n_samples, seed=11)
    codes, lens = dac.encode(audio)
    ref_codes = g[f"{tag}.codes"]
    assert codes.shape == ref_codes.shape and int(lens[0]) == ref_codes.shape[-1]
    same = (codes.cpu() == ref_codes)
    frac = float(same.float().mean())
    assert frac >= min_code_match, frac
    if frac < 1.0:
        taps = {}
        torch.set_num_threads(16)
        assert torch.equal(R.dac_encode_codes(_enc_weights(cfg), cfg, audio, taps), ref_codes)
        gap = taps["vq_gap"]
        for b, t in (~same.all(dim=1)).nonzero().tolist():
            k = int((~same[b, :, t]).nonzero()[0])
            assert float(gap[b, k, t]) < 1e-5, f"{tag}: frame {t} differs first at stage {k} where the oracle's argmax margin is {float(gap[b, k, t]):.3e}"
    ok = same.all(dim=1)[0]                                   # frames with every code equal
    zq = dac.encode_zq(audio).cpu()
    ref_zq = g[f"{tag}.zq"]
    e = float((zq[0][:, ok] - ref_zq[0][:, ok]).pow(2).mean().sqrt())
    assert e < 1e-5 * max(1.0, U.rms(ref_zq)), (e, U.rms(ref_zq))
    pca = R.make_pca(cfg, 80, 0)
    st = E.PCAState(pca.pca_components, pca.pca_mean, pca.latent_scale)
    lat = E.ae_encode(dac, st, audio).cpu()
    ref_lat = g[f"{tag}.ae_encode"]
    e2 = float((lat[0][ok] - ref_lat[0][ok]).pow(2).mean().sqrt())
    assert e2 < 1e-5 * max(1.0, U.rms(ref_lat)), e2
    return dac, st, audio, frac


def test_dac_encode_tiny_matches_reference(golden):
    from tests.golden_defs import TINY_ENC_SAMPLES
    dac, st, audio, frac = _check_encode(golden, "enc_tiny", TINY_DAC, TINY_ENC_SAMPLES, 0.99)
    lat, mask = E.get_speaker_latent_and_mask(dac, st, audio[0].to(DEV), max_speaker_latent_length=24, audio_chunk_size=4 * 2048)
    assert lat.shape == golden["enc_tiny.spk_latent"].shape and torch.equal(mask.cpu().to(torch.uint8), golden["enc_tiny.spk_mask"])
    if frac == 1.0:
        assert rms(lat, golden["enc_tiny.spk_latent"]) < 1e-5 * max(1.0, U.rms(golden["enc_tiny.spk_latent"]))


def test_dac_encode_full_size_matches_reference(golden):
    """Full-size encoder: 4-layer window-512 transformer over 600 positions, pre_module, VQ 4096 + 9 x 1024."""
    from tests.golden_defs import FULL_ENC_SAMPLES
    _check_encode(golden, "enc_full", R.DacConfig(), FULL_ENC_SAMPLES, 0.99)


def test_dac_encode_is_causal(golden):
    """Size-independent property: the encoder is causal, so the codes of a longer signal start with those of its prefix
    (whole frames), and a round trip decode(encode(x)) has the length of the padded input."""
    cfg = TINY_DAC
    dac = E.DAC(cfg, _enc_weights(cfg), device=DEV)
    audio = R.make_test_audio(2048 * 12, seed=5)
    ca, _ = dac.encode(audio[..., : 2048 * 7])
    cb, _ = dac.encode(audio)
    assert torch.equal(ca, cb[..., :7])
    wav = dac.decode_zq(dac.encode_zq(audio))
    assert wav.shape == (1, 1, 2048 * 12) and bool(torch.isfinite(wav).all())


def test_dac_full_size_160_frames_against_oracle():
    """Full-size Fish S1-DAC decode of 160 latent frames (327 680 samples, a quarter of config C2's 640) against the CPU
    oracle on the same seeded weights: waveform within the north-star 1e-4 RMS (the 8-frame reference fixture does not
    exercise the window-128 attention mask or the large-M convolution tiles)."""
    cfg = R.DacConfig()
    w = R.make_dac_weights(cfg, 0)
    z = torch.randn((1, cfg.latent_dim, 160), generator=torch.Generator().manual_seed(7))
    torch.set_num_threads(16)
    want = R.dac_decode_zq(w, cfg, z)
    dac = E.DAC(cfg, w, device=DEV)
    got = dac.decode_zq(z)
    e = rms(got, want)
    print(f"DAC full size, 160 frames: waveform rms error {e:.3e} (signal rms {U.rms(want):.3e})")
    assert got.shape == want.shape and e < WAV_TOL and e < 1e-3 * U.rms(want), (e, U.rms(want))


def test_dac_conv_tails_equal_the_generic_tail_bit_for_bit(tmp_path):
    """The branch-free conv tails of the tile kernels (gemm_epilogue NTAIL >= 2: clamped unconditional residual loads, idle threads' stores to
    a sink, per-column operands in front of the pass loop) must compute exactly what the generic tail computes: a full-size decode of 160
    frames with the conv tails on and off (ECHO_NT_CONV_TAILS, read once per process: two child processes), Snake on sinf() in both, must
    give the same waveform bit for bit.  A third run with the default Snake (v_sin_f32) must stay within 1e-3 of the north-star tolerance of it."""
    import subprocess
    script = tmp_path / "decode.py"
    script.write_text(
        "import sys, torch\n"
        f"sys.path.insert(0, {os.path.dirname(os.path.dirname(os.path.abspath(__file__)))!r})\n"
        "import echo_tts_amd as E\n"
        "from oracle import echo_ref as R\n"
        "cfg = R.DacConfig(); w = R.make_dac_weights(cfg, 0)\n"
        "z = torch.randn((1, cfg.latent_dim, 160), generator=torch.Generator().manual_seed(7))\n"
        "dac = E.DAC(cfg, w, device='cuda:0')\n"
        "torch.save(dac.decode_zq(z).cpu(), sys.argv[1])\n")
    outs = {}
    for name, env in (("conv", {"ECHO_NT_CONV_TAILS": "1", "ECHO_DAC_FAST_SIN": "0"}), ("generic", {"ECHO_NT_CONV_TAILS": "0", "ECHO_DAC_FAST_SIN": "0"}),
                      ("default", {})):
        out = tmp_path / f"{name}.pt"
        r = subprocess.run([sys.executable, str(script), str(out)], env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs[name] = torch.load(out, weights_only=True)
    assert torch.equal(outs["conv"], outs["generic"])
    d = rms(outs["default"], outs["generic"])
    print(f"DAC conv tails: bit-identical to the generic tail; v_sin_f32 Snake vs sinf: rms {d:.3e} (signal rms {U.rms(outs['generic']):.3e})")
    assert d < 1e-3 * WAV_TOL


@pytest.mark.parametrize("size, B, T", [("tiny", 3, 40), ("full", 5, 200)])
def test_dac_decode_batch_equals_single_decodes(size, B, T):
    """echo_dac_decode_batch (ABI 6; `DAC.decode_latent` on a batch): the PCA inverse, the post_module transformer and its norm run once on
    the B * T stacked rows (window-128 causal attention per item and head: T = 200 crosses the window; the items must not see each
    other), the convolution stack per item.  Every item must equal its own single decode up to the fp32 summation order of the row-wise
    GEMMs (other M, other tile plan), and the batch call must be reproducible."""
    cfg = TINY_DAC if size == "tiny" else R.DacConfig()
    w = R.make_dac_weights(cfg, 0)
    dac = E.DAC(cfg, w, device=DEV)
    pca = R.make_pca(cfg, 80, 0)
    st = E.PCAState(pca.pca_components, pca.pca_mean, pca.latent_scale)
    lat = torch.randn((B, T, 80), generator=torch.Generator().manual_seed(11))
    lat[1] *= 3.0                                            # items of different scale: a leak between items would show
    allb = E.ae_decode(dac, st, lat)
    again = E.ae_decode(dac, st, lat)
    assert torch.equal(allb, again) and bool(torch.isfinite(allb).all())
    worst = 0.0
    for b in range(B):
        one = E.ae_decode(dac, st, lat[b:b + 1])
        e = rms(allb[b:b + 1], one)
        worst = max(worst, e / max(U.rms(one), 1e-12))
        assert e < 3e-5 * U.rms(one) + 1e-7, (b, e, U.rms(one))       # measured 6e-6 (full size): split-K / tile plans differ with M
    print(f"DAC decode batch ({size}, {B} x {T} frames): worst relative rms difference to single decodes {worst:.2e}")


def test_dac_is_causal_and_length_independent(golden):
    """Size-independent property: decoding a longer input reproduces the shorter one's samples."""
    cfg = TINY_DAC
    dac = E.DAC(cfg, R.make_dac_weights(cfg, 0), device=DEV)
    z = torch.randn((1, cfg.latent_dim, 40), generator=torch.Generator().manual_seed(1))
    a = dac.decode_zq(z[..., :24])
    b = dac.decode_zq(z)
    assert rms(a, b[..., : a.shape[-1]]) < 1e-6


def test_c1_full_depth_fp32_against_oracle():
    """BASELINE config C1 at FULL size (24-layer EchoDiT, 14-layer encoders): '[S1] Hello world.' (18 tokens), no speaker
    reference, S=128, 10 Euler steps, CFG off.  The fp32 engine must match the CPU oracle (fp32, same seeded weights)
    within the north-star latent tolerance."""
    from echo_tts_amd.inference import get_text_input_ids_and_mask
    cfg = R.DiTConfig()
    w = R.make_dit_weights(cfg, seed=0, with_blockwise=False)
    ids, tmask = get_text_input_ids_and_mask(["[S1] Hello world."], max_length=None)
    assert ids.shape == (1, 18)
    spk = torch.zeros((1, 4, 80))
    smask = torch.zeros((1, 4), dtype=torch.bool)
    x0 = torch.randn((1, 128, 80), generator=torch.Generator().manual_seed(0))
    assert abs(float(x0[0, 0, 0]) + 1.12584) < 1e-4       # first value of the CPU generator, seed 0
    kw = dict(num_steps=10, cfg_scale_text=3.0, cfg_scale_speaker=8.0, cfg_min_t=1.1, cfg_max_t=1.0, truncation_factor=None,
              rescale_k=None, rescale_sigma=None, speaker_kv_scale=None, speaker_kv_max_layers=None, speaker_kv_min_t=None)
    torch.set_num_threads(16)
    want = R.sample_euler(w, cfg, torch.float32, spk, smask, ids, tmask, rng_seed=0, sequence_length=128, x_init=x0, **kw)
    m = E.EchoDiT(cfg, w, dtype=torch.float32, device=DEV)
    got = E.sample_euler_cfg_independent_guidances(m, spk, smask, ids, tmask, rng_seed=0, sequence_length=128, x_init=x0, **kw)
    e = rms(got, want)
    assert e < LAT_TOL, (e, U.rms(want))
    del m
    mb = E.EchoDiT(cfg, {k: v.bfloat16() for k, v in w.items()}, dtype=torch.bfloat16, device=DEV)
    gotb = E.sample_euler_cfg_independent_guidances(mb, spk, smask, ids, tmask, rng_seed=0, sequence_length=128, x_init=x0, **kw)
    eb = rms(gotb, want)
    # calibration: PyTorch's own bf16 run (the oracle in bf16 on the CPU) against its fp32 run, same weights
    wantb = R.sample_euler({k: v.bfloat16() for k, v in w.items()}, cfg, torch.bfloat16, spk, smask, ids, tmask, rng_seed=0,
                           sequence_length=128, x_init=x0, **kw)
    eref = rms(wantb, want)
    print(f"C1 full depth: fp32 engine rms {e:.3e}; bf16 engine rms {eb:.3e} vs fp32 oracle; PyTorch bf16 vs fp32 {eref:.3e}; "
          f"bf16 engine vs PyTorch bf16 {rms(gotb, wantb):.3e} (latent rms {U.rms(want):.3f})")
    assert eb < 1.5 * eref + 1e-3, (eb, eref)


C2_KW = dict(cfg_scale_text=3.0, cfg_scale_speaker=8.0, cfg_min_t=0.5, cfg_max_t=1.0, truncation_factor=None,
             rescale_k=None, rescale_sigma=None, speaker_kv_scale=None, speaker_kv_max_layers=None, speaker_kv_min_t=None)


@pytest.fixture(scope="module")
def full_size():
    """Full-size EchoDiT (24 layers, d = 2048; 14-layer encoders) on seeded weights: the fp32 parity engine, the bf16 production
    engine (the one bench.py times) and the weights themselves for the oracle; built once for the C2 / C3 tests below."""
    cfg = R.DiTConfig()
    w = R.make_dit_weights(cfg, seed=0, with_blockwise=False)
    wb = {k: v.bfloat16() for k, v in w.items()}
    return {"cfg": cfg, "w": w, "wb": wb,
            "f32": E.EchoDiT(cfg, w, dtype=torch.float32, device=DEV),
            "bf16": E.EchoDiT(cfg, wb, dtype=torch.bfloat16, device=DEV)}


def _c2_inputs():
    g = torch.Generator().manual_seed(1234)
    ids = torch.zeros((1, 768), dtype=torch.int32)
    ids[0, 1:436] = torch.randint(32, 127, (435,), generator=g, dtype=torch.int32)
    tmask = torch.zeros((1, 768), dtype=torch.bool)
    tmask[0, :436] = True
    spk = torch.randn((1, 2560, 80), generator=g)
    smask = torch.ones((1, 2560), dtype=torch.bool)
    x0 = torch.randn((1, 640, 80), generator=torch.Generator().manual_seed(0))
    return ids, tmask, spk, smask, x0


def _eager_gpu_bf16_vs_fp32(fs, spk, smask, ids, tmask, x0, S, kw):
    """The reference's own bf16 noise floor at this shape: PyTorch-ROCm eager (the oracle's torch ops on the MI355X) in bf16
    against the same ops in fp32, same weights and noise (SURVEY.md §A.4 measured it on the CPU at small sizes)."""
    dev = lambda t: t.to(DEV)
    wf = {k: v.to(DEV) for k, v in fs["w"].items()}
    a = R.sample_euler(wf, fs["cfg"], torch.float32, dev(spk), dev(smask), dev(ids), dev(tmask), rng_seed=0, sequence_length=S, x_init=dev(x0), **kw)
    del wf
    wb = {k: v.to(DEV) for k, v in fs["wb"].items()}
    b = R.sample_euler(wb, fs["cfg"], torch.bfloat16, dev(spk), dev(smask), dev(ids), dev(tmask), rng_seed=0, sequence_length=S, x_init=dev(x0), **kw)
    del wb
    torch.cuda.empty_cache()
    return a.cpu(), b.cpu()


def test_c2_shape_full_depth_cfg_fp32_and_bf16_against_oracle(full_size):
    """BASELINE config C2's SHAPES at full size with CFG on: 24-layer EchoDiT, S = 640, text of 436 tokens padded to 768,
    speaker reference of 2560 latents (640 keys), cfg_text 3 / cfg_speaker 8 - 3 of C2's 40 Euler steps, of which the first two
    are 3-row CFG steps (t = 0.999, 0.666 >= 0.5) and the last a 1-row step; the CPU oracle needs ~20 s per step.
    * fp32 engine vs the fp32 CPU oracle on the same noise: the north-star latent tolerance (1e-3 RMS);
    * bf16 engine - gemm_pp_kernel, attn_kernel<bf16>, the fused QKV tail at M = 1920 / 640, i.e. the production kernels - vs
      the same fp32 oracle: no farther than 1.5 x PyTorch's own bf16-vs-fp32 distance at this shape (+1e-3), where PyTorch's
      bf16 run is the oracle's torch ops executed eagerly on the MI355X (the "vendored PyTorch path" of the north star)."""
    fs = full_size
    ids, tmask, spk, smask, x0 = _c2_inputs()
    kw = dict(C2_KW, num_steps=3)
    torch.set_num_threads(16)
    want = R.sample_euler(fs["w"], fs["cfg"], torch.float32, spk, smask, ids, tmask, rng_seed=0, sequence_length=640, x_init=x0, **kw)
    got = E.sample_euler_cfg_independent_guidances(fs["f32"], spk, smask, ids, tmask, rng_seed=0, sequence_length=640, x_init=x0, **kw)
    e = rms(got, want)
    print(f"C2 shapes, full depth, 3 steps with CFG: fp32 engine rms {e:.3e} vs fp32 oracle (latent rms {U.rms(want):.3f})")
    assert e < LAT_TOL, (e, U.rms(want))
    gotb = E.sample_euler_cfg_independent_guidances(fs["bf16"], spk, smask, ids, tmask, rng_seed=0, sequence_length=640, x_init=x0, **kw)
    ea, eb_ref = _eager_gpu_bf16_vs_fp32(fs, spk, smask, ids, tmask, x0, 640, kw)
    eref = rms(eb_ref, want)
    eb = rms(gotb, want)
    print(f"C2 shapes: bf16 engine rms {eb:.3e} vs fp32 oracle; PyTorch-ROCm eager bf16 vs fp32 oracle {eref:.3e}; eager fp32 on the GPU vs "
          f"CPU oracle {rms(ea, want):.3e}; bf16 engine vs eager bf16 {rms(gotb, eb_ref):.3e}")
    assert bool(torch.isfinite(gotb).all())
    assert eb < 1.5 * eref + 1e-3, (eb, eref)


@pytest.mark.parametrize("B, check_rows", [(8, range(8)), (24, (0, 11, 23))], ids=["batch8", "batch24"])
def test_bench_workload_batch_rows_match_single_fp32_calls(golden, full_size, B, check_rows):
    """The exact workload bench.py times (BASELINE C2 / C3 on one GPU): 8 (the round-1 default) and 24 (the default now) utterances
    through one bf16 sampler call at full depth - M = 15360 / 46080 GEMM rows in the CFG steps (7.5 / 22.5 output tiles per workgroup
    of the persistent kernel), 24- / 72-row attention launches, one reference voice shared by all rows - with the mixed text lengths of text_presets.txt (token counts
    of the first eight presets; the texts themselves stay in the reference).  4 Euler steps = 2 CFG steps + 2 plain ones.
    Each row must match a SINGLE-utterance call of the fp32 parity engine (itself pinned to the oracle above) within the bf16
    budget of that shape, the batched call must be bit-reproducible, and it must equal the bf16 engine's own single calls to
    within the same budget (another batch shape means other tile plans, i.e. other rounding points)."""
    fs = full_size
    lens8 = golden["__meta__"]["host"]["preset_token_lengths"][:8]
    lens = [lens8[b % 8] for b in range(B)]
    S = 640
    g = torch.Generator().manual_seed(77)
    ids = torch.zeros((B, 768), dtype=torch.int32)
    tmask = torch.zeros((B, 768), dtype=torch.bool)
    for b, n in enumerate(lens):
        ids[b, 1:n] = torch.randint(32, 127, (n - 1,), generator=g, dtype=torch.int32)
        tmask[b, :n] = True
    spk = torch.randn((1, 2560, 80), generator=g)
    smask = torch.ones((1, 2560), dtype=torch.bool)
    x0 = torch.randn((B, S, 80), generator=g)
    kw = dict(C2_KW, num_steps=4)
    run = lambda m, sl: E.sample_euler_cfg_independent_guidances(m, spk, smask, ids[sl], tmask[sl], rng_seed=0, sequence_length=S,
                                                                 x_init=x0[sl], **kw).cpu()
    all8 = run(fs["bf16"], slice(0, B))
    again = run(fs["bf16"], slice(0, B))
    assert torch.equal(all8, again), "the batched bf16 sampler call is not bit-reproducible"
    assert bool(torch.isfinite(all8).all())
    # the reference's own bf16 noise floor at this shape, on row 0's inputs (PyTorch-ROCm eager bf16 vs fp32)
    ea, eb_ref = _eager_gpu_bf16_vs_fp32(fs, spk, smask, ids[:1], tmask[:1], x0[:1], S, kw)
    eref = rms(eb_ref, ea)
    worst, worst_self = 0.0, 0.0
    for b in check_rows:
        ref = run(fs["f32"], slice(b, b + 1))
        one = run(fs["bf16"], slice(b, b + 1))
        e, es = rms(all8[b:b + 1], ref), rms(all8[b:b + 1], one)
        worst, worst_self = max(worst, e), max(worst_self, es)
        assert e < 1.5 * eref + 1e-3, (b, e, eref)
        assert es < 1.5 * eref + 1e-3, (b, es, eref)
        if b == 0:
            assert rms(ref, ea) < LAT_TOL      # the fp32 engine agrees with eager fp32 on the GPU at this shape as well
    print(f"batch {B} x full depth x 4 steps: worst row rms {worst:.3e} vs fp32 single calls, {worst_self:.3e} vs bf16 single calls; "
          f"PyTorch-ROCm bf16 vs fp32 at this shape {eref:.3e}")


def test_full_width_layer_rows_span_many_tiles(golden):
    """One full-width layer (WIDE1: d = 2048 x 16 heads, encoders 1280 x 10) with 12 rows x 640 latents = 7680 GEMM rows (960 tiles
    in the QKVG projection): bf16 engine vs fp32 engine on the same inputs, one forward; the fused QKV / SwiGLU / residual tails
    of the ping-pong kernel at a multi-tile shape inside the engine, without the depth that blurs a single wrong tile."""
    cfg = WIDE1
    w = R.make_dit_weights(cfg, seed=0)
    mf = E.EchoDiT(cfg, w, dtype=torch.float32, device=DEV)
    mb = E.EchoDiT(cfg, {k: v.bfloat16() for k, v in w.items()}, dtype=torch.bfloat16, device=DEV)
    gen = torch.Generator().manual_seed(5)
    B, S = 4, 640
    ids = torch.randint(1, 256, (B, 300), generator=gen, dtype=torch.int32)
    tmask = torch.ones((B, 300), dtype=torch.bool)
    tmask[1, 200:] = False
    spk, smask = torch.randn((1, 512, 80), generator=gen), torch.ones((1, 512), dtype=torch.bool)
    x = torch.randn((3 * B, S, 80), generator=gen)
    tm3 = torch.cat([tmask, torch.zeros_like(tmask), tmask], 0)
    sm1 = smask.expand(B, -1)
    sm3 = torch.cat([sm1, sm1, torch.zeros_like(sm1)], 0)
    outs = []
    for m, dt in ((mf, torch.float32), (mb, torch.bfloat16)):
        kvt, kvs = m.get_kv_cache_text(ids, tmask), m.get_kv_cache_speaker(spk, smask)
        # t = 0.75 is exact in bf16: the reference's bf16 path rounds t before the timestep embedding (inference.py:489), and
        # bf16(0.7) would move the embedding's phases by up to 0.8 rad - a property of the reference, not of a kernel
        outs.append(m(x.to(dt), torch.full((3 * B,), 0.75).to(dt), tm3, sm3, _concat_kv_caches(kvt, kvt, kvt), kvs).float().cpu())
    rel = rms(outs[0], outs[1]) / U.rms(outs[0])
    per_row = (outs[0] - outs[1]).pow(2).mean(dim=(1, 2)).sqrt() / outs[0].pow(2).mean(dim=(1, 2)).sqrt()
    print(f"full-width layer, 12 x 640 rows: bf16 vs fp32 engine relative rms {rel:.3e}, worst row {float(per_row.max()):.3e}")
    assert rel < 2e-2 and float(per_row.max()) < 3e-2, (rel, per_row)


@pytest.mark.parametrize("dname,dt", [("f32", torch.float32), ("bf16", torch.bfloat16)])
def test_pipeline_and_handler_surface(golden, tiny_models, dname, dt):
    """sample_pipeline / sample_pipeline_chunked / handler.synthesize run end to end on the tiny models (text front end,
    sampler, ae_decode, crop, chunking, cross-fade) and agree with the oracle evaluated stage by stage."""
    from functools import partial
    from echo_tts_amd import handler as H
    from echo_tts_amd import inference as inf
    m = tiny_models[dname]
    dw = R.make_dac_weights(TINY_DAC, 0)
    dac = E.DAC(TINY_DAC, dw, device=DEV)
    pca = R.make_pca(TINY_DAC, 80, 0)
    st = E.PCAState(pca.pca_components, pca.pca_mean, pca.latent_scale)
    fn = partial(E.sample_euler_cfg_independent_guidances, sequence_length=32, **SAMPLER_CASES["cfg_default"])
    g = golden
    audio, norm = E.sample_pipeline(m, dac, st, fn, "Hello world.", None, 3, pad_to_max_text_length=64,
                                    speaker_latent=g["tiny.spk"], speaker_mask=g["tiny.smask"].bool())
    assert norm == "[S1] Hello world." and audio.shape[:2] == (1, 1) and audio.shape[-1] % 2048 == 0
    assert bool(torch.isfinite(audio).all())
    if dt == torch.float32:
        # the same request through the oracle: text ids, sampler (with the device's own noise), decode, crop
        ids, tmask = inf.get_text_input_ids_and_mask(["Hello world."], max_length=64)
        x0 = torch.randn((1, 32, 80), device=DEV, dtype=torch.float32, generator=torch.Generator(device=DEV).manual_seed(3)).cpu()
        w = R.make_dit_weights(TINY, seed=0)
        lat = R.sample_euler(w, TINY, torch.float32, g["tiny.spk"], g["tiny.smask"].bool(), ids, tmask, rng_seed=3,
                             sequence_length=32, x_init=x0, **SAMPLER_CASES["cfg_default"])
        wav = R.ae_decode(dw, TINY_DAC, pca, lat)
        wav = wav[..., : R.find_flattening_point(lat[0]) * 2048]
        assert wav.shape == audio.shape
        assert rms(audio, wav) < WAV_TOL
    long_text = "First sentence here. Second sentence follows, with a clause. Third one!"
    a2, n2 = E.sample_pipeline_chunked(m, dac, st, fn, long_text, None, 0, max_chars_per_chunk=30, pad_to_max_text_length=64)
    assert n2.count("\n") == len(inf.chunk_text(long_text, 30)) - 1 and bool(torch.isfinite(a2).all())
    out = H.synthesize({"text": long_text, "parameters": {"num_steps": 4, "sequence_length": 32, "target_duration_seconds": 2.5}},
                       m, dac, st)
    assert "error" not in out, out.get("traceback")
    assert out["chunks"] == len(H.chunk_text_for_audio(long_text, 300, 2.5)) and out["audio"].shape[-1] > 0


@pytest.mark.parametrize("dname,dt,tol", [("f32", torch.float32, 2e-5), ("bf16", torch.bfloat16, 8e-2)])
def test_batched_call_equals_single_calls_at_full_width(dname, dt, tol):
    """BASELINE config C3's shape of work at real widths (d = 2048 x 16 heads, encoders 1280 x 10, one layer each): four
    utterances with different text lengths and speaker lengths through ONE sampler call (the reference's batch axis, which
    is what bench.py times) must give what four single-utterance calls give.  fp32: same arithmetic up to the GEMM plan
    (tile / split-K choice changes the summation order); bf16: two different sets of bf16 rounding points (other tiles, other
    split-K) amplified by CFG scale 8 on random weights, so only a sanity bound (measured 3e-2)."""
    from echo_tts_amd.inference import get_text_input_ids_and_mask
    cfg = WIDE1
    w = R.make_dit_weights(cfg, seed=0)
    m = E.EchoDiT(cfg, {k: v.to(dt) for k, v in w.items()}, dtype=dt, device=DEV)
    texts = ["[S1] Hello world.", "[S1] A somewhat longer sentence, with a clause in the middle.", "[S1] Short.",
             "[S1] The quick brown fox jumps over the lazy dog near the river bank at dawn."]
    ids, tmask = get_text_input_ids_and_mask(texts, max_length=96)
    g = torch.Generator().manual_seed(9)
    spk = torch.randn((4, 64, 80), generator=g)
    smask = torch.ones((4, 64), dtype=torch.bool)
    smask[1, 48:] = False
    smask[3, 32:] = False
    x0 = torch.randn((4, 64, 80), generator=g)
    kw = dict(SAMPLER_CASES["cfg_default"], num_steps=4)
    both = E.sample_euler_cfg_independent_guidances(m, spk, smask, ids, tmask, rng_seed=0, sequence_length=64, x_init=x0, **kw).cpu()
    for b in range(4):
        one = E.sample_euler_cfg_independent_guidances(m, spk[b:b + 1], smask[b:b + 1], ids[b:b + 1], tmask[b:b + 1], rng_seed=0,
                                                       sequence_length=64, x_init=x0[b:b + 1], **kw).cpu()
        e = rms(both[b:b + 1], one)
        assert e < tol * max(1.0, U.rms(one)), (b, e, U.rms(one))


def test_handler_batches_the_chunks_of_a_request(golden, tiny_models):
    """handler.synthesize sends the independent text chunks of a request through ONE sampler call (parameters.max_chunk_batch,
    default 8) instead of one call per chunk (handler.py:747-759): same seeds, same noise per chunk; on the fp32 engine the
    result must equal the sequential path within the waveform tolerance."""
    from echo_tts_amd import handler as H
    m = tiny_models["f32"]
    dac = E.DAC(TINY_DAC, R.make_dac_weights(TINY_DAC, 0), device=DEV)
    pca = R.make_pca(TINY_DAC, 80, 0)
    st = E.PCAState(pca.pca_components, pca.pca_mean, pca.latent_scale)
    text = ("The quick brown fox jumps over the lazy dog near the quiet river bank. " * 3 + "Then it rests for a while under the old oak tree. " * 3).strip()
    base = dict(SAMPLER_CASES["cfg_default"], sequence_length=32, seed=5, max_chars_per_chunk=90, normalize_boundaries=False, enable_crossfade=False)
    outs = {}
    for mb in (1, 8, 2):
        out = H.synthesize({"text": text, "parameters": dict(base, max_chunk_batch=mb)}, m, dac, st,
                           speaker_latent=golden["tiny.spk"], speaker_mask=golden["tiny.smask"].bool())
        assert "error" not in out, out.get("traceback")
        outs[mb] = out
    assert outs[1]["chunks"] >= 3 and outs[8]["chunks"] == outs[1]["chunks"]
    for mb in (8, 2):
        a, b = outs[1]["audio"].float().cpu(), outs[mb]["audio"].float().cpu()
        assert a.shape == b.shape, (a.shape, b.shape)
        assert rms(a, b) <= WAV_TOL, rms(a, b)


@pytest.mark.statistical
def test_fp8_weight_numerics_c5(golden, tiny_models):
    """BASELINE config C5 as a parity case: DiT block linears stored as OCP fp8-e4m3 (per-row scale), 100 Euler steps.  No
    tolerance is promised for fp8: the RMS distance to the bf16 engine is reported and only sanity-bounded."""
    from echo_tts_amd.weights import fp8_weight_state
    g, tag = golden, "tiny"
    w = {k: v.bfloat16() for k, v in R.make_dit_weights(TINY, seed=0).items()}
    m8 = E.EchoDiT(TINY, fp8_weight_state(w), dtype=torch.bfloat16, device=DEV)
    kw = dict(SAMPLER_CASES["cfg_default"], num_steps=100)
    args = (g[f"{tag}.spk"], g[f"{tag}.smask"].bool(), g[f"{tag}.ids"], g[f"{tag}.tmask"].bool())
    a = E.sample_euler_cfg_independent_guidances(tiny_models["bf16"], *args, rng_seed=0, sequence_length=32, x_init=g[f"{tag}.x0"], **kw)
    b = E.sample_euler_cfg_independent_guidances(m8, *args, rng_seed=0, sequence_length=32, x_init=g[f"{tag}.x0"], **kw)
    e = rms(a, b)
    print(f"C5 (tiny, 100 steps): fp8-e4m3 weights vs bf16 weights: latent rms distance {e:.3e} (latent rms {U.rms(a):.3f})")
    assert bool(torch.isfinite(b).all()) and 0.0 < e < 0.5 * U.rms(a)


@pytest.mark.statistical
def test_fp8_mfma_engine_c5(golden, tiny_models):
    """BASELINE config C5 on the real fp8 path (EchoDiT(fp8=True): e4m3 weights per output row + e4m3 activations per token
    row on the block-scaled MFMA): 100 Euler steps on the tiny model and one forward at full width, distance to the bf16
    engine reported and sanity-bounded (no tolerance is promised for fp8); deterministic."""
    g, tag = golden, "tiny"
    w = {k: v.bfloat16() for k, v in R.make_dit_weights(TINY, seed=0).items()}
    m8 = E.EchoDiT(TINY, w, dtype=torch.bfloat16, device=DEV, fp8=True)
    kw = dict(SAMPLER_CASES["cfg_default"], num_steps=100)
    args = (g[f"{tag}.spk"], g[f"{tag}.smask"].bool(), g[f"{tag}.ids"], g[f"{tag}.tmask"].bool())
    a = E.sample_euler_cfg_independent_guidances(tiny_models["bf16"], *args, rng_seed=0, sequence_length=32, x_init=g[f"{tag}.x0"], **kw)
    b = E.sample_euler_cfg_independent_guidances(m8, *args, rng_seed=0, sequence_length=32, x_init=g[f"{tag}.x0"], **kw)
    b2 = E.sample_euler_cfg_independent_guidances(m8, *args, rng_seed=0, sequence_length=32, x_init=g[f"{tag}.x0"], **kw)
    e = rms(a, b)
    print(f"C5 (tiny, 100 steps): fp8 MFMA path vs bf16 engine: latent rms distance {e:.3e} (latent rms {U.rms(a):.3f})")
    assert torch.equal(b, b2)
    assert bool(torch.isfinite(b).all()) and 0.0 < e < 0.5 * U.rms(a)
    with pytest.raises(RuntimeError, match="dit_fp8 needs precision"):
        E.EchoDiT(TINY, {k: v.float() for k, v in w.items()}, dtype=torch.float32, device=DEV, fp8=True)
    # full width, one layer of each stack: a single velocity prediction
    ww = {k: v.bfloat16() for k, v in R.make_dit_weights(WIDE1, seed=0).items()}
    mb, mf = E.EchoDiT(WIDE1, ww, dtype=torch.bfloat16, device=DEV), E.EchoDiT(WIDE1, ww, dtype=torch.bfloat16, device=DEV, fp8=True)
    gen = torch.Generator().manual_seed(3)
    ids = torch.randint(1, 256, (1, 40), generator=gen, dtype=torch.int32)
    tmask = torch.ones((1, 40), dtype=torch.bool)
    spk, smask = torch.randn((1, 64, 80), generator=gen), torch.ones((1, 64), dtype=torch.bool)
    x = torch.randn((1, 200, 80), generator=gen)
    outs = []
    for m in (mb, mf):
        kvt, kvs = m.get_kv_cache_text(ids, tmask), m.get_kv_cache_speaker(spk, smask)
        outs.append(m(x, torch.full((1,), 0.6), tmask, smask, kvt, kvs).float().cpu())
    rel = rms(outs[0], outs[1]) / U.rms(outs[0])
    print(f"C5 (full width, 1 layer): fp8 forward vs bf16 forward: relative rms {rel:.3e}")
    assert rel < 0.1, rel


@pytest.mark.statistical
def test_fp8_engine_against_fake_quant_restatement(golden):
    """BASELINE config C5 pinned to something other than itself.  The reference has no fp8 path (parity with the reference:
    unpinned by nature), but the algorithm the fp8 engine states - e4m3 operands for the four block linears, one scale per weight
    row and per token row (amax / 448), fp32 accumulation, bf16 tails - is restated in the oracle (`set_fp8_block_linears`).  One
    velocity prediction at FULL width (one layer of each stack) and on the tiny model: the engine must be closer to that
    restatement than the restatement is to plain bf16 (fp8's own effect).
    This END-TO-END comparison is loose by construction, and the bound says so: re-quantising to e4m3 amplifies every legitimate
    bf16-level difference upstream into 6-12 % steps of the affected codes - the restatement moves by 0.32 (d = 2048) / 0.23 (d = 256)
    of fp8's effect against ITSELF when its SDPA is replaced by a flash-style evaluation (P rounded to bf16), 5x more than the plain
    bf16 forward moves under the same swap (measured on the CPU, round 3).  The engine was measured at 0.50-0.53 of the effect on five
    boxes; the bound is 0.8 (1.5x margin).  The arithmetic itself is pinned where it can be pinned tightly: per linear, teacher-forced,
    in tests/test_gpu_kernels.py::test_fp8_block_linears_teacher_forced_against_the_restatement (scales bit-equal, >= 99.8 % of the
    e4m3 codes identical, outputs >= 4x closer than fp8's effect on that linear)."""
    for cfg, S, T in ((WIDE1, 200, 40), (TINY, 32, 24)):
        wb = {k: v.bfloat16() for k, v in R.make_dit_weights(cfg, seed=0).items()}
        gen = torch.Generator().manual_seed(3)
        ids = torch.randint(1, 256, (1, T), generator=gen, dtype=torch.int32)
        tmask = torch.ones((1, T), dtype=torch.bool)
        spk, smask = torch.randn((1, 64, 80), generator=gen).bfloat16(), torch.ones((1, 64), dtype=torch.bool)
        x = torch.randn((1, S, 80), generator=gen).bfloat16()
        t = torch.full((1,), 0.75).bfloat16()
        m8 = E.EchoDiT(cfg, wb, dtype=torch.bfloat16, device=DEV, fp8=True)
        got = m8(x, t, tmask, smask, m8.get_kv_cache_text(ids, tmask), m8.get_kv_cache_speaker(spk, smask)).float().cpu()
        kvt, kvs = R.kv_cache_text(wb, cfg, ids, tmask), R.kv_cache_speaker(wb, cfg, spk)
        plain = R.dit_forward(wb, cfg, x, t, tmask, smask, kvt, kvs)
        R.set_fp8_block_linears(True)
        try:
            want = R.dit_forward(wb, cfg, x, t, tmask, smask, kvt, kvs)
        finally:
            R.set_fp8_block_linears(False)
        e, effect = rms(got, want), rms(want, plain)
        print(f"C5 fp8 engine vs fake-quant restatement (d = {cfg.model_size}): rms {e:.3e}; fp8's own effect (restatement vs bf16) {effect:.3e}; "
              f"output rms {U.rms(want):.3f}")
        assert effect > 0 and e < 0.8 * effect + 2e-3 * U.rms(want), (e, effect)


@pytest.mark.statistical
def test_fp8_static_activation_scales_calibration_and_restatement(golden):
    """SURVEY 8f-4 "fp8 calibration" (C5): `fp8_calibration_start` / `_finish` record the largest dynamic row scale of the attention output
    and of the SwiGLU output per block; with those installed (`set_fp8_static_scales`) the attention epilogue and the SwiGLU tail write
    the e4m3 operands of wo / w2 themselves.  Pinned like the dynamic engine: at full width and on the tiny model one velocity prediction
    must be much closer to the oracle's restatement of THAT arithmetic (`set_fp8_block_linears(True, act_static=...)`: one calibrated
    scale for those two operands, saturating) than the restatement is to plain bf16; it stays close to the dynamic engine; clearing the
    scales gives the dynamic engine's bits back; a scale table of the wrong size fails loudly."""
    for cfg, S, T in ((WIDE1, 200, 40), (TINY, 32, 24)):
        wb = {k: v.bfloat16() for k, v in R.make_dit_weights(cfg, seed=0).items()}
        gen = torch.Generator().manual_seed(3)
        ids = torch.randint(1, 256, (1, T), generator=gen, dtype=torch.int32)
        tmask = torch.ones((1, T), dtype=torch.bool)
        spk, smask = torch.randn((1, 64, 80), generator=gen).bfloat16(), torch.ones((1, 64), dtype=torch.bool)
        x = torch.randn((1, S, 80), generator=gen).bfloat16()
        t = torch.full((1,), 0.75).bfloat16()
        m8 = E.EchoDiT(cfg, wb, dtype=torch.bfloat16, device=DEV, fp8=True)
        fwd = lambda: m8(x, t, tmask, smask, m8.get_kv_cache_text(ids, tmask), m8.get_kv_cache_speaker(spk, smask)).float().cpu()
        dyn = fwd()
        m8.fp8_calibration_start()
        during = fwd()
        scales = m8.fp8_calibration_finish(margin=1.0)
        assert torch.equal(during, dyn), "recording the maxima changed the forward"
        assert scales.shape == (cfg.num_layers, 2) and bool((scales > 0).all()) and bool(torch.isfinite(scales).all())
        m8.set_fp8_static_scales(scales)
        got = fwd()
        assert torch.equal(got, fwd()), "the static-scale forward is not reproducible"
        kvt, kvs = R.kv_cache_text(wb, cfg, ids, tmask), R.kv_cache_speaker(wb, cfg, spk)
        plain = R.dit_forward(wb, cfg, x, t, tmask, smask, kvt, kvs)
        act = {}
        for i in range(cfg.num_layers):
            act[f"blocks.{i}.attention.wo"] = float(scales[i, 0])
            act[f"blocks.{i}.mlp.w2"] = float(scales[i, 1])
        R.set_fp8_block_linears(True, act_static=act)
        try:
            want = R.dit_forward(wb, cfg, x, t, tmask, smask, kvt, kvs)
        finally:
            R.set_fp8_block_linears(False)
        e, effect, vs_dyn = rms(got, want), rms(want, plain), rms(got, dyn)
        print(f"C5 static activation scales (d = {cfg.model_size}): rms {e:.3e} vs the static restatement; fp8's own effect {effect:.3e}; "
              f"static vs dynamic engine {vs_dyn:.3e}; output rms {U.rms(want):.3f}; scales {scales.min():.3e} .. {scales.max():.3e}")
        # loose by construction (see test_fp8_engine_against_fake_quant_restatement): measured 0.47 / 0.50 of the effect (d = 2048 / 256), bound 1.5x that
        assert effect > 0 and e < 0.8 * effect + 2e-3 * U.rms(want), (e, effect)
        assert vs_dyn < 1.3 * effect + 2e-3 * U.rms(want), (vs_dyn, effect)            # measured 0.76 / 0.83
        # half the calibrated range: the operands saturate, the forward stays finite and moves away
        m8.set_fp8_static_scales(scales * 0.25)
        sat = fwd()
        assert bool(torch.isfinite(sat).all()) and rms(sat, got) > 0
        m8.set_fp8_static_scales(None)
        assert torch.equal(fwd(), dyn), "clearing the static scales did not restore the dynamic path"
        with pytest.raises(L.EchoHipError):
            m8.set_fp8_static_scales(torch.ones((cfg.num_layers + 1, 2)))
    # the scales travel as a small JSON file next to a checkpoint
    import tempfile
    from echo_tts_amd import weights as Wt
    with tempfile.TemporaryDirectory() as td:
        Wt.save_fp8_scales(os.path.join(td, "s.json"), scales, {"note": "test"})
        assert torch.equal(Wt.load_fp8_scales(os.path.join(td, "s.json")), scales)


def test_voice_cloning_pipeline_from_audio(golden, tiny_models):
    """The whole reference flow of handler.py:750-758 on the tiny models, starting from speaker AUDIO: DAC encode ->
    get_speaker_latent_and_mask -> sampler -> ae_decode -> crop, against the oracle run stage by stage on the same inputs."""
    from functools import partial
    from echo_tts_amd import handler as H
    from echo_tts_amd import inference as inf
    m = tiny_models["f32"]
    dw = R.make_dac_weights(TINY_DAC, 0)
    dw.update(R.make_dac_encoder_weights(TINY_DAC, 0))
    dac = E.DAC(TINY_DAC, dw, device=DEV)
    pca = R.make_pca(TINY_DAC, 80, 0)
    st = E.PCAState(pca.pca_components, pca.pca_mean, pca.latent_scale)
    voice = R.make_test_audio(2048 * 16 + 700, seed=21)[0]               # (1, n): 16 whole frames + a partial one
    fn = partial(E.sample_euler_cfg_independent_guidances, sequence_length=32, **SAMPLER_CASES["cfg_default"])
    audio, norm = E.sample_pipeline(m, dac, st, fn, "Hello world.", voice, 3, pad_to_max_text_length=64)
    assert norm == "[S1] Hello world." and bool(torch.isfinite(audio).all())
    # oracle, same stages (the 30 s zero-padded chunk gives 640 pre_module positions: more than the tiny block_size, so the
    # oracle's rope cache is built longer; its values do not depend on the cache length)
    import dataclasses
    spk, smask = R.get_speaker_latent_and_mask(dw, dataclasses.replace(TINY_DAC, post_block_size=4096), pca, voice)
    assert spk.shape == (1, 16, 80)
    got_spk, got_mask = E.get_speaker_latent_and_mask(dac, st, voice.to(DEV))
    assert torch.equal(got_mask.cpu(), smask) and rms(got_spk, spk) < 1e-5 * max(1.0, U.rms(spk))
    ids, tmask = inf.get_text_input_ids_and_mask(["Hello world."], max_length=64)
    x0 = torch.randn((1, 32, 80), device=DEV, dtype=torch.float32, generator=torch.Generator(device=DEV).manual_seed(3)).cpu()
    w = R.make_dit_weights(TINY, seed=0)
    lat = R.sample_euler(w, TINY, torch.float32, spk, smask, ids, tmask, rng_seed=3, sequence_length=32, x_init=x0,
                         **SAMPLER_CASES["cfg_default"])
    wav = R.ae_decode(dw, TINY_DAC, pca, lat)
    wav = wav[..., : R.find_flattening_point(lat[0]) * 2048]
    assert wav.shape == audio.shape and rms(audio, wav) < WAV_TOL
    out = H.synthesize({"text": "Hello world. Again.", "parameters": {"num_steps": 4, "sequence_length": 32}}, m, dac, st,
                       speaker_audio=voice)
    assert "error" not in out, out.get("traceback")


@pytest.mark.parametrize("case,opts,cont", [("plain", "cfg_default", False), ("cont_opts", "all_options", True)])
def test_blockwise_sampler_bf16_within_reference_noise(golden, tiny_models, case, opts, cont):
    g, m = golden, tiny_models["bf16"]
    xi = [g[f"tiny.blk_x{j}"] for j in range(3)]
    lat = E.sample_blockwise_euler_cfg_independent_guidances(
        m, g["tiny.spk"], g["tiny.smask"].bool(), g["tiny.ids"], g["tiny.tmask"].bool(), rng_seed=0, block_sizes=[16, 8, 8],
        continuation_latent=g["tiny.blk_cont"] if cont else None, x_inits=xi, **SAMPLER_CASES[opts])
    e = rms(lat, g[f"tiny.f32.blockwise.{case}"])
    assert e < _bf16_budget(g, f"tiny.bf16.blockwise.{case}", f"tiny.f32.blockwise.{case}"), e


# ------------------------------------------------------------------------------------------------ SURVEY §8f rows
def test_voice_cache_is_bit_identical_and_shared_voice_equals_replicated(golden, tiny_models):
    """Per-voice cache (§8f-1; the reference re-encodes the voice per chunk: handler.py:750-758, model.py:615-621):
    * a captured voice bound instead of get_kv_cache_speaker gives BIT-identical latents, repeatedly (the bound copy is private,
      so the in-place speaker-KV scaling of one request cannot leak into the cached voice), on both engines;
    * one voice with batch 1 shared by two text rows (stride-0 addressing) equals the same voice replicated per row."""
    g = golden
    ids, tm = g["tinyb2.ids"], g["tinyb2.tmask"].bool()
    spk1, sm1 = g["tinyb2.spk"][:1], g["tinyb2.smask"][:1].bool()
    x0 = g["tinyb2.x0"]
    for name in ("f32", "bf16"):
        m = tiny_models[name]
        for case in ("cfg_default", "all_options"):
            kw = SAMPLER_CASES[case]
            def run(spk=spk1, sm=sm1, **extra):
                return E.sample_euler_cfg_independent_guidances(m, spk, sm, ids, tm, rng_seed=0, sequence_length=32, x_init=x0, **kw, **extra)
            fresh = run()
            voice = m.capture_voice(m.get_kv_cache_speaker(spk1.to(m.dtype), sm1))
            assert voice.nbytes > 0
            a, b = run(speaker_kv=voice), run(speaker_kv=voice)
            assert torch.equal(fresh, a) and torch.equal(a, b), (name, case)
            rep = run(spk=spk1.expand(2, -1, -1).contiguous(), sm=sm1.expand(2, -1).contiguous())
            assert torch.equal(fresh, rep), (name, case)
            voice.close()
    # a voice from another model is refused (different precision)
    v = tiny_models["f32"].capture_voice(tiny_models["f32"].get_kv_cache_speaker(spk1, sm1))
    with pytest.raises(RuntimeError, match="different model"):
        tiny_models["bf16"].bind_voice(v)


def test_handler_entry_points_with_voice_cache(golden, tiny_models):
    """`handler(job)` / `_synthesize(job_input, job_id)` (reference handler.py:682-816): validation messages, seed fallback to the
    top-level key, chunking switch, response shape; a registered `speaker_voice` is encoded once and served from the per-voice cache
    afterwards with identical audio."""
    from echo_tts_amd import handler as H
    m = tiny_models["f32"]
    dac = E.DAC(TINY_DAC, R.make_dac_weights(TINY_DAC, 0), device=DEV)
    pca = R.make_pca(TINY_DAC, 80, 0)
    st = E.PCAState(pca.pca_components, pca.pca_mean, pca.latent_scale)
    H.configure(m, dac, st, voices={"alice": (golden["tiny.spk"], golden["tiny.smask"].bool())})
    assert H.handler({"id": "1", "input": {}}) == {"error": "Missing or invalid 'text' field (expected string)"}
    assert H._synthesize({"text": 5}) == {"error": "Missing or invalid 'text' field (expected string)"}
    assert H._synthesize({"text": "   "}) == {"error": "Text cannot be empty"}
    assert H._synthesize({"text": "x" * 4001}) == {"error": "Text too long: 4001 characters (max 4000)"}
    assert H._synthesize({"text": "Hi.", "speaker_voice": "bob"}) == {"error": "speaker_voice 'bob' not found"}
    assert H._synthesize({"action": "health_check"})["status"] == "healthy"
    text = "First sentence here. Second sentence follows, with a clause. Third one, which is a little longer than the others!"
    p = {"num_steps": 4, "sequence_length": 32, "target_duration_seconds": 3.0}
    job = {"id": "j", "input": {"text": text, "speaker_voice": "alice", "seed": 11, "parameters": p}}
    r1 = H.handler(job)
    assert r1.get("status") == "completed", r1
    assert r1["metadata"]["seed"] == 11 and r1["metadata"]["sample_rate"] == 44100 and r1["metadata"]["chunks"] >= 2
    assert abs(r1["metadata"]["duration"] - r1["audio"].shape[-1] / 44100) < 1e-9 and r1["audio"].dim() == 2
    misses = H._State.cache.misses
    r2 = H.handler(job)
    assert H._State.cache.misses == misses and H._State.cache.hits >= 1
    assert torch.equal(r1["audio"], r2["audio"])
    # parameters.seed wins over the top-level seed; max_chars_per_chunk <= 0 and junk switch chunking off / fall back to 300
    r3 = H._synthesize({"text": text, "seed": 11, "parameters": dict(p, seed=12, max_chars_per_chunk=0)})
    assert r3["metadata"]["seed"] == 12 and r3["metadata"]["chunks"] == 1
    r4 = H._synthesize({"text": text, "parameters": dict(p, max_chars_per_chunk="junk", num_steps=2)})
    assert r4.get("status") == "completed" and r4["metadata"]["chunks"] == len(H.chunk_text_for_audio(text, 300, 3.0))
    assert H._build_sample_fn({"sequence_length": None}).keywords["sequence_length"] is None      # the sampler reads None as 640
    # the batched, cached, device-post-processed path equals the reference-shaped sequential path (one call per chunk, torch post-processing)
    seq = H.synthesize({"text": text, "parameters": dict(p, seed=11, max_chunk_batch=1)}, m, dac, st,
                       speaker_latent=golden["tiny.spk"], speaker_mask=golden["tiny.smask"].bool())
    assert seq["audio"].shape == r1["audio"].shape and rms(seq["audio"], r1["audio"]) <= WAV_TOL


def test_streaming_decode_equals_whole_utterance_decode():
    """§8f-3: DACStream (causal chunked decode with carried left context) against the whole-utterance decode, tiny and
    FULL-SIZE Fish S1-DAC: concatenated chunks == one-shot decode within 1e-6 RMS (and within 1e-4 of the signal RMS)."""
    for cfg, T, blocks in ((TINY_DAC, 40, (8, 8, 4, 20)), (R.DacConfig(), 64, (16, 16, 16, 16))):
        w = R.make_dac_weights(cfg, 0)
        dac = E.DAC(cfg, w, device=DEV)
        pca = R.make_pca(cfg, 80, 0)
        st = E.PCAState(pca.pca_components, pca.pca_mean, pca.latent_scale)
        lat = torch.randn((1, T, 80), generator=torch.Generator().manual_seed(2))
        whole = E.ae_decode(dac, st, lat)
        from echo_tts_amd.autoencoder import DACStream
        stream = DACStream(dac, st)
        assert 9 <= stream.context <= 16
        parts, pos = [], 0
        for n in blocks:
            parts.append(stream.push(lat[:, pos:pos + n]))
            assert parts[-1].shape == (1, 1, n * 2048)
            pos += n
        got = torch.cat(parts, dim=-1)
        e = rms(got, whole)
        print(f"streaming decode ({T} frames in {len(blocks)} chunks, context {stream.context}): rms {e:.3e} (signal {U.rms(whole):.3e})")
        assert got.shape == whole.shape and e < 1e-6 and e < 1e-4 * U.rms(whole)
        # a too-short context must be visible (the test can fail): context 0 differs right after every chunk boundary
        bad = DACStream(dac, st, context=0)
        bad.push(lat[:, :blocks[0]])
        tail = bad.push(lat[:, blocks[0]:blocks[0] + blocks[1]])
        assert rms(tail, whole[..., blocks[0] * 2048:(blocks[0] + blocks[1]) * 2048]) > 1e-5


def test_blockwise_streaming_api(golden, tiny_models):
    """sample_blockwise_stream yields every block as it is sampled and reproduces the reference-shaped blockwise sampler bit for
    bit; stream_audio_blockwise turns the blocks into audio chunk by chunk, equal to decoding the final latent at once."""
    from echo_tts_amd import inference_blockwise as IB
    g, m = golden, tiny_models["f32"]
    args = (g["tiny.spk"], g["tiny.smask"].bool(), g["tiny.ids"], g["tiny.tmask"].bool())
    kw = dict(SAMPLER_CASES["cfg_default"])
    blocks = [8, 8, 16]
    whole = IB.sample_blockwise_euler_cfg_independent_guidances(m, *args, rng_seed=4, block_sizes=blocks, **kw)
    seen, starts = [], []
    for start, blk, prefix in IB.sample_blockwise_stream(m, *args, rng_seed=4, block_sizes=blocks, **kw):
        starts.append(start)
        seen.append(blk.clone())
        assert prefix.shape[1] == start + blk.shape[1]
    assert starts == [0, 8, 16] and torch.equal(torch.cat(seen, dim=1), whole)
    dac = E.DAC(TINY_DAC, R.make_dac_weights(TINY_DAC, 0), device=DEV)
    pca = R.make_pca(TINY_DAC, 80, 0)
    st = E.PCAState(pca.pca_components, pca.pca_mean, pca.latent_scale)
    chunks = list(IB.stream_audio_blockwise(m, dac, st, *args, rng_seed=4, block_sizes=blocks, **kw))
    assert [c.shape[-1] for c in chunks] == [b * 2048 for b in blocks]
    assert rms(torch.cat(chunks, dim=-1), E.ae_decode(dac, st, whole)) < 1e-6


def test_post_processing_kernels_against_reference_kats(golden):
    """§8f-2: the HIP post-processing kernels (csrc/postproc.hip) against the known answers the reference produced
    (tests/golden/host.safetensors: find_flattening_point on three latents, crossfade_chunks and normalize_chunk_boundaries on
    three chunks) and against the torch form on edge cases (short chunks whose overlap is len // 4, silent / loud tails,
    a window longer than the chunk, zero-length results)."""
    from echo_tts_amd import handler as H
    from echo_tts_amd import inference as inf
    want = golden["__meta__"]["host"]["flattening"]
    lats = torch.stack([golden[f"flat.{i}"] for i in range(3)]).to(DEV)
    assert inf.find_flattening_points(lats) == want
    assert [inf.find_flattening_point(lats[i]) for i in range(3)] == want
    assert inf.find_flattening_points(torch.zeros((1, 30, 80), device=DEV)) == [0]
    assert inf.find_flattening_points(torch.ones((1, 30, 80), device=DEV) * 3) == [30]
    noisy = torch.randn((2, 640, 80), generator=torch.Generator().manual_seed(1))
    noisy[1, 200:] = 0
    assert inf.find_flattening_points(noisy.to(DEV)) == [inf.find_flattening_point(noisy[0]), inf.find_flattening_point(noisy[1])] == [640, 200]
    a, b, c = (golden[k].to(DEV) for k in ("post.a", "post.b", "post.c"))
    xf = H.crossfade_chunks_device([a, b, c], 4410)
    assert xf.shape == golden["post.crossfade"].shape and float((xf.cpu() - golden["post.crossfade"]).abs().max()) < 1e-6
    nb = H.normalize_chunk_boundaries_device([a, b, c], min_silence_samples=2000)
    assert nb.shape == golden["post.normalize"].shape and float((nb.cpu() - golden["post.normalize"]).abs().max()) < 1e-6
    gen = torch.Generator().manual_seed(3)
    cases = []
    for lens in ((50, 7, 3, 400), (4410 * 4, 100, 4410 * 8), (1, 1), (30000, 20000)):
        chunks = [torch.randn((1, n), generator=gen) * 0.2 for n in lens]
        chunks[0][..., -min(lens[0], 9):] = 0.001                   # a quiet tail shorter than the minimum silence
        cases.append(chunks)
    loud = [torch.ones((1, 5000)) * 0.5, torch.zeros((1, 3000)), torch.ones((1, 100)) * 0.5]     # no silence / all silence
    cases.append(loud)
    for chunks in cases:
        dev = [x.to(DEV) for x in chunks]
        for ms in (22050, 2000, 4):
            ref = H.normalize_chunk_boundaries([x.clone() for x in chunks], min_silence_samples=ms)
            got = H.normalize_chunk_boundaries_device(dev, min_silence_samples=ms)
            assert got.shape == ref.shape and float((got.cpu() - ref).abs().max()) < 1e-6, ([x.shape[-1] for x in chunks], ms)
        ref = H.crossfade_chunks([x.clone() for x in chunks], 4410)
        got = H.crossfade_chunks_device(dev, 4410)
        assert got.shape == ref.shape and float((got.cpu() - ref).abs().max()) < 1e-6
        rows = [x.reshape(-1) for x in dev]
        assert H.trailing_quiet_device(rows, 2 * 2000, 0.01) == [H._trailing_quiet(x, min(x.shape[-1], 4000), 0.01) for x in chunks]


def test_checkpoint_loaders_round_trip(tmp_path, golden):
    """§8f-4: load_model_from_path / load_fish_ae_from_path / load_pca_state_from_path read safetensors files with the reference's
    key layout (inference.py:14-113 read the same files from the Hugging Face cache) and give the same outputs as the in-memory
    constructors; workspace sizing (SURVEY §8b `echo_workspace_bytes`) grows the caches up front."""
    import safetensors.torch as sft
    from echo_tts_amd import inference as inf
    w = R.make_dit_weights(TINY, seed=0)
    sft.save_file({k: v.contiguous() for k, v in w.items()}, str(tmp_path / "dit.safetensors"))
    dw = R.make_dac_weights(TINY_DAC, 0)
    sft.save_file({k: v.contiguous() for k, v in dw.items()}, str(tmp_path / "dac.safetensors"))
    pca = R.make_pca(TINY_DAC, 80, 0)
    sft.save_file({"pca_components": pca.pca_components.contiguous(), "pca_mean": pca.pca_mean.contiguous(),
                   "latent_scale": torch.tensor(pca.latent_scale)}, str(tmp_path / "pca.safetensors"))
    m = inf.load_model_from_path(str(tmp_path / "dit.safetensors"), dtype=torch.float32, config=TINY)
    dac = inf.load_fish_ae_from_path(str(tmp_path / "dac.safetensors"), config=TINY_DAC)
    st = inf.load_pca_state_from_path(str(tmp_path / "pca.safetensors"))
    g = golden
    args = (g["tiny.spk"], g["tiny.smask"].bool(), g["tiny.ids"], g["tiny.tmask"].bool())
    before = m.workspace_bytes()
    m.reserve_workspace(1, 32, g["tiny.ids"].shape[1], g["tiny.spk"].shape[1])
    grown = m.workspace_bytes()
    assert grown > before
    lat = E.sample_euler_cfg_independent_guidances(m, *args, rng_seed=0, sequence_length=32, x_init=g["tiny.x0"], **SAMPLER_CASES["cfg_default"])
    assert rms(lat, g["tiny.f32.euler.cfg_default"]) < LAT_TOL
    wav = E.ae_decode(dac, st, lat)
    assert rms(wav, R.ae_decode(dw, TINY_DAC, pca, lat.cpu())) < WAV_TOL
    m2 = inf.load_model_from_path(str(tmp_path / "dit.safetensors"), dtype=torch.float32, config=TINY, delete_blockwise_modules=True)
    assert not m2.has_latent_encoder and m.has_latent_encoder


@pytest.mark.parametrize("dname,dt", [("f32", torch.float32), ("bf16", torch.bfloat16)])
def test_forward_with_per_row_timesteps(golden, tiny_models, dname, dt):
    """model.py:563-604 takes any t (R,): rows of one forward may carry different timesteps (model.py:27-43 embeds each).  The engine
    groups rows by timestep (`echo_dit_forward_t`: one modulation table per distinct t, the AdaLN launches per run of equal rows).
    fp32 engine vs the oracle with the same per-row t: the north-star tolerance; bf16 engine: the bf16 budget of the tiny forward; and
    a mixed-t call must reproduce, row by row, the bits of uniform-t calls (the t-independent launches see the same rows)."""
    g, m = golden, tiny_models[dname]
    w = R.make_dit_weights(TINY, seed=0)
    tag = "tinyb2"
    ids, tmask = g[f"{tag}.ids"], g[f"{tag}.tmask"].bool()
    spk, smask = g[f"{tag}.spk"], g[f"{tag}.smask"].bool()
    B, S = ids.shape[0], 32
    rows = 3 * B
    x = torch.randn((rows, S, 80), generator=torch.Generator().manual_seed(9))
    t = torch.tensor([0.75, 0.25, 0.75, 0.5, 0.5, 0.125])[:rows]                # bf16-exact values; rows 0 / 2 share one, 3 / 4 another
    tm3 = torch.cat([tmask, torch.zeros_like(tmask), tmask], 0)
    sm3 = torch.cat([smask, smask, torch.zeros_like(smask)], 0)
    kvt, kvs = m.get_kv_cache_text(ids, tmask), m.get_kv_cache_speaker(spk, smask)
    kvt3, kvs3 = _concat_kv_caches(kvt, kvt, kvt), _concat_kv_caches(kvs, kvs, kvs)
    got = m(x.to(dt), t.to(dt), tm3, sm3, kvt3, kvs3).float().cpu()
    ow = w if dt == torch.float32 else {k: v.bfloat16() for k, v in w.items()}
    okt, oks = R.kv_cache_text(ow, TINY, ids, tmask), R.kv_cache_speaker(ow, TINY, spk.to(dt))
    cat3 = lambda c: [(torch.cat([k, k, k], 0), torch.cat([v, v, v], 0)) for k, v in c]
    want = R.dit_forward(w, TINY, x, t, tm3, sm3, cat3(R.kv_cache_text(w, TINY, ids, tmask)), cat3(R.kv_cache_speaker(w, TINY, spk)))
    e = rms(got, want)
    if dt == torch.float32:
        assert e < 1e-4 * max(1.0, U.rms(want)), e
    else:
        wantb = R.dit_forward(ow, TINY, x.to(dt), t.to(dt), tm3, sm3, cat3(okt), cat3(oks))
        assert e < 1.5 * rms(wantb, want) + 1e-3, (e, rms(wantb, want))
    # row by row against uniform-t forwards of the same rows (the per-segment GEMMs have another M, i.e. possibly another tile plan
    # and summation order: equal up to that, not bit for bit)
    for tv in (0.75, 0.5):
        uni = m(x.to(dt), torch.full((rows,), tv).to(dt), tm3, sm3, kvt3, kvs3).float().cpu()
        for r in range(rows):
            if float(t[r]) == tv:
                assert rms(uni[r], got[r]) < (1e-6 if dt == torch.float32 else 2e-2) * U.rms(uni[r]), (tv, r)
            else:
                assert rms(uni[r], got[r]) > 1e-4 * U.rms(uni[r]), (tv, r)           # another timestep really is another output (tiny model: ~3e-3)
    # a scalar-like t (one entry) still means "all rows"
    one = m(x.to(dt), torch.tensor([0.5]).to(dt), tm3, sm3, kvt3, kvs3).float().cpu()
    assert torch.equal(one, m(x.to(dt), torch.full((rows,), 0.5).to(dt), tm3, sm3, kvt3, kvs3).float().cpu())
    with pytest.raises(ValueError):
        m(x.to(dt), torch.tensor([0.5, 0.25]).to(dt), tm3, sm3, kvt3, kvs3)


def _forward3(m, dt, x, tval, ids, tmask, spk, smask, before_forward=None):
    """One CFG-shaped forward through the engine: rows = [cond | text-uncond | speaker-uncond] x B (inference.py:474-475).
    `before_forward()` runs between the KV encodes and the EchoDiT forward (test instruments are armed there)."""
    B = ids.shape[0]
    kvt, kvs = m.get_kv_cache_text(ids, tmask), m.get_kv_cache_speaker(spk, smask)
    tm3 = torch.cat([tmask, torch.zeros_like(tmask), tmask], 0)
    sm1 = smask.expand(B, -1)
    sm3 = torch.cat([sm1, sm1, torch.zeros_like(sm1)], 0)
    if before_forward is not None:
        before_forward()
    return m(x.to(dt), torch.full((3 * B,), tval).to(dt), tm3, sm3, _concat_kv_caches(kvt, kvt, kvt), kvs).float().cpu()


def test_full_depth_forward_budgets_have_teeth(golden, full_size):
    """VERDICT round 2, "give the full-depth bf16 tests teeth".  Measured first (CPU oracle, round 3): what made the old full-depth
    budgets vacuous is NOT 24 layers of accumulated rounding - the reference's bf16 path rounds t to bf16 before the timestep
    embedding (inference.py:488-489; bf16(0.666) moves the phases by up to 2 rad), and that alone is its ~1 RMS distance from its
    fp32 run.  At a bf16-exact t the same 24-layer forward is 0.024 from fp32 on outputs of RMS 0.89 (0.017 with residual branches
    scaled by 1 / sqrt(2 L): conditioning is not the issue, so the seeded N(0, 0.02) recipe stays).  Hence ONE full-depth velocity
    prediction at t = 0.75 with the production kernels, at C2's shape (3 CFG rows x 640 latents = M 1920, text 436 of 768, 2560
    speaker latents) and at the bench's batch-24 shape (72 rows = M 46080, mixed preset lengths, one shared voice):
      (a) the fp32 engine equals PyTorch-ROCm eager fp32 (the oracle's ops on the MI355X) to 1e-4 relative;
      (b) the bf16 engine - single call and batch-24 call - is no farther from the fp32 engine than 1.5 x PyTorch-ROCm's own eager
          bf16 run is from eager fp32, overall and for its worst row: a budget 40x tighter than the 1.5 x 1.04 of round 2;
      (c) measured, not asserted: utterance 0's rows inside the batch-24 call (every linear on gemm_pp_kernel) against the same rows
          through the single call (wo / w2 on other tile kernels: other fp32 summation orders, same rounding points) differ by 1.7e-2 -
          as much as the engine differs from eager bf16: 24 layers of bf16 amplify ANY summation-order change to the level of the
          floor, so no full-depth bf16 DISTANCE can be tighter than (b); what can be tighter is an exact property:
      (d) ROW-PERMUTATION EQUIVARIANCE, bit for bit: the batch-24 forward with its utterances rotated by 7 must return exactly the
          rotated outputs.  Every GEMM of that call runs on the persistent ping-pong kernel, whose per-element arithmetic (K order, MFMA
          chain) does not depend on where a row sits in the tile walk; attention and the row kernels work per row.  A wrong entry in the
          tile walk, a tile computed from a neighbour's operands or written to the wrong place breaks it - however small the error;
      (e) TEETH: with ONE 256 x 256 output tile of ONE wo / w2 launch of block 12 negated (`echo_debug_corrupt_tile`; these launches write
          the residual stream, so the tile holds x for tokens 256..511 of the call's first row) the forward moves that row by 0.42-0.43
          (measured) - 12x outside budget (b) - and breaks (d); the old budget of 1.5 x 1.04 would have let it pass."""
    fs = full_size
    cfg = fs["cfg"]
    S, B = 640, 24
    g = torch.Generator().manual_seed(4242)
    lens8 = golden["__meta__"]["host"]["preset_token_lengths"][:8]
    lens = [436] + [lens8[b % 8] for b in range(1, B)]
    ids = torch.zeros((B, 768), dtype=torch.int32)
    tmask = torch.zeros((B, 768), dtype=torch.bool)
    for b, n in enumerate(lens):
        ids[b, 1:n] = torch.randint(32, 127, (n - 1,), generator=g, dtype=torch.int32)
        tmask[b, :n] = True
    spk, smask = torch.randn((1, 2560, 80), generator=g), torch.ones((1, 2560), dtype=torch.bool)
    x = torch.randn((3 * B, S, 80), generator=g)
    rows0 = [0, B, 2 * B]                                         # utterance 0's cond / text-uncond / speaker-uncond rows
    x1, ids1, tm1 = x[rows0], ids[:1], tmask[:1]
    mb, mf = fs["bf16"], fs["f32"]
    single = lambda m, dt: _forward3(m, dt, x1, 0.75, ids1, tm1, spk, smask)
    batched = lambda m, dt: _forward3(m, dt, x, 0.75, ids, tmask, spk, smask)
    ref1 = single(mf, torch.float32)
    got1 = single(mb, torch.bfloat16)
    got24 = batched(mb, torch.bfloat16)
    # PyTorch-ROCm eager on utterance 0's three rows: the fp32 anchor and the reference's own bf16 noise floor at this shape
    dev = lambda t_: t_.to(DEV)
    eager = {}
    for name, ww, dt in (("f32", fs["w"], torch.float32), ("bf16", fs["wb"], torch.bfloat16)):
        wd = {k: v.to(DEV) for k, v in ww.items()}
        with torch.inference_mode():
            kvt = R.kv_cache_text(wd, cfg, dev(ids1), dev(tm1))
            kvs = R.kv_cache_speaker(wd, cfg, dev(spk).to(dt))
            c3 = lambda c: [(k.expand(3, -1, -1, -1), v.expand(3, -1, -1, -1)) for k, v in c]
            tm3 = torch.cat([tm1, torch.zeros_like(tm1), tm1], 0)
            sm3 = torch.cat([smask, smask, torch.zeros_like(smask)], 0)
            eager[name] = R.dit_forward(wd, cfg, dev(x1).to(dt), torch.full((3,), 0.75, device=DEV).to(dt), dev(tm3), dev(sm3), c3(kvt), c3(kvs)).float().cpu()
        del wd, kvt, kvs
        torch.cuda.empty_cache()
    row_rms = lambda a, b: (a - b).pow(2).mean(dim=(1, 2)).sqrt()
    out_rms = U.rms(eager["f32"])
    anchor = rms(ref1, eager["f32"])
    floor, floor_row = rms(eager["bf16"], eager["f32"]), float(row_rms(eager["bf16"], eager["f32"]).max())
    assert anchor < 1e-4 * out_rms, (anchor, out_rms)                                             # (a)
    e1, e1_row = rms(got1, ref1), float(row_rms(got1, ref1).max())
    e24, e24_row = rms(got24[rows0], ref1), float(row_rms(got24[rows0], ref1).max())
    self_row = float(row_rms(got24[rows0], got1).max())
    print(f"full depth, t = 0.75, C2 shape and batch 24: output rms {out_rms:.3f}; fp32 engine vs eager fp32 {anchor:.2e}; PyTorch-ROCm bf16 vs fp32 "
          f"{floor:.3e} (worst row {floor_row:.3e}); bf16 engine vs fp32 engine: single call {e1:.3e} (worst row {e1_row:.3e}), batch-24 rows {e24:.3e} "
          f"({e24_row:.3e}); batch-24 rows vs single call, worst row {self_row:.3e}; bf16 engine vs eager bf16 {rms(got1, eager['bf16']):.3e}")
    assert 3e-3 * out_rms < floor < 6e-2 * out_rms, (floor, out_rms)                               # the floor is where a budget means something
    assert e1 < 1.5 * floor and e1_row < 1.5 * floor_row, (e1, floor, e1_row, floor_row)          # (b)
    assert e24 < 1.5 * floor and e24_row < 1.5 * floor_row, (e24, floor, e24_row, floor_row)
    assert torch.equal(got1, single(mb, torch.bfloat16)) and torch.equal(got24, batched(mb, torch.bfloat16))
    # (d) row-permutation equivariance of the batch-24 call, bit for bit
    perm = torch.tensor([(b + 7) % B for b in range(B)])
    perm3 = torch.cat([perm, perm + B, perm + 2 * B])
    permuted = lambda: _forward3(mb, torch.bfloat16, x[perm3], 0.75, ids[perm], tmask[perm], spk, smask)
    got24p = permuted()
    assert torch.equal(got24p, got24[perm3]), f"not permutation-equivariant: rms {rms(got24p, got24[perm3]):.3e}"
    # (e) teeth: eligible launches of a forward are in_proj (1), then wo (2 + 2 l) and w2 (3 + 2 l) of block l: 26 / 27 = block 12's
    for nth in (26, 27):
        try:       # armed AFTER the text / speaker encoders (their wo / w2 launches are eligible too)
            bad = _forward3(mb, torch.bfloat16, x, 0.75, ids, tmask, spk, smask,
                            before_forward=lambda: L.check(mb._lib.echo_debug_corrupt_tile(mb._ctx, nth)))
        finally:
            L.check(mb._lib.echo_debug_corrupt_tile(mb._ctx, 0))
        moved = row_rms(bad, got24)
        hit = [int(i) for i in torch.nonzero(moved > 0).flatten()]
        vs_ref = float(row_rms(bad[rows0], ref1).max())
        print(f"  one negated 256 x 256 tile in eligible GEMM launch {nth} ({'wo' if nth == 26 else 'w2'} of block 12) of the batch-24 forward: rows moved {hit} "
              f"by {float(moved.max()):.3e}; utterance 0's rows vs the fp32 engine, worst row {vs_ref:.3e} (budget (b) {1.5 * floor_row:.3e}: "
              f"{'still inside' if vs_ref < 1.5 * floor_row else 'outside'})")
        assert hit == [0], hit                                     # tokens 256..511 of the call's first row (utterance 0, cond), nothing else
        assert vs_ref > 1.5 * floor_row                            # budget (b) rejects the corrupted forward ...
        assert not torch.equal(got24p, bad[perm3])                 # ... and so does the equivariance check (d)
        assert vs_ref < 1.5 * 1.04                                 # ... while round 2's budget (1.5 x the t-rounding distance) would have passed it
    assert torch.equal(got24p, permuted())                                                            # the instrument disarmed itself


def test_unlisted_gemm_shapes_are_bit_reproducible_across_processes(tmp_path):
    """a seed is reproducible in every process.  Shapes the shipped plan table does not list used to be tuned from first-use
    timings (box- and run-dependent tile plans, i.e. summation orders); they now take a fixed rule.  Two FRESH processes run one
    full-width forward at a shape no table entry covers (M = 3 x 217 rows, 53 text tokens) and must produce identical bits; a third
    with the timing tuner switched on explicitly is only required to stay inside the bf16 noise."""
    import subprocess
    import sys
    script = tmp_path / "one_forward.py"
    script.write_text(
        "import sys, torch\n"
        f"sys.path.insert(0, {repr(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))})\n"
        "import echo_tts_amd as E\n"
        "from oracle import echo_ref as R\n"
        "from tests.golden_defs import WIDE1\n"
        "w = {k: v.bfloat16() for k, v in R.make_dit_weights(WIDE1, seed=0).items()}\n"
        "m = E.EchoDiT(WIDE1, w, dtype=torch.bfloat16, device='cuda:0')\n"
        "g = torch.Generator().manual_seed(11)\n"
        "ids = torch.randint(1, 256, (1, 53), generator=g, dtype=torch.int32); tm = torch.ones((1, 53), dtype=torch.bool)\n"
        "spk = torch.randn((1, 68, 80), generator=g); sm = torch.ones((1, 68), dtype=torch.bool)\n"
        "x = torch.randn((3, 217, 80), generator=g)\n"
        "kt, ks = m.get_kv_cache_text(ids, tm), m.get_kv_cache_speaker(spk, sm)\n"
        "from echo_tts_amd.inference import _concat_kv_caches as cat\n"
        "tm3 = torch.cat([tm, torch.zeros_like(tm), tm], 0); sm3 = torch.cat([sm, sm, torch.zeros_like(sm)], 0)\n"
        "y = m(x.bfloat16(), torch.full((3,), 0.75).bfloat16(), tm3, sm3, cat(kt, kt, kt), ks).float().cpu()\n"
        "torch.save(y, sys.argv[1])\n")
    outs = []
    for i, extra in enumerate(({}, {}, {"ECHO_GEMM_TUNE": "1"})):
        env = dict(os.environ, **extra)
        env.pop("ECHO_GEMM_PLANS_SAVE", None)
        if not extra:
            env.pop("ECHO_GEMM_TUNE", None)
        out = tmp_path / f"y{i}.pt"
        r = subprocess.run([sys.executable, str(script), str(out)], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(torch.load(out, weights_only=True))
    assert torch.equal(outs[0], outs[1]), f"two fresh processes differ: rms {rms(outs[0], outs[1]):.3e}"
    assert rms(outs[0], outs[2]) < 3e-2 * U.rms(outs[0])


def test_bench_spawns_and_measures_two_ranks_end_to_end():
    """SURVEY 8e / VERDICT round 2 item 4: `python bench.py --gpus 2` with no torchrun environment must start two ranks itself (before it
    touches the GPU) and report them.  Rehearsed on the one GPU of this box: gloo instead of RCCL, both ranks on device 0 - the
    weight broadcast, the barriers, the max-over-ranks timing, the C3 leg (8 mixed-length units sharded round-robin, one sampler
    call per rank, ordered gather on rank 0) are the code the 8-GPU job runs."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--force-device", "0", "--steps", "1",
           "--warmup", "0", "--batch", "4", "--concurrency", "1", "--no-cpu-baseline", "--no-eager-baseline", "--no-c5"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["distributed"]["ranks_seen"] == 2 and out["distributed"]["backend"] == "gloo"
    assert out["distributed"]["weight_broadcast"]["bytes"] > 4e9 and out["distributed"]["weight_broadcast"]["GB_per_s"] > 0
    assert out["value"] > 0 and out["scaling"] == "weak" and out["per_gpu"] * 2 == pytest.approx(out["value"], rel=1e-3)
    c3 = out["c3"]
    assert c3["units"] == 8 and c3["ranks"] == 2 and c3["gathered_in_order_on_rank0"] is True and c3["value"] > 0
    assert out["single_request"]["roofline"]["frac"] > 0
    # a world size that does not match --gpus is refused
    env2 = dict(env, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r2 = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--no-roofline"], env=env2, capture_output=True, text=True, timeout=300)
    assert r2.returncode != 0 and "WORLD_SIZE" in (r2.stderr + r2.stdout)


def test_c4_blockwise_at_its_stated_size_with_crossfade(golden):
    """BASELINE config C4 at its stated geometry: `sample_blockwise` with block_sizes = [160] * 4 (640 latents per chunk; every block
    attends to the latent-prefix KV of the blocks before it, positions 4 i < start_pos, inference_blockwise.py:59-121), a long prompt
    cut into two text chunks by the handler's chunker, each chunk sampled with its own seed, decoded and cross-faded (handler.py:126-170).
    Full-width model with one layer per stack (WIDE1, incl. the latent-prefix encoder), 6 Euler steps per block (4 CFG steps x3 rows
    of 160 = M 480, 2 plain ones), fp32 engine against the fp32 oracle (pinned bit for bit to the reference's blockwise sampler on the
    tiny fixtures): latents <= 1e-3 RMS per chunk; the tiny DAC decodes both chunks (waveform <= 1e-4) and the device cross-fade of the two
    waveforms equals the reference-pinned host cross-fade of the oracle's waveforms to the waveform tolerance."""
    from echo_tts_amd import handler as H
    from echo_tts_amd import inference as inf
    cfg = WIDE1
    w = R.make_dit_weights(cfg, seed=0)
    m = E.EchoDiT(cfg, w, dtype=torch.float32, device=DEV)
    import dataclasses
    dcfg = dataclasses.replace(TINY_DAC, post_block_size=4096)      # 640 frames per chunk: the post_module's rope table must reach them
    dw = R.make_dac_weights(dcfg, 0)
    dac = E.DAC(dcfg, dw, device=DEV)
    pca = R.make_pca(dcfg, 80, 0)
    st = E.PCAState(pca.pca_components, pca.pca_mean, pca.latent_scale)
    text = ("The quick brown fox jumps over the lazy dog near the quiet river bank, while the evening sun paints the sky in orange. "
            "A second sentence follows so that the chunker has to cut the request into more than one piece for the sampler.")
    pieces = H.chunk_text_for_audio(text, max_chars=120, target_duration_seconds=10.0)
    assert len(pieces) >= 2
    pieces = pieces[:2]
    gen = torch.Generator().manual_seed(12)
    spk, smask = torch.randn((1, 256, 80), generator=gen), torch.ones((1, 256), dtype=torch.bool)
    kw = dict(SAMPLER_CASES["cfg_default"])
    torch.set_num_threads(16)
    wav_got, wav_want = [], []
    for ci, piece in enumerate(pieces):
        ids, tmask = inf.get_text_input_ids_and_mask([piece], max_length=768)
        xi = [torch.randn((1, 160, 80), generator=torch.Generator().manual_seed(100 * ci + j)) for j in range(4)]
        got = E.sample_blockwise_euler_cfg_independent_guidances(m, spk, smask, ids, tmask, rng_seed=ci, block_sizes=[160] * 4, x_inits=xi, **kw)
        want = R.sample_blockwise(w, cfg, torch.float32, spk, smask, ids, tmask, rng_seed=ci, block_sizes=[160] * 4, x_inits=xi, **kw)
        assert got.shape == (1, 640, 80)
        e = rms(got, want)
        print(f"C4 chunk {ci}: blockwise [160] x 4 latents rms {e:.3e} vs the fp32 oracle (latent rms {U.rms(want):.3f})")
        assert e < LAT_TOL, e
        a = E.ae_decode(dac, st, got)
        b = R.ae_decode(dw, dcfg, pca, want)
        assert rms(a, b) < WAV_TOL
        wav_got.append(a[0])
        wav_want.append(b[0])
    out = H.crossfade_chunks_device(wav_got)
    ref = H.crossfade_chunks(wav_want, 4410)
    assert out.shape == ref.shape and out.shape[-1] == 2 * 640 * 2048 - 4410
    assert rms(out, ref) < WAV_TOL


def test_load_audio_resampler_on_the_device(tmp_path):
    """SURVEY 8f-4 `load_audio` (reference inference.py:104-113).  PARITY UNPINNED by the reference: torchcodec / torchaudio are not in
    this image, so there is no reference output to compare with.  Checked instead: (1) `echo_op_resample` against a float64
    evaluation of torchaudio's published sinc_interp_hann formula (conv1d with stride on the zero-padded signal); (2) analytic: a
    1 kHz sine at 48 kHz comes out as a 1 kHz sine at 44.1 kHz (amplitude and phase, away from the edges); (3) `load_audio` on a
    stereo 48 kHz WAV: channel mean, 44.1 kHz, peak scaled to <= 1, on the GPU."""
    import math
    from echo_tts_amd import audio_io as A
    g = torch.Generator().manual_seed(1)
    for o, n, length in ((48000, 44100, 5000), (16000, 44100, 1777), (44100, 24000, 4410)):
        x = torch.randn((2, length), generator=g)
        got = A.resample(x, o, n, device=DEV).cpu()
        bank, up, down, width = A.sinc_resample_bank(o, n)
        xp = torch.nn.functional.pad(x.double(), (width, width + down))
        ref = torch.nn.functional.conv1d(xp[:, None], bank.double()[:, None], stride=down).transpose(1, 2).reshape(2, -1)
        target = math.ceil(up * length / down)
        assert got.shape == (2, target)
        assert float((got.double() - ref[:, :target]).abs().max()) < 2e-5 * float(ref.abs().max())
    sr, f0 = 48000, 1000.0
    tt = torch.arange(48000, dtype=torch.float64) / sr
    y = A.resample(torch.sin(2 * math.pi * f0 * tt).float()[None], sr, 44100, device=DEV).cpu()[0].double()
    t2 = torch.arange(y.shape[0], dtype=torch.float64) / 44100
    mid = slice(500, y.shape[0] - 500)
    assert float((y[mid] - torch.sin(2 * math.pi * f0 * t2)[mid]).abs().max()) < 2e-3
    wav = torch.stack([0.5 * torch.sin(2 * math.pi * 440 * tt), 2.5 * torch.sin(2 * math.pi * 440 * tt)]).float()     # mean peaks at 1.5
    p = tmp_path / "v.wav"
    import struct
    payload = wav.t().contiguous().numpy().tobytes()
    fmt = struct.pack("<HHIIHH", 3, 2, sr, sr * 8, 8, 32)
    p.write_bytes(b"RIFF" + struct.pack("<I", 4 + 8 + len(fmt) + 8 + len(payload)) + b"WAVE" + b"fmt " + struct.pack("<I", len(fmt)) + fmt +
                  b"data" + struct.pack("<I", len(payload)) + payload)
    a = E.load_audio(str(p), max_duration=300)
    assert a.is_cuda and a.shape == (1, 44100) and abs(float(a.abs().max()) - 1.0) < 2e-3
    a2 = E.load_audio(str(p), max_duration=0.5)
    assert a2.shape == (1, 22050)
