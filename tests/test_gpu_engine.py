"""GPU: the HIP engine, through the reference-shaped Python API, against the golden vectors the
reference produced (tests/golden) and against the CPU oracle on the same seeded inputs.

Tolerances (BASELINE.json north_star): latents <= 1e-3 RMS, waveform <= 1e-4 RMS, asserted for the
fp32 engine against the fp32 reference.  The bf16 engine cannot meet 1e-3 against a bf16 PyTorch
run any more than PyTorch meets it against itself, so it is held to: no farther
from the fp32 reference than 1.5x the reference's own bf16 run (+1e-3)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import echo_ref as R  # noqa: E402  (checker only)
from tests import gpu_util as U  # noqa: E402
from tests.golden_defs import SAMPLER_CASES, TINY, TINY_DAC, WIDE1  # noqa: E402

import echo_tts_amd as E  # noqa: E402
from echo_tts_amd.inference import _concat_kv_caches  # noqa: E402

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


@pytest.mark.parametrize("case,opts,cont", [("plain", "cfg_default", False), ("cont_opts", "all_options", True)])
def test_blockwise_sampler_f32(golden, tiny_models, case, opts, cont):
    g, m = golden, tiny_models["f32"]
    xi = [g[f"tiny.blk_x{j}"] for j in range(3)]
    lat = E.sample_blockwise_euler_cfg_independent_guidances(
        m, g["tiny.spk"], g["tiny.smask"].bool(), g["tiny.ids"], g["tiny.tmask"].bool(), rng_seed=0, block_sizes=[16, 8, 8],
        continuation_latent=g["tiny.blk_cont"] if cont else None, x_inits=xi, **SAMPLER_CASES[opts])
    e = rms(lat, g[f"tiny.f32.blockwise.{case}"])
    assert e < LAT_TOL, e


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


def test_dac_is_causal_and_length_independent(golden):
    """Size-independent property: decoding a longer input reproduces the shorter one's samples."""
    cfg = TINY_DAC
    dac = E.DAC(cfg, R.make_dac_weights(cfg, 0), device=DEV)
    z = torch.randn((1, cfg.latent_dim, 40), generator=torch.Generator().manual_seed(1))
    a = dac.decode_zq(z[..., :24])
    b = dac.decode_zq(z)
    assert rms(a, b[..., : a.shape[-1]]) < 1e-6
