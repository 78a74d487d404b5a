"""CPU: the oracle (oracle/echo_ref.py) must reproduce, bit for bit, what the reference itself
produced on the same seeded weights and inputs (tests/golden/*, made by tests/make_goldens.py)."""
import hashlib

import pytest
import torch

from oracle import echo_ref as R
from tests.golden_defs import SAMPLER_CASES, TINY, TINY_DAC, WIDE1


def _digest(w):
    h = hashlib.sha256()
    for k in sorted(w):
        h.update(k.encode())
        h.update(w[k].detach().contiguous().cpu().numpy().tobytes())
    return h.hexdigest()[:16]


def _cast(w, dt):
    return {k: v.to(dt) for k, v in w.items()}


@pytest.fixture(scope="module")
def tiny_w():
    return R.make_dit_weights(TINY, seed=0)


def test_weight_recipe_is_reproducible(golden, tiny_w):
    assert _digest(tiny_w) == golden["__meta__"]["tiny.weights_digest"]
    assert _digest(R.make_dac_weights(TINY_DAC, 0)) == golden["__meta__"]["dac_tiny.weights_digest"]


@pytest.mark.parametrize("tag,batch", [("tiny", 1), ("tinyb2", 2)])
@pytest.mark.parametrize("dname,dt", [("f32", torch.float32), ("bf16", torch.bfloat16)])
def test_dit_forward_and_kv_bit_exact(golden, tiny_w, tag, batch, dname, dt):
    g = golden
    w = _cast(tiny_w, dt)
    ids, tm = g[f"{tag}.ids"], g[f"{tag}.tmask"].bool()
    spk, sm, x0 = g[f"{tag}.spk"], g[f"{tag}.smask"].bool(), g[f"{tag}.x0"]
    with torch.inference_mode():
        kvt = R.kv_cache_text(w, TINY, ids, tm)
        kvs = R.kv_cache_speaker(w, TINY, spk.to(dt))
        assert torch.equal(kvt[-1][0].float(), g[f"{tag}.{dname}.kvt_k_last"])
        assert torch.equal(kvt[-1][1].float(), g[f"{tag}.{dname}.kvt_v_last"])
        assert torch.equal(kvs[-1][0].float(), g[f"{tag}.{dname}.kvs_k_last"])
        assert torch.equal(kvs[-1][1].float(), g[f"{tag}.{dname}.kvs_v_last"])
        v = R.dit_forward(w, TINY, x0.to(dt), torch.full((batch,), 0.7).to(dt), tm, sm, kvt, kvs)
        assert torch.equal(v, g[f"{tag}.{dname}.forward_v"])
        kvt3, kvs3 = R._cat3(kvt), R._cat3(kvs)
        tm3 = torch.cat([tm, torch.zeros_like(tm), tm], 0)
        sm3 = torch.cat([sm, sm, torch.zeros_like(sm)], 0)
        v3 = R.dit_forward(w, TINY, torch.cat([x0, x0, x0], 0).to(dt), torch.full((3 * batch,), 0.7).to(dt),
                           tm3, sm3, kvt3, kvs3)
        assert torch.equal(v3, g[f"{tag}.{dname}.forward_v3"])


@pytest.mark.parametrize("tag", ["tiny", "tinyb2"])
@pytest.mark.parametrize("dname,dt", [("f32", torch.float32), ("bf16", torch.bfloat16)])
@pytest.mark.parametrize("case", list(SAMPLER_CASES))
def test_euler_sampler_bit_exact(golden, tiny_w, tag, dname, dt, case):
    g = golden
    w = _cast(tiny_w, dt)
    lat = R.sample_euler(w, TINY, dt, g[f"{tag}.spk"], g[f"{tag}.smask"].bool(), g[f"{tag}.ids"],
                         g[f"{tag}.tmask"].bool(), rng_seed=0, sequence_length=32, x_init=g[f"{tag}.x0"],
                         **SAMPLER_CASES[case])
    assert torch.equal(lat, g[f"{tag}.{dname}.euler.{case}"])


@pytest.mark.parametrize("dname,dt", [("f32", torch.float32), ("bf16", torch.bfloat16)])
@pytest.mark.parametrize("case,opts,cont", [("plain", "cfg_default", False), ("cont_opts", "all_options", True)])
def test_blockwise_sampler_bit_exact(golden, tiny_w, dname, dt, case, opts, cont):
    g = golden
    w = _cast(tiny_w, dt)
    xi = [g[f"tiny.blk_x{j}"] for j in range(3)]
    lat = R.sample_blockwise(w, TINY, dt, g["tiny.spk"], g["tiny.smask"].bool(), g["tiny.ids"],
                             g["tiny.tmask"].bool(), rng_seed=0, block_sizes=[16, 8, 8],
                             continuation_latent=g["tiny.blk_cont"] if cont else None, x_inits=xi,
                             **SAMPLER_CASES[opts])
    assert torch.equal(lat, g[f"tiny.{dname}.blockwise.{case}"])


@pytest.mark.parametrize("dname,dt", [("f32", torch.float32), ("bf16", torch.bfloat16)])
def test_full_width_single_layer_bit_exact(golden, dname, dt):
    g = golden
    w = R.make_dit_weights(WIDE1, seed=0)
    assert _digest(w) == g["__meta__"]["wide1.weights_digest"]
    w = _cast(w, dt)
    ids, tm = g["wide1.ids"], g["wide1.tmask"].bool()
    spk, sm, x0 = g["wide1.spk"], g["wide1.smask"].bool(), g["wide1.x0"]
    with torch.inference_mode():
        kvt = R.kv_cache_text(w, WIDE1, ids, tm)
        kvs = R.kv_cache_speaker(w, WIDE1, spk.to(dt))
        assert torch.equal(kvt[-1][0].float(), g[f"wide1.{dname}.kvt_k_last"])
        assert torch.equal(kvs[-1][1].float(), g[f"wide1.{dname}.kvs_v_last"])
        v = R.dit_forward(w, WIDE1, x0.to(dt), torch.full((1,), 0.7).to(dt), tm, sm, kvt, kvs)
        assert torch.equal(v, g[f"wide1.{dname}.forward_v"])


def test_dac_tiny_bit_exact(golden):
    g = golden
    w = R.make_dac_weights(TINY_DAC, 0)
    taps = {}
    wav = R.dac_decode_zq(w, TINY_DAC, g["dac_tiny.z"], taps)
    assert torch.equal(taps["post_module"], g["dac_tiny.post_module"])
    assert torch.equal(taps["upsample"], g["dac_tiny.upsample"])
    assert torch.equal(wav, g["dac_tiny.wav"])
    pca = R.make_pca(TINY_DAC, 80, 0)
    assert torch.equal(R.ae_decode(w, TINY_DAC, pca, g["dac_tiny.latent"]), g["dac_tiny.ae_decode"])


def test_dac_full_size_matches_reference(golden):
    g = golden
    cfg = R.DacConfig()
    w = R.make_dac_weights(cfg, 0)
    assert _digest(w) == g["__meta__"]["dac_full.weights_digest"]
    wav = R.dac_decode_zq(w, cfg, g["dac_full.z"])
    assert wav.shape == (1, 1, 8 * 2048)
    # fp32 conv reductions are thread-count dependent in ATen (8e-9 RMS), so allow 1e-6 abs
    assert (wav - g["dac_full.wav"]).abs().max().item() < 1e-6
    pca = R.make_pca(cfg, 80, 0)
    out = R.ae_decode(w, cfg, pca, g["dac_full.latent"])
    assert (out - g["dac_full.ae_decode"]).abs().max().item() < 1e-6


def test_flattening_point_kat(golden):
    g = golden
    want = g["__meta__"]["host"]["flattening"]
    got = [R.find_flattening_point(g[f"flat.{i}"]) for i in range(3)]
    assert got == want == [64, 40, 25]


def test_rescale_scalar_kat(golden):
    val = float(R.temporal_score_rescale(torch.tensor(1.0), torch.tensor(2.0), torch.tensor(0.5), 1.2, 3.0))
    assert val == golden["__meta__"]["host"]["rescale_scalar"]
    assert abs(val - 1.8824) < 1e-4


# ----------------------------------------------------------------------- speaker-reference encode path
def _enc_weights(cfg):
    w = R.make_dac_weights(cfg, 0)
    w.update(R.make_dac_encoder_weights(cfg, 0))
    return w


def test_dac_encode_tiny_bit_exact(golden):
    """Encoder -> downsample -> pre_module -> semantic VQ + residual VQs -> from_codes, PCA, and the chunked
    get_speaker_latent_and_mask: codes identical, every float output bit-identical to the reference's."""
    from tests.golden_defs import TINY_ENC_SAMPLES
    g, cfg = golden, TINY_DAC
    w = _enc_weights(cfg)
    audio = R.make_test_audio(TINY_ENC_SAMPLES, seed=11)
    taps = {}
    zq = R.dac_encode_zq(w, cfg, audio, taps)
    assert torch.equal(taps["codes"], g["enc_tiny.codes"])
    assert torch.equal(taps["encoder"][..., :8], g["enc_tiny.encoder_head"])
    assert torch.equal(zq, g["enc_tiny.zq"])
    pca = R.make_pca(cfg, 80, seed=0)
    assert torch.equal(R.ae_encode(w, cfg, pca, audio), g["enc_tiny.ae_encode"])
    lat, mask = R.get_speaker_latent_and_mask(w, cfg, pca, audio[0], max_speaker_latent_length=24, audio_chunk_size=4 * 2048)
    assert torch.equal(lat, g["enc_tiny.spk_latent"])
    assert torch.equal(mask.to(torch.uint8), g["enc_tiny.spk_mask"])


def test_dac_encode_full_size_matches_reference(golden):
    """Full-size Fish S1-DAC encode (4-layer window-512 encoder transformer over 600 positions, 8-layer pre_module, VQ 4096 +
    9 x 1024) on 307 200 samples: codes identical, z_q and latents bit-identical."""
    from tests.golden_defs import FULL_ENC_SAMPLES
    g, cfg = golden, R.DacConfig()
    w = _enc_weights(cfg)
    audio = R.make_test_audio(FULL_ENC_SAMPLES, seed=11)
    taps = {}
    zq = R.dac_encode_zq(w, cfg, audio, taps)
    assert torch.equal(taps["codes"], g["enc_full.codes"])
    assert torch.equal(zq, g["enc_full.zq"])
    assert torch.equal(R.ae_encode(w, cfg, R.make_pca(cfg, 80, seed=0), audio), g["enc_full.ae_encode"])
