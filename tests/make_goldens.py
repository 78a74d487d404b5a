#!/usr/bin/env python3
"""Generate tests/golden/* by running the REFERENCE itself (imported from /root/reference, CPU).

Run in the build container only (the reference never travels to the GPU box):

    python tests/make_goldens.py

What it does: builds the reference's own nn.Modules (model.EchoDiT, autoencoder.DAC) at small
sizes, loads seeded weights produced by oracle.echo_ref.make_*_weights (the recipe is repo code,
so the GPU box regenerates identical weights from the seed), runs the reference functions
(EchoDiT.forward / get_kv_cache_*, sample_euler_cfg_independent_guidances,
sample_blockwise_euler_cfg_independent_guidances, DAC.decode_zq, ae_decode, tokenizer & chunking
helpers) and stores inputs + outputs as small safetensors / JSON fixtures.  Fixtures are data
(inputs and expected outputs), never reference source.

torchaudio / torchcodec / runpod / boto3 are not installed here; they are only used by the
reference for audio file I/O, S3 and logging, so empty stub modules are registered before import
(SURVEY.md §8c).
"""
from __future__ import annotations

import hashlib
import json
import os
import sys
import tempfile
import types
from functools import partial

import torch
from safetensors.torch import save_file

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("ECHO_REFERENCE", "/root/reference")
GOLD = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)

from oracle import echo_ref as R  # noqa: E402


def _stub(name: str, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def import_reference():
    class _Dummy:
        def __init__(self, *a, **k):
            pass

        def __getattr__(self, n):
            return lambda *a, **k: None

    _stub("torchaudio", functional=types.SimpleNamespace(resample=None))
    _stub("torchcodec")
    _stub("torchcodec.decoders", AudioDecoder=_Dummy)
    rp = _stub("runpod", RunPodLogger=_Dummy)
    rp.serverless = types.SimpleNamespace(start=lambda *a, **k: None)
    _stub("boto3", client=lambda *a, **k: None)
    _stub("botocore")
    _stub("botocore.exceptions", ClientError=Exception, NoCredentialsError=Exception)
    tmp = tempfile.mkdtemp()
    os.environ.setdefault("AUDIO_VOICES_DIR", tmp)
    os.environ.setdefault("OUTPUT_AUDIO_DIR", tmp)
    sys.path.insert(0, REF)
    import model as ref_model  # noqa
    import autoencoder as ref_ae  # noqa
    import inference as ref_inf  # noqa
    import inference_blockwise as ref_blk  # noqa
    try:
        import handler as ref_handler  # noqa
    except Exception as e:  # handler has many service deps; only its pure helpers are used
        print("handler import failed (helpers skipped):", repr(e))
        ref_handler = None
    return ref_model, ref_ae, ref_inf, ref_blk, ref_handler


from tests.golden_defs import TINY, WIDE1, TINY_DAC, SAMPLER_CASES, tiny_inputs  # noqa: E402


def build_ref_dit(ref_model, cfg: R.DiTConfig, weights, dtype):
    m = ref_model.EchoDiT(
        latent_size=cfg.latent_size, model_size=cfg.model_size, num_layers=cfg.num_layers, num_heads=cfg.num_heads,
        intermediate_size=cfg.intermediate_size, norm_eps=cfg.norm_eps, text_vocab_size=cfg.text_vocab_size,
        text_model_size=cfg.text_model_size, text_num_layers=cfg.text_num_layers, text_num_heads=cfg.text_num_heads,
        text_intermediate_size=cfg.text_intermediate_size, speaker_patch_size=cfg.speaker_patch_size,
        speaker_model_size=cfg.speaker_model_size, speaker_num_layers=cfg.speaker_num_layers,
        speaker_num_heads=cfg.speaker_num_heads, speaker_intermediate_size=cfg.speaker_intermediate_size,
        timestep_embed_size=cfg.timestep_embed_size, adaln_rank=cfg.adaln_rank)
    missing, unexpected = m.load_state_dict({k: v.to(dtype) for k, v in weights.items()}, strict=True), None
    return m.eval().to(dtype)


def build_ref_dac(ref_ae, cfg: R.DacConfig, weights):
    if cfg.latent_dim == 1024:
        dac = ref_ae.build_ae()
    else:
        qcfg = ref_ae.ModelArgs(block_size=cfg.post_block_size, n_layer=cfg.post_layers, n_head=cfg.post_heads,
                                dim=cfg.latent_dim, intermediate_size=cfg.post_ffn, head_dim=cfg.post_head_dim,
                                norm_eps=cfg.norm_eps, dropout_rate=0.1, attn_dropout_rate=0.1, channels_first=True)

        def mk():
            return ref_ae.WindowLimitedTransformer(causal=True, window_size=cfg.post_window,
                                                   input_dim=cfg.latent_dim, config=qcfg)

        def tgc(**kw):   # the shape of build_ae's transformer_general_config (autoencoder.py:1163-1177)
            return ref_ae.ModelArgs(block_size=kw.get("block_size", cfg.encoder_block_size), n_layer=kw.get("n_layer", 8),
                                    n_head=kw.get("n_head", 8), dim=kw.get("dim", 512),
                                    intermediate_size=kw.get("intermediate_size", 1536), n_local_heads=kw.get("n_local_heads", -1),
                                    head_dim=kw.get("head_dim", 64), rope_base=kw.get("rope_base", 10000),
                                    norm_eps=kw.get("norm_eps", 1e-5), dropout_rate=0.1, attn_dropout_rate=0.1, channels_first=True)

        q = ref_ae.DownsampleResidualVectorQuantize(
            input_dim=cfg.latent_dim, n_codebooks=cfg.n_codebooks, codebook_size=cfg.codebook_size, codebook_dim=cfg.codebook_dim,
            quantizer_dropout=0.5, downsample_factor=cfg.upsample_factors, semantic_codebook_size=cfg.semantic_codebook_size,
            pre_module=mk(), post_module=mk())
        dac = ref_ae.DAC(encoder_dim=cfg.encoder_dim, encoder_rates=list(cfg.encoder_rates), latent_dim=cfg.latent_dim,
                         decoder_dim=cfg.decoder_dim, decoder_rates=list(cfg.decoder_rates), quantizer=q,
                         sample_rate=44100, causal=True, encoder_transformer_layers=list(cfg.encoder_transformer_layers),
                         decoder_transformer_layers=[0, 0, 0, 0], transformer_general_config=tgc)
    res = dac.load_state_dict(weights, strict=False)
    assert not res.unexpected_keys, res.unexpected_keys
    bad = [k for k in res.missing_keys if not (k.endswith("freqs_cis") or k.endswith("causal_mask"))]
    need_enc = any(k.startswith("encoder.") for k in weights)
    if not need_enc:
        bad = [k for k in bad if k.startswith("decoder.") or k.startswith("quantizer.post_module.layers") or k.startswith("quantizer.upsample")]
    assert not bad, bad[:10]
    return dac.eval()


def sha(t: torch.Tensor) -> str:
    return hashlib.sha256(t.detach().contiguous().cpu().numpy().tobytes()).hexdigest()[:16]


def weights_digest(w) -> str:
    h = hashlib.sha256()
    for k in sorted(w):
        h.update(k.encode())
        h.update(w[k].detach().contiguous().cpu().numpy().tobytes())
    return h.hexdigest()[:16]


def gen_dit(ref_model, ref_inf, ref_blk, tag: str, cfg: R.DiTConfig, out: dict, meta: dict, S: int, tt: int, tv: int,
            ts: int, run_samplers: bool, batch: int = 1):
    w = R.make_dit_weights(cfg, seed=0)
    meta[f"{tag}.weights_digest"] = weights_digest(w)
    ids, tmask, spk, smask, x0 = tiny_inputs(cfg, batch=batch, S=S, tt=tt, tv=tv, ts=ts)
    out[f"{tag}.ids"], out[f"{tag}.tmask"], out[f"{tag}.spk"], out[f"{tag}.smask"], out[f"{tag}.x0"] = \
        ids, tmask.to(torch.uint8), spk, smask.to(torch.uint8), x0
    for dname, dtype in (("f32", torch.float32), ("bf16", torch.bfloat16)):
        m = build_ref_dit(ref_model, cfg, w, dtype)
        with torch.inference_mode():
            kvt = m.get_kv_cache_text(ids, tmask)
            kvs = m.get_kv_cache_speaker(spk.to(dtype))
            out[f"{tag}.{dname}.kvt_k_last"] = kvt[-1][0].float().clone()
            out[f"{tag}.{dname}.kvt_v_last"] = kvt[-1][1].float().clone()
            out[f"{tag}.{dname}.kvs_k_last"] = kvs[-1][0].float().clone()
            out[f"{tag}.{dname}.kvs_v_last"] = kvs[-1][1].float().clone()
            t = torch.full((batch,), 0.7).to(dtype)
            v = m(x=x0.to(dtype), t=t, text_mask=tmask, speaker_mask=smask, kv_cache_text=kvt, kv_cache_speaker=kvs)
            out[f"{tag}.{dname}.forward_v"] = v.clone()
            # 3-row CFG-style forward (masks zeroed per row)
            kvt3 = ref_inf._concat_kv_caches(kvt, kvt, kvt)
            kvs3 = ref_inf._concat_kv_caches(kvs, kvs, kvs)
            tm3 = torch.cat([tmask, torch.zeros_like(tmask), tmask], 0)
            sm3 = torch.cat([smask, smask, torch.zeros_like(smask)], 0)
            v3 = m(x=torch.cat([x0, x0, x0], 0).to(dtype), t=torch.full((3 * batch,), 0.7).to(dtype), text_mask=tm3,
                   speaker_mask=sm3, kv_cache_text=kvt3, kv_cache_speaker=kvs3)
            out[f"{tag}.{dname}.forward_v3"] = v3.clone()
            if not run_samplers:
                continue
            # sampler cases: torch.randn is monkeypatched so the device RNG is not part of the fixture
            real_randn = torch.randn
            for cname, kw in SAMPLER_CASES.items():
                torch.randn = lambda *a, **k: x0.clone()
                try:
                    lat = ref_inf.sample_euler_cfg_independent_guidances(
                        m, spk, smask, ids, tmask, rng_seed=0, sequence_length=S, **kw)
                finally:
                    torch.randn = real_randn
                out[f"{tag}.{dname}.euler.{cname}"] = lat.clone()
            if "latent_encoder.in_proj.weight" in w:
                bs = [16, 8, 8]
                g = torch.Generator().manual_seed(99)
                xi = [real_randn((batch, n, cfg.latent_size), generator=g) for n in bs]
                cont = real_randn((batch, 8, cfg.latent_size), generator=g)
                for j, n in enumerate(bs):
                    out[f"{tag}.blk_x{j}"] = xi[j]
                out[f"{tag}.blk_cont"] = cont
                for cname, kw, c in (("plain", SAMPLER_CASES["cfg_default"], None),
                                     ("cont_opts", SAMPLER_CASES["all_options"], cont)):
                    it = iter(xi)
                    torch.randn = lambda *a, **k: next(it).clone()
                    try:
                        lat = ref_blk.sample_blockwise_euler_cfg_independent_guidances(
                            m, spk, smask, ids, tmask, rng_seed=0, block_sizes=bs, continuation_latent=c, **kw)
                    finally:
                        torch.randn = real_randn
                    out[f"{tag}.{dname}.blockwise.{cname}"] = lat.clone()


def gen_dac(ref_ae, ref_inf, tag: str, cfg: R.DacConfig, out: dict, meta: dict, T: int):
    w = R.make_dac_weights(cfg, seed=0)
    meta[f"{tag}.weights_digest"] = weights_digest(w)
    dac = build_ref_dac(ref_ae, cfg, w)
    g = torch.Generator().manual_seed(3)
    z = torch.randn((1, cfg.latent_dim, T), generator=g)
    out[f"{tag}.z"] = z
    with torch.inference_mode():
        zp = dac.quantizer.post_module(z)
        out[f"{tag}.post_module"] = zp.clone()
        zu = dac.quantizer.upsample(zp)
        out[f"{tag}.upsample"] = zu.clone()
        wav = dac.decode_zq(z)
        out[f"{tag}.wav"] = wav.clone()
        pca = R.make_pca(cfg, 80, seed=0)
        lat = torch.randn((1, T, 80), generator=g)
        out[f"{tag}.latent"] = lat
        st = ref_inf.PCAState(pca_components=pca.pca_components, pca_mean=pca.pca_mean, latent_scale=pca.latent_scale)
        out[f"{tag}.ae_decode"] = ref_inf.ae_decode(dac, st, lat).clone()
    meta[f"{tag}.wav_sha"] = sha(out[f"{tag}.wav"])


def gen_dac_encode(ref_ae, ref_inf, tag: str, cfg: R.DacConfig, out: dict, meta: dict, n_samples: int, chunked: bool):
    """Speaker-reference encode path: DAC.encode / encode_zq, ae_encode, get_speaker_latent_and_mask on seeded audio."""
    w = R.make_dac_weights(cfg, seed=0)
    w.update(R.make_dac_encoder_weights(cfg, seed=0))
    meta[f"{tag}.enc_weights_digest"] = weights_digest({k: v for k, v in w.items() if k.startswith("encoder.") or "pre_module" in k
                                                        or "downsample" in k or "quantizers" in k})
    dac = build_ref_dac(ref_ae, cfg, w)
    audio = R.make_test_audio(n_samples, seed=11)
    meta[f"{tag}.audio_sha"] = sha(audio)
    with torch.inference_mode():
        z = dac.encoder(torch.nn.functional.pad(audio, (0, (-n_samples) % 2048)))
        out[f"{tag}.encoder_head"] = z[..., :8].clone()
        meta[f"{tag}.encoder_sha"] = sha(z)
        codes, _ = dac.encode(audio)
        out[f"{tag}.codes"] = codes.clone()
        zq = dac.encode_zq(audio)
        out[f"{tag}.zq"] = zq.clone()
        pca = R.make_pca(cfg, 80, seed=0)
        st = ref_inf.PCAState(pca_components=pca.pca_components, pca_mean=pca.pca_mean, latent_scale=pca.latent_scale)
        out[f"{tag}.ae_encode"] = ref_inf.ae_encode(dac, st, audio).clone()
        if chunked:
            lat, mask = ref_inf.get_speaker_latent_and_mask(dac, st, audio[0], max_speaker_latent_length=24, audio_chunk_size=4 * 2048)
            out[f"{tag}.spk_latent"] = lat.clone()
            out[f"{tag}.spk_mask"] = mask.to(torch.uint8).clone()


def gen_host(ref_inf, ref_handler, meta: dict):
    texts = [
        "Hello world.",
        "[S1] Hello world.",
        "(laughs) It’s “quoted” — really; yes: no…\nnew line",
        "S2 appears here so no prefix",
        "Ünïcödé ✓ text",
    ]
    kat = {"tokenizer": [], "ids_mask": [], "chunk_text": [], "chunk_text_for_audio": []}
    for t in texts:
        ids, norm = ref_inf.tokenizer_encode(t, return_normalized_text=True)
        kat["tokenizer"].append({"text": t, "ids": ids.tolist(), "normalized": norm})
    ids, mask, norm = ref_inf.get_text_input_ids_and_mask(texts[:2], max_length=768, return_normalized_text=True,
                                                          pad_to_max=False)
    kat["ids_mask"].append({"texts": texts[:2], "max_length": 768, "pad_to_max": False, "shape": list(ids.shape),
                            "valid": mask.sum(1).tolist(), "first": ids[:, :24].tolist(), "normalized": norm})
    ids, mask = ref_inf.get_text_input_ids_and_mask(texts[:2], max_length=None)
    kat["ids_mask"].append({"texts": texts[:2], "max_length": None, "pad_to_max": True, "shape": list(ids.shape),
                            "valid": mask.sum(1).tolist(), "first": ids[:, :24].tolist()})
    presets = [l.strip() for l in open(os.path.join(REF, "text_presets.txt"), encoding="utf-8") if l.strip()]
    long_text = " ".join(presets[:3])
    for mc in (60, 120, 300):
        kat["chunk_text"].append({"text": long_text, "max_chars": mc, "chunks": ref_inf.chunk_text(long_text, mc)})
    kat["chunk_text"].append({"text": "  ", "max_chars": 10, "chunks": ref_inf.chunk_text("  ", 10)})
    kat["chunk_text"].append({"text": "abcdefghijklmnopqrstuvwxyz", "max_chars": 10,
                              "chunks": ref_inf.chunk_text("abcdefghijklmnopqrstuvwxyz", 10)})
    if ref_handler is not None:
        for dur in (5.0, 10.0):
            kat["chunk_text_for_audio"].append({"text": long_text, "max_chars": 300, "dur": dur,
                                                "chunks": ref_handler.chunk_text_for_audio(long_text, 300, dur)})
    # preset token lengths (benchmark inputs): lengths only, not the text
    kat["preset_token_lengths"] = [int(ref_inf.tokenizer_encode(p).shape[0]) for p in presets]
    # flattening point
    g = torch.Generator().manual_seed(11)
    lat = torch.randn((64, 80), generator=g)
    lat2 = lat.clone()
    lat2[40:] = 0.0
    lat3 = lat.clone()
    lat3[25:] = 0.01 * torch.randn((39, 80), generator=g)
    kat["flattening"] = [int(ref_inf.find_flattening_point(x)) for x in (lat, lat2, lat3)]
    kat["rescale_scalar"] = float(ref_inf._temporal_score_rescale(torch.tensor(1.0), torch.tensor(2.0),
                                                                   torch.tensor(0.5), 1.2, 3.0))
    meta["host"] = kat
    post = {}
    if ref_handler is not None:
        a = torch.randn((1, 30000), generator=g) * 0.1
        b = torch.randn((1, 20000), generator=g) * 0.1
        c = torch.randn((1, 9000), generator=g) * 0.1
        a[:, -3000:] = 0.0
        post["post.a"], post["post.b"], post["post.c"] = a, b, c
        post["post.crossfade"] = ref_handler.crossfade_chunks([a, b, c], 4410)
        post["post.normalize"] = ref_handler.normalize_chunk_boundaries([a, b, c], min_silence_samples=2000)
    return post, (lat, lat2, lat3)


def main_encode_only():
    """`--encode`: only the speaker-reference encode fixtures (tests/golden/dac_encode.safetensors), merged into meta.json."""
    from tests.golden_defs import TINY_ENC_SAMPLES, FULL_ENC_SAMPLES
    torch.manual_seed(0)
    torch.set_num_threads(8)
    ref_model, ref_ae, ref_inf, ref_blk, ref_handler = import_reference()
    meta = json.load(open(os.path.join(GOLD, "meta.json"), encoding="utf-8"))
    out = {}
    gen_dac_encode(ref_ae, ref_inf, "enc_tiny", TINY_DAC, out, meta, TINY_ENC_SAMPLES, chunked=True)
    gen_dac_encode(ref_ae, ref_inf, "enc_full", R.DacConfig(), out, meta, FULL_ENC_SAMPLES, chunked=False)
    save_file({k: v.contiguous() for k, v in out.items()}, os.path.join(GOLD, "dac_encode.safetensors"))
    with open(os.path.join(GOLD, "meta.json"), "w", encoding="utf-8") as f:
        json.dump(meta, f, indent=1, ensure_ascii=False)
    print({k: tuple(v.shape) for k, v in out.items()})


def main():
    if "--encode" in sys.argv:
        return main_encode_only()
    torch.manual_seed(0)
    torch.set_num_threads(8)
    ref_model, ref_ae, ref_inf, ref_blk, ref_handler = import_reference()
    os.makedirs(GOLD, exist_ok=True)
    meta = {"torch": torch.__version__}

    out = {}
    gen_dit(ref_model, ref_inf, ref_blk, "tiny", TINY, out, meta, S=32, tt=40, tv=24, ts=32, run_samplers=True)
    gen_dit(ref_model, ref_inf, ref_blk, "tinyb2", TINY, out, meta, S=32, tt=40, tv=24, ts=32, run_samplers=True, batch=2)
    save_file({k: v.contiguous() for k, v in out.items()}, os.path.join(GOLD, "dit_tiny.safetensors"))

    out = {}
    gen_dit(ref_model, ref_inf, ref_blk, "wide1", WIDE1, out, meta, S=64, tt=48, tv=30, ts=64, run_samplers=False)
    save_file({k: v.contiguous() for k, v in out.items()}, os.path.join(GOLD, "dit_wide1.safetensors"))

    out = {}
    gen_dac(ref_ae, ref_inf, "dac_tiny", TINY_DAC, out, meta, T=16)
    save_file({k: v.contiguous() for k, v in out.items()}, os.path.join(GOLD, "dac_tiny.safetensors"))

    from tests.golden_defs import TINY_ENC_SAMPLES, FULL_ENC_SAMPLES
    out = {}
    gen_dac_encode(ref_ae, ref_inf, "enc_tiny", TINY_DAC, out, meta, TINY_ENC_SAMPLES, chunked=True)
    gen_dac_encode(ref_ae, ref_inf, "enc_full", R.DacConfig(), out, meta, FULL_ENC_SAMPLES, chunked=False)
    save_file({k: v.contiguous() for k, v in out.items()}, os.path.join(GOLD, "dac_encode.safetensors"))

    out = {}
    gen_dac(ref_ae, ref_inf, "dac_full", R.DacConfig(), out, meta, T=8)
    # keep the full-size fixture small: drop the wide intermediates, keep input + waveform + ae_decode
    out = {k: v for k, v in out.items() if k.split(".")[-1] in ("z", "wav", "latent", "ae_decode")}
    save_file({k: v.contiguous() for k, v in out.items()}, os.path.join(GOLD, "dac_full.safetensors"))

    post, flats = gen_host(ref_inf, ref_handler, meta)
    post["flat.0"], post["flat.1"], post["flat.2"] = flats
    save_file({k: v.contiguous() for k, v in post.items()}, os.path.join(GOLD, "host.safetensors"))

    with open(os.path.join(GOLD, "meta.json"), "w", encoding="utf-8") as f:
        json.dump(meta, f, indent=1, ensure_ascii=False)
    for fn in sorted(os.listdir(GOLD)):
        print(fn, os.path.getsize(os.path.join(GOLD, fn)))


if __name__ == "__main__":
    main()
