"""BASELINE config C4 on the GPU box: blockwise sampler, block_sizes [160] x 4, 40 steps, full-size EchoDiT with the latent
encoder, + Fish S1-DAC decode.  Reports audio-s/s for batch 1 and batch 4 (8 text chunks = 2 calls of 4)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import echo_tts_amd as E
from echo_tts_amd.weights import random_dac_state, random_dit_state

dev = torch.device("cuda:0")
cfg, dcfg = E.EchoDiTConfig(), E.DACConfig()
model = E.EchoDiT(cfg, random_dit_state(cfg, dev, torch.bfloat16, seed=0, with_blockwise=True), dtype=torch.bfloat16, device=dev)
dac = E.DAC(dcfg, random_dac_state(dcfg, dev, seed=0), device=dev)
g = torch.Generator().manual_seed(1234)
q, _ = torch.linalg.qr(torch.randn(dcfg.latent_dim, cfg.latent_size, generator=g))
pca = E.PCAState(q.T.contiguous().to(dev), (0.1 * torch.randn(dcfg.latent_dim, generator=g)).to(dev), 1.0)
kw = dict(block_sizes=[160] * 4, num_steps=40, cfg_scale_text=3.0, cfg_scale_speaker=8.0, cfg_min_t=0.5, cfg_max_t=1.0,
          truncation_factor=None, rescale_k=None, rescale_sigma=None, speaker_kv_scale=None, speaker_kv_max_layers=None,
          speaker_kv_min_t=None)
for B in (1, 4):
    ids = torch.zeros((B, 768), dtype=torch.int32)
    ids[:, 1:436] = torch.randint(32, 127, (B, 435), generator=g, dtype=torch.int32)
    tmask = torch.zeros((B, 768), dtype=torch.bool); tmask[:, :436] = True
    spk = torch.randn((B, 2560, 80), generator=g).to(dev); smask = torch.ones((B, 2560), dtype=torch.bool)
    ts = []
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        lat = E.sample_blockwise_euler_cfg_independent_guidances(model, spk, smask, ids.to(dev), tmask, it, **kw)
        wav = E.ae_decode(dac, pca, lat)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    best = min(ts[1:])
    print(f"C4 blockwise [160]x4, 40 steps, batch {B}: {best*1e3:.1f} ms per call, {B * 640 * 2048 / 44100 / best:.1f} audio-s/s "
          f"(finite: {bool(torch.isfinite(wav).all())})", flush=True)
