"""Re-tune GEMM plans for the DAC decode (T = 640, single + batch 24) and the single-request / C3-share sampler shapes with the timing tuner.
Run on the GPU box:  ECHO_GEMM_TUNE=1 ECHO_GEMM_PLANS=/nonexistent ECHO_GEMM_PLANS_SAVE=gpurun_out/plans_new.txt ECHO_GEMM_VERBOSE=1 python tools/retune_plans.py
(the tuner times every tile configuration x split-K factor on the real operands with a flushed cache and appends the winner)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import echo_tts_amd as E
from echo_tts_amd.weights import random_dac_state, random_dit_state

dev = torch.device("cuda:0")
what = sys.argv[1] if len(sys.argv) > 1 else "dac"
if what in ("dac", "all"):
    dcfg = E.DACConfig()
    dac = E.DAC(dcfg, random_dac_state(dcfg, dev, seed=0), device=dev)
    g = torch.Generator().manual_seed(0)
    q, _ = torch.linalg.qr(torch.randn(dcfg.latent_dim, 80, generator=g))
    pca = E.PCAState(q.T.contiguous().to(dev), (0.1 * torch.randn(dcfg.latent_dim, generator=g)).to(dev), 1.0)
    for B in (1, 24):
        lat = torch.randn((B, 640, 80), device=dev)
        w = E.ae_decode(dac, pca, lat)
        torch.cuda.synchronize()
        print("decoded", B, tuple(w.shape), flush=True)
if what in ("dit", "all"):
    cfg = E.EchoDiTConfig()
    m = E.EchoDiT(cfg, random_dit_state(cfg, dev, torch.bfloat16, seed=0), dtype=torch.bfloat16, device=dev)
    kw = dict(num_steps=4, cfg_scale_text=3.0, cfg_scale_speaker=8.0, cfg_min_t=0.5, cfg_max_t=1.0, truncation_factor=None, rescale_k=None, rescale_sigma=None,
              speaker_kv_scale=None, speaker_kv_max_layers=None, speaker_kv_min_t=None, sequence_length=640)
    spk, smask = torch.randn((1, 2560, 80), device=dev), torch.ones((1, 2560), dtype=torch.bool)
    for B in (1, 4):
        ids = torch.zeros((B, 768), dtype=torch.int32); ids[:, 1:436] = 65
        tm = torch.zeros((B, 768), dtype=torch.bool); tm[:, :436] = True
        E.sample_euler_cfg_independent_guidances(m, spk, smask, ids.to(dev), tm, rng_seed=0, **kw)
        torch.cuda.synchronize()
        print("sampled", B, flush=True)
