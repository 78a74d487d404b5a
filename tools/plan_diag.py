"""Run a full-width 2-layer EchoDiT sampler under forced GEMM plans and compare the latents (debugging aid)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    sys.path.insert(0, ROOT)
    import echo_tts_amd as E
    from echo_tts_amd.weights import random_dit_state
    dev = torch.device("cuda:0")
    cfg = E.EchoDiTConfig(num_layers=2, text_num_layers=1, speaker_num_layers=1)
    m = E.EchoDiT(cfg, random_dit_state(cfg, dev, torch.bfloat16, seed=0), dtype=torch.bfloat16, device=dev)
    g = torch.Generator().manual_seed(0)
    ids = torch.zeros((1, 768), dtype=torch.int32); ids[0, 1:436] = torch.randint(32, 127, (435,), generator=g, dtype=torch.int32)
    tm = torch.zeros((1, 768), dtype=torch.bool); tm[0, :436] = True
    spk = torch.randn((1, 2560, 80), generator=g); sm = torch.ones((1, 2560), dtype=torch.bool)
    x0 = torch.randn((1, 640, 80), generator=g)
    lat = E.sample_euler_cfg_independent_guidances(m, spk, sm, ids, tm, rng_seed=0, num_steps=4, cfg_scale_text=3.0, cfg_scale_speaker=8.0,
        cfg_min_t=0.5, cfg_max_t=1.0, truncation_factor=None, rescale_k=None, rescale_sigma=None, speaker_kv_scale=None,
        speaker_kv_max_layers=None, speaker_kv_min_t=None, sequence_length=640, x_init=x0)
    torch.save(lat.cpu(), sys.argv[2])
    sys.exit(0)
import torch
outs = {}
for cfg in range(5):
    for ks in (1, 2, 4, 8):
        tag = f"{cfg},{ks}"
        f = f"/tmp/lat_{cfg}_{ks}.pt"
        env = dict(os.environ, ECHO_GEMM_FORCE=tag)
        r = subprocess.run([sys.executable, __file__, "child", f], env=env, capture_output=True, text=True)
        if r.returncode != 0:
            print(tag, "FAILED", r.stderr[-300:]); continue
        outs[tag] = torch.load(f)
ref = outs["0,1"]
for tag, lat in outs.items():
    d = (lat - ref)
    print(f"plan {tag}: finite {bool(torch.isfinite(lat).all())} rms {float(lat.pow(2).mean().sqrt()):.4f} diff-rms vs 0,1 {float(d.pow(2).mean().sqrt()):.3e} max {float(d.abs().max()):.3e}", flush=True)
