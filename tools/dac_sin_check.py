"""Full-size Fish S1-DAC decode with sinf() and with sin_fast() in the conv tails' Snake: difference of the two waveforms."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import echo_tts_amd as E
from echo_tts_amd.weights import random_dac_state

dev = torch.device("cuda:0")
dcfg = E.DACConfig()
state = random_dac_state(dcfg, dev, seed=0)
g = torch.Generator().manual_seed(0)
q, _ = torch.linalg.qr(torch.randn(dcfg.latent_dim, 80, generator=g))
pca = E.PCAState(q.T.contiguous().to(dev), (0.1 * torch.randn(dcfg.latent_dim, generator=g)).to(dev), 1.0)
lat = torch.randn((1, 640, 80), generator=g).to(dev)
w = {}
for fast in ("0", "1"):
    os.environ["ECHO_DAC_FAST_SIN"] = fast
    dac = E.DAC(dcfg, state, device=dev)
    w[fast] = E.ae_decode(dac, pca, lat).double()
    torch.cuda.synchronize()
d = w["1"] - w["0"]
print(f"signal rms {w['0'].pow(2).mean().sqrt().item():.4e}; sin_fast vs sinf: rms {d.pow(2).mean().sqrt().item():.3e}, max abs {d.abs().max().item():.3e}")
