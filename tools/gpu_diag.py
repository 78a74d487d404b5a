import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import echo_ref as R
from tests.golden_defs import TINY_DAC
from safetensors.torch import load_file
import echo_tts_amd as E
g = {}
for fn in os.listdir("tests/golden"):
    if fn.endswith(".safetensors"): g.update(load_file(os.path.join("tests/golden", fn)))
def rms(a, b=None):
    a = a.float().cpu()
    if b is not None: a = a - b.float().cpu()
    return float(a.pow(2).mean().sqrt())
for tag, cfg in (("dac_tiny", TINY_DAC), ("dac_full", R.DacConfig())):
    w = R.make_dac_weights(cfg, 0)
    dac = E.DAC(cfg, w, device="cuda:0")
    wav = dac.decode_zq(g[f"{tag}.z"])
    print(tag, "decode_zq err", rms(wav, g[f"{tag}.wav"]), "signal", rms(g[f"{tag}.wav"]), "maxabs", float((wav.cpu()-g[f"{tag}.wav"]).abs().max()))
    pca = R.make_pca(cfg, 80, 0)
    out = E.ae_decode(dac, E.PCAState(pca.pca_components, pca.pca_mean, pca.latent_scale), g[f"{tag}.latent"])
    print(tag, "ae_decode err", rms(out, g[f"{tag}.ae_decode"]), "signal", rms(g[f"{tag}.ae_decode"]))
    # oracle on this host, same inputs, for thread/BLAS sensitivity
    ow = R.dac_decode_zq(w, cfg, g[f"{tag}.z"])
    print(tag, "oracle-here vs golden", rms(ow, g[f"{tag}.wav"]), " engine vs oracle-here", rms(wav, ow))
    # sensitivity: perturb input by 1e-6 relative
    z2 = g[f"{tag}.z"] * (1 + 1e-6)
    ow2 = R.dac_decode_zq(w, cfg, z2)
    print(tag, "oracle sensitivity to 1e-6 input scaling", rms(ow2, ow))
