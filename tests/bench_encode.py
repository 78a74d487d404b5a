"""Speaker-reference encode on the GPU box: agreement with the reference fixtures and time per 30 s chunk."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import echo_ref as R        # checker / weight recipe only
import echo_tts_amd as E
from safetensors.torch import load_file
from tests.golden_defs import FULL_ENC_SAMPLES

cfg = R.DacConfig()
w = R.make_dac_weights(cfg, 0); w.update(R.make_dac_encoder_weights(cfg, 0))
dac = E.DAC(cfg, w, device="cuda:0")
g = load_file(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "dac_encode.safetensors"))
audio = R.make_test_audio(FULL_ENC_SAMPLES, seed=11)
codes, _ = dac.encode(audio)
same = codes.cpu() == g["enc_full.codes"]
print(f"full-size codes equal to the reference: {float(same.float().mean()):.4f} (frames fully equal {float(same.all(dim=1).float().mean()):.4f})")
zq = dac.encode_zq(audio).cpu()
print("z_q rms error on all frames:", float((zq - g["enc_full.zq"]).pow(2).mean().sqrt()), "signal rms", float(g["enc_full.zq"].pow(2).mean().sqrt()))
pca = R.make_pca(cfg, 80, 0)
st = E.PCAState(pca.pca_components, pca.pca_mean, pca.latent_scale)
chunk = R.make_test_audio(640 * 2048, seed=3).to("cuda:0")
for i in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    lat = E.ae_encode(dac, st, chunk)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"ae_encode of one 29.7 s chunk (1 310 720 samples -> {tuple(lat.shape)}): {dt*1e3:.1f} ms")
spk = R.make_test_audio(int(118.94 * 44100), seed=4)[0].to("cuda:0")
torch.cuda.synchronize(); t0 = time.perf_counter()
sl, sm = E.get_speaker_latent_and_mask(dac, st, spk)
torch.cuda.synchronize(); print(f"get_speaker_latent_and_mask of a 118.9 s reference -> {tuple(sl.shape)}: {(time.perf_counter()-t0)*1e3:.1f} ms")
