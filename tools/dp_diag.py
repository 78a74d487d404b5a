import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import echo_tts_amd as E
from echo_tts_amd import parallel as P
from echo_tts_amd.weights import dit_param_shapes, random_dit_state
rank, world, local = P.init_distributed(backend=sys.argv[1] if len(sys.argv) > 1 else None)
dev = torch.device("cuda:0")
full = os.environ.get("DIAG_FULL") == "1"
cfg = E.EchoDiTConfig() if full else E.EchoDiTConfig(num_layers=2, text_num_layers=1, speaker_num_layers=1)
serial = os.environ.get("DIAG_SERIAL") == "1"
ref = random_dit_state(cfg, dev, torch.bfloat16, seed=0)
sd = ref if rank == 0 else None
spec = dit_param_shapes(cfg, with_blockwise=False)
out = P.broadcast_state(spec, sd, dev, torch.bfloat16)
torch.cuda.synchronize()
bad = [n for n, _ in spec if not torch.equal(out[n], ref[n])]
nonfin = [n for n, _ in spec if not bool(torch.isfinite(out[n].float()).all())]
print(f"rank {rank}: {len(spec)} tensors, mismatching {len(bad)} {bad[:3]}, nonfinite {len(nonfin)} {nonfin[:3]}", flush=True)
if serial and world > 1 and rank == 1:
    dist.barrier()
m = E.EchoDiT(cfg, out, dtype=torch.bfloat16, device=dev)
ids = torch.zeros((1, 64), dtype=torch.int32); ids[0, 1:40] = 65
tm = torch.zeros((1, 64), dtype=torch.bool); tm[0, :40] = True
spk = torch.randn((1, 64, 80)); sm = torch.ones((1, 64), dtype=torch.bool)
lat = E.sample_euler_cfg_independent_guidances(m, spk, sm, ids, tm, rng_seed=rank, num_steps=4, cfg_scale_text=3.0, cfg_scale_speaker=8.0,
    cfg_min_t=0.5, cfg_max_t=1.0, truncation_factor=None, rescale_k=None, rescale_sigma=None, speaker_kv_scale=None,
    speaker_kv_max_layers=None, speaker_kv_min_t=None, sequence_length=640 if full else 128)
print(f"rank {rank}: latent finite {bool(torch.isfinite(lat).all())} rms {float(lat.pow(2).mean().sqrt()):.4f}", flush=True)
kt, vt_ = m._read_kv(0, 0)
print(f"rank {rank}: text K finite {bool(torch.isfinite(kt).all())}", flush=True)
if serial and world > 1 and rank == 0:
    torch.cuda.synchronize(); dist.barrier()
if world > 1:
    dist.barrier(); dist.destroy_process_group()
