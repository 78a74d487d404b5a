"""In-kernel cycle accounting of the ping-pong GEMM (cfg 105 = correct kernel + s_memtime sums per wave)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import gpu_util as U

for (M, N, K) in ((7680, 8192, 2048), (7680, 2048, 5888), (2560, 8192, 2048)):
    A = (torch.rand((M + 256, K), device="cuda") * 2 - 1).to(torch.bfloat16)
    W = (torch.rand(((N + 255) // 256 * 256, K), device="cuda") * 2 - 1).to(torch.bfloat16)
    C = torch.zeros((M, N), dtype=torch.bfloat16, device="cuda")
    ws = torch.zeros((256 * 8 * 8,), dtype=torch.int64, device="cuda")
    kw = dict(M=M, N=N, K=K, lda=K, ldw=K, ldc=N, Npad=(N + 127) // 128 * 128)
    for _ in range(40):     # long enough for the clock to settle under load
        U.gemm(A, W, C, cfg=105, ws=ws, **kw)
    torch.cuda.synchronize()
    r = ws.view(256, 8, 8).double()
    used = r[:, :, 6] > 0
    names = ["total", "kloop", "epilogue"]
    for grp, sl in (("G0 (waves 0-3)", slice(0, 4)), ("G1 (waves 4-7)", slice(4, 8))):
        x = r[:, sl][used[:, sl]]
        tiles = x[:, 6].mean().item()
        line = f"M={M} N={N} K={K} {grp}: tiles/wg {tiles:.2f} |"
        for i, n in enumerate(names):
            line += f" {n} {x[:, i].mean().item() / tiles:9.0f}"
        clk = (x[:, 0] / (x[:, 3] / 100e6)).mean().item() / 1e9
        print(line + f"  (cycles per tile, s_memtime ticks); in-kernel clock {clk:.2f} GHz (s_memtime / s_memrealtime)", flush=True)

# per-phase breakdown (cfg 108)
M, N, K = 7680, 8192, 2048
A = (torch.rand((M + 256, K), device="cuda") * 2 - 1).to(torch.bfloat16)
W = (torch.rand(((N + 255) // 256 * 256, K), device="cuda") * 2 - 1).to(torch.bfloat16)
C = torch.zeros((M, N), dtype=torch.bfloat16, device="cuda")
ws = torch.zeros((256 * 8 * 16,), dtype=torch.int64, device="cuda")
for _ in range(2):
    U.gemm(A, W, C, cfg=108, ws=ws, M=M, N=N, K=K, lda=K, ldw=K, ldc=N, Npad=N)
torch.cuda.synchronize()
r = ws.view(256, 8, 4, 4).double()
nph = 32 * 3.75   # K-tiles per workgroup
for grp, sl in (("G0", slice(0, 4)), ("G1", slice(4, 8))):
    x = r[:, sl].mean(dim=(0, 1)) / nph
    for q in range(4):
        print(f"{grp} phase q{q}: load {x[q,0]:6.0f}  wait-bar1 {x[q,1]:6.0f}  mfma {x[q,2]:6.0f}  wait-bar2 {x[q,3]:6.0f}   (cycles per phase)")
