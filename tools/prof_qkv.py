"""In-kernel cycle accounting (cfg 105: s_memtime sums per wave) of the ping-pong GEMM at the sampler's QKVG shape, with and without the
fused head-norm / RoPE / V-transpose tail, and wall time of the production kernel (cfg 5) for both."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import gpu_util as U

def run(M, S=640, D=2048, K=2048, rope_heads=8):
    N = 4 * D
    A = (torch.rand((M + 256, K), device="cuda") * 2 - 1).to(torch.bfloat16)
    W = ((torch.rand((N, K), device="cuda") * 2 - 1) * 0.05).to(torch.bfloat16)
    C = torch.zeros((M, N), dtype=torch.bfloat16, device="cuda")
    B = M // S
    vt = torch.zeros((B, D, S), dtype=torch.bfloat16, device="cuda")
    qk_w = torch.ones((2 * D,), dtype=torch.bfloat16, device="cuda")
    ang = torch.rand((S, 64), device="cuda")
    rope = torch.stack([torch.cos(ang), torch.sin(ang)], -1).contiguous()
    qkv = dict(D=D, S=S, rope_heads=rope_heads, pos0=0, eps=1e-6, qk_w=qk_w, rope=rope, vt=vt, vt_ld=S, vt_row_stride=D * S)
    kw = dict(M=M, N=N, K=K, lda=K, ldw=K, ldc=N, Npad=N)
    for name, extra in (("plain", {}), ("fused QKVG tail", {"qkv": qkv})):
        for _ in range(10):
            U.gemm(A, W, C, cfg=5, **kw, **extra)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20):
            U.gemm(A, W, C, cfg=5, **kw, **extra)
        e.record(); torch.cuda.synchronize()
        us = s.elapsed_time(e) / 20 * 1e3
        ws = torch.zeros((256 * 8 * 8,), dtype=torch.int64, device="cuda")
        for _ in range(10):
            U.gemm(A, W, C, cfg=105, ws=ws, **kw, **extra)
        torch.cuda.synchronize()
        r = ws.view(256, 8, 8).double()
        line = f"M={M} {name:16s}: {us:7.1f} us = {2.0 * M * N * K / us / 1e6:6.0f} TFLOP/s |"
        for grp, sl in (("G0", slice(0, 4)), ("G1", slice(4, 8))):
            x = r[:, sl][(r[:, sl, 6] > 0)]
            tiles = x[:, 6].mean().item()
            line += f" {grp}: tiles {tiles:.2f} total {x[:, 0].mean().item() / tiles:7.0f} kloop {x[:, 1].mean().item() / tiles:7.0f} epilogue {x[:, 2].mean().item() / tiles:7.0f} |"
        print(line, flush=True)

run(15360); run(5120)
