"""Ping-pong GEMM kernel (cfg 5) experiments: interleaved rounds of several variants in ONE process (cdna guide rule 24).
cfg 101/102/103 are timing-only diagnostic builds (no LDS-DMA / no fragment reads / no MFMAs in the loop)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import gpu_util as U


def run(shapes, cfgs, rounds=5, iters=5, **extra):
    for (M, N, K) in shapes:
        A = (torch.rand((M + 256, K), device="cuda") * 2 - 1).to(torch.bfloat16)
        W = (torch.rand(((N + 255) // 256 * 256, K), device="cuda") * 2 - 1).to(torch.bfloat16)
        C = torch.zeros((M, N), dtype=torch.bfloat16, device="cuda")
        kw = dict(M=M, N=N, K=K, lda=K, ldw=K, ldc=N, Npad=(N + 127) // 128 * 128, **extra)
        t = {c: [] for c in cfgs}
        for c in cfgs:
            U.gemm(A, W, C, cfg=c, **kw)
        torch.cuda.synchronize()
        for r in range(rounds):
            for c in cfgs:
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                for _ in range(iters):
                    U.gemm(A, W, C, cfg=c, **kw)
                e.record()
                torch.cuda.synchronize()
                t[c].append(s.elapsed_time(e) / iters)
        line = f"M={M:5d} N={N:6d} K={K:5d} |"
        for c in cfgs:
            ms = sorted(t[c])[len(t[c]) // 2]
            line += f" c{c}: {ms * 1e3:7.1f} us {2.0 * M * N * K / ms / 1e9:7.1f} TF |"
        print(line, flush=True)


if __name__ == "__main__":
    cfgs = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [5, 101, 102, 103, 4]
    run([(7680, 8192, 2048), (7680, 2048, 5888), (2560, 8192, 2048), (2560, 2048, 2048), (4096, 4096, 4096), (8192, 8192, 8192)], cfgs)
