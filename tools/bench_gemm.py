"""Micro-benchmark of gemm_nt on the EchoDiT shapes: every tile config x split-K (run on the GPU box)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import gpu_util as U

def bench(M, N, K, dtype=torch.bfloat16, swiglu=0, iters=10, cfgs=(0, 1, 2, 3, 4, 5), splits=(1, 2, 4, 8), split3=0):
    A = (torch.randn((M + 256, K), device="cuda") * 0.5).to(dtype)
    W = (torch.randn(((N + 255) // 256 * 256, K), device="cuda") * 0.05).to(dtype)
    C = torch.zeros((M, N if not swiglu else N // 2), dtype=dtype, device="cuda")
    ldc = C.shape[1]
    res = []
    for cfg in cfgs:
        for ks in splits:
            nk = K // (64 if dtype == torch.bfloat16 else 32)
            if ks > 1 and nk // ks < 4:
                continue
            kw = dict(M=M, N=N, K=K, lda=K, ldw=K, ldc=ldc, swiglu=swiglu, cfg=cfg, ksplit=ks, Npad=(N + 127) // 128 * 128, split3=split3)
            for _ in range(2):
                U.gemm(A, W, C, **kw)
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(iters):
                U.gemm(A, W, C, **kw)
            e.record()
            torch.cuda.synchronize()
            ms = s.elapsed_time(e) / iters
            res.append((ms, cfg, ks))
    res.sort()
    tf = lambda ms: 2.0 * M * N * K / ms / 1e9
    best = res[0]
    line = " | ".join(f"c{c}k{k}:{ms*1e3:6.1f}us" for ms, c, k in res[:6])
    print(f"{str(dtype)[6:]:8s} M={M:5d} N={N:6d} K={K:5d} sw={swiglu} best c{best[1]}k{best[2]} {best[0]*1e3:7.1f} us {tf(best[0]):7.1f} TF || {line}", flush=True)

if __name__ == "__main__":
    for M in (7680, 3840, 2560, 1920, 1280, 640):
        bench(M, 8192, 2048)
        bench(M, 2048, 2048)
        bench(M, 11776, 2048, swiglu=1)
        bench(M, 2048, 5888)
    bench(4096, 4096, 4096, splits=(1,))
    bench(8192, 8192, 8192, iters=3, splits=(1,))
    bench(4096, 4096, 4096, dtype=torch.float32, iters=3, splits=(1,))
    bench(20480, 768, 768, dtype=torch.float32, iters=5, splits=(1, 2))
    bench(163840, 384, 384, dtype=torch.float32, iters=3, splits=(1,))
    print("--- fp32 with 3 x bf16 MFMA (split3)")
    bench(4096, 4096, 4096, dtype=torch.float32, iters=3, splits=(1,), split3=1)
    bench(20480, 768, 768, dtype=torch.float32, iters=5, splits=(1, 2), split3=1)
    bench(163840, 384, 384, dtype=torch.float32, iters=3, splits=(1,), split3=1)
