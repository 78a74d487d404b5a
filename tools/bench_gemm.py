"""Micro-benchmark of gemm_nt on the EchoDiT shapes (run on the GPU box)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import gpu_util as U

def bench(M, N, K, dtype=torch.bfloat16, swiglu=0, iters=20):
    A = (torch.randn((M, K), device="cuda") * 0.5).to(dtype)
    W = (torch.randn(((N + 127) // 128 * 128, K), device="cuda") * 0.05).to(dtype)
    C = torch.zeros((M, N if not swiglu else N // 2), dtype=dtype, device="cuda")
    ldc = C.shape[1]
    for _ in range(3):
        U.gemm(A, W, C, M=M, N=N, K=K, lda=K, ldw=K, ldc=ldc, swiglu=swiglu)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        U.gemm(A, W, C, M=M, N=N, K=K, lda=K, ldw=K, ldc=ldc, swiglu=swiglu)
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e) / iters
    tf = 2.0 * M * N * K / ms / 1e9
    print(f"{str(dtype):16s} M={M:5d} N={N:6d} K={K:5d} swiglu={swiglu}: {ms*1e3:8.1f} us  {tf:7.1f} TFLOP/s", flush=True)
    return ms

if __name__ == "__main__":
    for M in (1920, 640):
        bench(M, 8192, 2048)
        bench(M, 2048, 2048)
        bench(M, 11776, 2048, swiglu=1)
        bench(M, 2048, 5888)
    bench(4096, 4096, 4096)
    bench(8192, 8192, 8192, iters=5)
    bench(4096, 4096, 4096, dtype=torch.float32, iters=5)
    bench(20480, 768, 768 , dtype=torch.float32, iters=5)
