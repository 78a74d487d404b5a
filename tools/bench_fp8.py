"""bf16 vs fp8 (e4m3, block-scaled MFMA) ping-pong GEMM on the EchoDiT block shapes, interleaved rounds in one process,
plus the row-quantisation kernel on the activation shapes."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import gpu_util as U


def timeit(fn, iters=5):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def main():
    M0 = int(sys.argv[1]) if len(sys.argv) > 1 else 15360
    for (M, N, K, sw) in ((M0, 8192, 2048, 0), (M0, 2048, 2048, 0), (M0, 11776, 2048, 1), (M0, 2048, 5888, 0)):
        A = (torch.rand((M + 256, K), device="cuda") * 2 - 1).to(torch.bfloat16)
        W = (torch.rand(((N + 255) // 256 * 256, K), device="cuda") * 2 - 1).to(torch.bfloat16)
        C = torch.zeros((M, N), dtype=torch.bfloat16, device="cuda")
        A8, sa = U.quant_rows_fp8(A)
        W8, sw_ = U.quant_rows_fp8(W)
        kw = dict(M=M, N=N, K=K, lda=K, ldw=K, ldc=N // 2 if sw else N, Npad=W.shape[0], swiglu=sw, cfg=5)
        f16 = lambda: U.gemm(A, W, C, **kw)
        f8 = lambda: U.gemm(A8, W8, C, a_scale=sa, w_scale=sw_, **kw)
        fq = lambda: U.quant_rows_fp8(A[:M])
        for f in (f16, f8, fq):
            f()
        torch.cuda.synchronize()
        t = {"bf16": [], "fp8": [], "quant": []}
        for r in range(5):
            t["bf16"].append(timeit(f16)); t["fp8"].append(timeit(f8)); t["quant"].append(timeit(fq))
        med = {k: sorted(v)[len(v) // 2] for k, v in t.items()}
        fl = 2.0 * M * N * K
        print(f"M={M} N={N} K={K} swiglu={sw} | bf16 {med['bf16']*1e3:7.1f} us {fl/med['bf16']/1e9:7.1f} TF | fp8 {med['fp8']*1e3:7.1f} us "
              f"{fl/med['fp8']/1e9:7.1f} TF | quant A ({M}x{K}) {med['quant']*1e3:6.1f} us {M*K*3/med['quant']/1e6:6.0f} GB/s", flush=True)


if __name__ == "__main__":
    main()
