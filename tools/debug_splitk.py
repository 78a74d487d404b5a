"""Debug aid: GEMM with split-K and every tail - are rows [0, 8) bit-identical when the same rows are run as M = 8 and as M = 16?"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import gpu_util as U
DEV = U.DEV
torch.manual_seed(0)
for dt in (torch.float32, torch.bfloat16):
    for (N, K) in ((256, 256), (256, 320), (1024, 256), (256, 384), (768, 256)):
        A = torch.randn((16, K), device=DEV).to(dt)
        Wr = torch.randn((N, K), device=DEV).to(dt) * 0.1
        W = U.pad_rows(Wr)
        bias = torch.randn((N,), device=DEV).to(dt)
        res = torch.randn((16, N), device=DEV).to(dt)
        for variant in ("plain", "bias_div", "res_inplace", "swiglu", "ldc"):
            for cfg in (0, 1):
                for ks in (1, 2, 3):
                    outs = []
                    for M in (8, 16):
                        if variant == "swiglu":
                            Wp = U.pack_swiglu(Wr[: N // 2].contiguous(), Wr[N // 2:].contiguous())
                            C = torch.zeros((16, N // 2), dtype=dt, device=DEV)
                            U.gemm(A, Wp, C, M=M, N=N, K=K, lda=K, ldw=K, ldc=N // 2, swiglu=1, Npad=Wp.shape[0], cfg=cfg, ksplit=ks)
                        elif variant == "bias_div":
                            C = torch.zeros((16, N), dtype=dt, device=DEV)
                            U.gemm(A, W, C, M=M, N=N, K=K, lda=K, ldw=K, ldc=N, bias=bias, div=6.0, cfg=cfg, ksplit=ks)
                        elif variant == "res_inplace":
                            C = res.clone()
                            U.gemm(A, W, C, M=M, N=N, K=K, lda=K, ldw=K, ldc=N, res=C, ldres=N, cfg=cfg, ksplit=ks)
                        elif variant == "ldc":
                            C = torch.zeros((16, 2 * N), dtype=dt, device=DEV)
                            U.gemm(A, W, C, M=M, N=N, K=K, lda=K, ldw=K, ldc=2 * N, cfg=cfg, ksplit=ks)
                        else:
                            C = torch.zeros((16, N), dtype=dt, device=DEV)
                            U.gemm(A, W, C, M=M, N=N, K=K, lda=K, ldw=K, ldc=N, cfg=cfg, ksplit=ks)
                        torch.cuda.synchronize()
                        outs.append(C[:8].float().clone())
                    d = float((outs[0] - outs[1]).abs().max())
                    if d != 0.0:
                        print("DIFF", dt, (N, K), variant, "cfg", cfg, "ks", ks, "M8 vs M16 max|diff|", d)
print("done")
