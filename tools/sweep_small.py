"""Small-M GEMM sweep (the single-request shapes): every tile configuration x split-K factor, weights rotated over 32 buffers so that
each launch streams its W from HBM like a layer of the sampler does."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import gpu_util as U

shapes = [(1920, 2048, 2048), (1920, 2048, 5888), (640, 2048, 2048), (640, 2048, 5888), (640, 8192, 2048), (1920, 8192, 2048)]
if len(sys.argv) > 1:
    shapes = [tuple(int(x) for x in a.split("x")) for a in sys.argv[1:]]
NB = 32
for (M, N, K) in shapes:
    A = (torch.rand((M + 256, K), device="cuda") * 2 - 1).to(torch.bfloat16)
    Ws = [(torch.rand(((N + 255) // 256 * 256, K), device="cuda") * 2 - 1).to(torch.bfloat16) for _ in range(NB)]
    C = torch.zeros((M, N), dtype=torch.bfloat16, device="cuda")
    ws = torch.zeros((8 * (M + 256) * (N + 256),), dtype=torch.float32, device="cuda")
    res = []
    for cfg in (0, 1, 2, 3, 4, 5):
        for ks in (1, 2, 3, 4, 6, 8):
            if (K // 64) % ks or (cfg == 5 and ks > 1 and K // 64 // ks < 4):
                continue
            kw = dict(M=M, N=N, K=K, lda=K, ldw=K, ldc=N, Npad=Ws[0].shape[0], cfg=cfg, ksplit=ks, ws=ws if ks > 1 else None)
            try:
                for i in range(4):
                    U.gemm(A, Ws[i], C, **kw)
                torch.cuda.synchronize()
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                for i in range(64):
                    U.gemm(A, Ws[i % NB], C, **kw)
                e.record(); torch.cuda.synchronize()
                res.append((s.elapsed_time(e) / 64 * 1e3, cfg, ks))
            except Exception as ex:
                pass
    res.sort()
    print(f"M={M} N={N} K={K}: " + " | ".join(f"cfg{c} ks{k} {t:.1f}us" for t, c, k in res[:8]), flush=True)
