"""Calibration only: what the vendor library (hipBLASLt through torch.matmul) reaches on the EchoDiT GEMM shapes.
Not part of the product path."""
import torch, sys
dev = "cuda:0"
shapes = [(1920, 8192, 2048), (1920, 2048, 2048), (1920, 11776, 2048), (1920, 2048, 5888),
          (640, 8192, 2048), (640, 2048, 2048), (640, 11776, 2048), (640, 2048, 5888),
          (7680, 8192, 2048), (7680, 2048, 2048), (7680, 11776, 2048), (7680, 2048, 5888),
          (4096, 4096, 4096), (8192, 8192, 8192)]
flush = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
for M, N, K in shapes:
    a = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    w = torch.randn(N, K, device=dev, dtype=torch.bfloat16)
    for _ in range(3):
        c = a @ w.t()
    res = []
    for cold in (0, 1):
        ts = []
        for r in range(5):
            if cold:
                flush.fill_(r)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); c = a @ w.t(); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        t = sorted(ts)[len(ts) // 2]
        res.append((t * 1e3, 2.0 * M * N * K / t / 1e9))
    print(f"torch bf16 M={M:5d} N={N:6d} K={K:5d} warm {res[0][0]:8.1f} us {res[0][1]:7.1f} TF | cold {res[1][0]:8.1f} us {res[1][1]:7.1f} TF", flush=True)
