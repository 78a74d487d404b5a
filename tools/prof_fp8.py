"""In-kernel cycle accounting of the fp8 (e4m3) ping-pong GEMM: cfg 105 = correct kernel + s_memtime sums per wave (total / K loops / epilogue per
tile), cfg 104 = the same kernel without its epilogue (wall-time share of the epilogue), cfg 108 = per-phase sums.  Random operands."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import gpu_util as U


def timeit(fn, iters=10):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3):
        fn()
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


M = int(sys.argv[1]) if len(sys.argv) > 1 else 15360
for (N, K, res) in ((8192, 2048, False), (2048, 2048, True), (2048, 5888, True)):
    A = (torch.rand((M + 256, K), device="cuda") * 2 - 1).to(torch.bfloat16)
    W = (torch.rand(((N + 255) // 256 * 256, K), device="cuda") * 2 - 1).to(torch.bfloat16)
    C = torch.zeros((M, N), dtype=torch.bfloat16, device="cuda")
    R = torch.randn((M, N), device="cuda").to(torch.bfloat16) if res else None
    cs = torch.rand((N,), device="cuda").to(torch.bfloat16) if res else None
    A8, sa = U.quant_rows_fp8(A)
    W8, sw = U.quant_rows_fp8(W)
    ws = torch.zeros((256 * 8 * 16,), dtype=torch.int64, device="cuda")
    kw = dict(M=M, N=N, K=K, lda=K, ldw=K, ldc=N, Npad=W.shape[0], a_scale=sa, w_scale=sw)
    if res:
        kw.update(res=R, ldres=N, colscale=cs)
    fl = 2.0 * M * N * K
    t5 = timeit(lambda: U.gemm(A8, W8, C, cfg=5, **kw))
    t4 = timeit(lambda: U.gemm(A8, W8, C, cfg=104, **kw))
    kwb = dict(M=M, N=N, K=K, lda=K, ldw=K, ldc=N, Npad=W.shape[0])
    if res:
        kwb.update(res=R, ldres=N, colscale=cs)
    tb = timeit(lambda: U.gemm(A, W, C, cfg=5, **kwb))
    tb4 = timeit(lambda: U.gemm(A, W, C, cfg=104, M=M, N=N, K=K, lda=K, ldw=K, ldc=N, Npad=W.shape[0]))
    print(f"M={M} N={N} K={K} res={int(res)}: fp8 {t5:7.1f} us ({fl/t5/1e6:6.0f} TF), without epilogue {t4:7.1f} us ({fl/t4/1e6:6.0f} TF) | "
          f"bf16 {tb:7.1f} us ({fl/tb/1e6:6.0f} TF), without epilogue {tb4:7.1f} us", flush=True)
    ws.zero_()
    for _ in range(20):
        U.gemm(A8, W8, C, cfg=105, ws=ws, **kw)
    torch.cuda.synchronize()
    r = ws.view(-1)[: 256 * 8 * 8].view(256, 8, 8).double()
    used = r[:, :, 6] > 0
    for grp, sl in (("G0", slice(0, 4)), ("G1", slice(4, 8))):
        x = r[:, sl][used[:, sl]]
        tiles = x[:, 6].mean().item()
        clk = (x[:, 0] / (x[:, 3] / 100e6)).mean().item() / 1e9
        print(f"    {grp}: tiles/wg {tiles:.2f} | per tile: total {x[:, 0].mean().item() / tiles:8.0f}  K loop {x[:, 1].mean().item() / tiles:8.0f}  "
              f"epilogue {x[:, 2].mean().item() / tiles:8.0f} cycles; in-kernel clock {clk:.2f} GHz", flush=True)
    ws.zero_()
    for _ in range(2):
        U.gemm(A8, W8, C, cfg=108, ws=ws, **kw)
    torch.cuda.synchronize()
    r = ws.view(256, 8, 4, 4).double() / 2
    nph = (K // 128) * ((M + 255) // 256) * ((N + 255) // 256) / 256.0
    for grp, sl in (("G0", slice(0, 4)), ("G1", slice(4, 8))):
        x = r[:, sl].mean(dim=(0, 1)) / nph
        print("    " + grp + " per phase: " + " | ".join(f"q{q}: load {x[q,0]:5.0f} bar1 {x[q,1]:5.0f} mfma {x[q,2]:5.0f} bar2 {x[q,3]:5.0f}" for q in range(4)), flush=True)
