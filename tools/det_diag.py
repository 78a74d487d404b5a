"""Determinism probes at full size (debugging aid): same launch twice must give bit-identical results."""
import ctypes as C, math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import gpu_util as U
from echo_tts_amd import _lib as L
dev = "cuda"
torch.manual_seed(0)
def gemm_twice(M, N, K, cfg, ks, swiglu=0):
    A = (torch.randn((M + 256, K), device=dev) * 0.5).bfloat16()
    W = (torch.randn(((N + 255) // 256 * 256, K), device=dev) * 0.05).bfloat16()
    outs = []
    for _ in range(3):
        Cc = torch.zeros((M, N // 2 if swiglu else N), dtype=torch.bfloat16, device=dev)
        U.gemm(A, W, Cc, M=M, N=N, K=K, lda=K, ldw=K, ldc=Cc.shape[1], cfg=cfg, ksplit=ks, swiglu=swiglu, Npad=(N + 127) // 128 * 128)
        torch.cuda.synchronize()
        outs.append(Cc.clone())
    same = all(torch.equal(outs[0], o) for o in outs[1:])
    ref = (A[:M].float() @ W[:N].float().T)
    if swiglu:
        return same, float("nan")
    err = float((outs[0].float() - ref).abs().max())
    return same, err
for (M, N, K) in ((1920, 8192, 2048), (1920, 2048, 5888), (640, 2048, 2048)):
    for cfg in range(5):
        for ks in (1, 4):
            same, err = gemm_twice(M, N, K, cfg, ks)
            print(f"gemm M={M} N={N} K={K} cfg{cfg} k{ks}: deterministic={same} maxerr={err:.3f}", flush=True)
for cfg in range(5):
    same, _ = gemm_twice(1920, 11776, 2048, cfg, 1, swiglu=1)
    print(f"gemm swiglu cfg{cfg}: deterministic={same}", flush=True)
# attention twice
sys.argv = [sys.argv[0]]
import importlib.util
spec = importlib.util.spec_from_file_location("bench_attn", os.path.join(os.path.dirname(__file__), "bench_attn.py"))
def attn_twice(R, S=640, H=16, Lt=436, Ls=640):
    D = H * 128
    qkvg = (torch.randn((R * S + 256, 4 * D), device=dev) * 0.5).bfloat16()
    pS, pT, pSp = (S + 63) // 64 * 64, (Lt + 63) // 64 * 64, (Ls + 63) // 64 * 64
    vt_self = torch.randn((R, H, 128, pS), device=dev).bfloat16()
    kt = torch.randn((Lt + 128, 48 * D), device=dev).bfloat16(); vt_t = torch.randn((1, H, 128, pT), device=dev).bfloat16()
    ksp = torch.randn((Ls + 128, 48 * D), device=dev).bfloat16(); vt_s = torch.randn((1, H, 128, pSp), device=dev).bfloat16()
    rows = [[S] * R, [Lt, 0, Lt][:R], [Ls, Ls, 0][:R]]
    nk = torch.tensor(rows, dtype=torch.int32, device=dev)
    outs = []
    for _ in range(3):
        out = torch.zeros((R * S, D), dtype=torch.bfloat16, device=dev)
        d = L.EchoAttnDesc()
        d.Q, d.q_ld, d.q_row_stride = qkvg.data_ptr(), 4 * D, S * 4 * D
        d.O, d.o_ld, d.o_row_stride = out.data_ptr(), D, S * D
        d.G, d.g_ld, d.g_row_stride = qkvg.data_ptr() + 3 * D * 2, 4 * D, S * 4 * D
        d.S, d.H, d.rows, d.nseg, d.causal, d.scale = S, H, R, 3, 0, 1 / math.sqrt(128)
        segs = [(qkvg.data_ptr() + D * 2, 4 * D, S * 4 * D, vt_self, pS, False), (kt.data_ptr(), 48 * D, 0, vt_t, pT, True),
                (ksp.data_ptr(), 48 * D, 0, vt_s, pSp, True)]
        for i, (kp, kld, krs, vt, pitch, shared) in enumerate(segs):
            sg = d.seg[i]
            sg.K, sg.k_ld, sg.k_head_stride, sg.k_row_stride = kp, kld, 128, krs
            sg.Vt, sg.vt_ld, sg.vt_head_stride = vt.data_ptr(), pitch, 128 * pitch
            sg.vt_row_stride = 0 if shared else H * 128 * pitch
            sg.nkeys = nk[i].data_ptr()
            sg.kv_mod = 1 if shared else 0
        L.check(U.lib().echo_op_attention_bf16(C.byref(d), U.stream()))
        torch.cuda.synchronize()
        outs.append(out.clone())
    print(f"attention R={R}: deterministic={all(torch.equal(outs[0], o) for o in outs[1:])} finite={bool(torch.isfinite(outs[0].float()).all())}", flush=True)
attn_twice(3); attn_twice(1)
