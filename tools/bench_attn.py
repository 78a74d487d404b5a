"""Micro-benchmark of attn_kernel on the C2 joint-attention shape (run on the GPU box)."""
import ctypes as C, math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import gpu_util as U
from echo_tts_amd import _lib as L

def run(R, S=640, H=16, Lt=436, Ls=640, iters=20):
    dev = "cuda"
    D = H * 128
    qkvg = (torch.randn((R * S + 256, 4 * D), device=dev) * 0.5).bfloat16()
    pS, pT, pSp = (S + 63) // 64 * 64, (Lt + 63) // 64 * 64, (Ls + 63) // 64 * 64
    vt_self = torch.randn((R, H, 128, pS), device=dev).bfloat16()
    kt = torch.randn((Lt + 128, 48 * D), device=dev).bfloat16(); vt_t = torch.randn((1, H, 128, pT), device=dev).bfloat16()
    ksp = torch.randn((Ls + 128, 48 * D), device=dev).bfloat16(); vt_s = torch.randn((1, H, 128, pSp), device=dev).bfloat16()
    out = torch.zeros((R * S, D), dtype=torch.bfloat16, device=dev)
    rows = [[S] * R, ([Lt, 0, Lt] * R)[:R] if R % 3 == 0 else [Lt] * R, ([Ls, Ls, 0] * R)[:R] if R % 3 == 0 else [Ls] * R]
    nk = torch.tensor(rows, dtype=torch.int32, device=dev)
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
    lib = U.lib()
    for _ in range(3):
        L.check(lib.echo_op_attention_bf16(C.byref(d), U.stream()))
    torch.cuda.synchronize()
    for _ in range(60):      # settle the clock under load
        lib.echo_op_attention_bf16(C.byref(d), U.stream())
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        lib.echo_op_attention_bf16(C.byref(d), U.stream())
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / iters
    keys = sum(sum(r[i] for r in rows) for i in range(R))
    fl = 4.0 * S * keys * D
    print(f"attn R={R} S={S}: {ms*1e3:7.1f} us  {fl/ms/1e9:7.1f} TFLOP/s", flush=True)
    nwg = ((S + 127) // 128) * H * R
    prof = torch.zeros((nwg * 8, 8), dtype=torch.int64, device=dev)
    d.prof = prof.data_ptr()
    L.check(lib.echo_op_attention_bf16(C.byref(d), U.stream()))
    torch.cuda.synchronize()
    pr = prof.cpu().double()
    tiles = pr[:, 3].clamp_min(1)
    print("   per-tile cycles (mean over waves): vmcnt+barrier %.0f (of which vmcnt(0) wait %.0f)  issue+mask %.0f  compute %.0f   | tiles/wave mean %.1f max %.0f" %
          ((pr[:, 0] / tiles).mean(), (pr[:, 4] / tiles).mean(), (pr[:, 1] / tiles).mean(), (pr[:, 2] / tiles).mean(), tiles.mean(), tiles.max()))
    print("   per-wave cycles from kernel start to the end of the tile loop: mean %.0f (tile loop accounts for %.0f); clock %.2f GHz; prologue %.0f" %
          (pr[:, 5].mean(), (pr[:, 0] + pr[:, 1] + pr[:, 2]).mean(), (pr[:, 5] / (pr[:, 6] / 100e6)).mean() / 1e9, (pr[:, 7].long() & 0xFFFFF).double().mean()))
    x = prof.cpu()[:, 7]
    print('   prologue split: setup (args, key counts, Q requested) %.0f | first tiles landed %.0f | first QK done %.0f' % (((x >> 20) & 0xFFFFF).double().mean(), ((x >> 40) & 0xFFFFF).double().mean(), (x & 0xFFFFF).double().mean()))
    big = pr[pr[:, 3] == pr[:, 3].max()]
    print("   longest waves: barrier %.0f issue %.0f compute %.0f (cycles per tile)" % tuple((big[:, i] / big[:, 3]).mean() for i in range(3)))

if __name__ == "__main__":
    run(12); run(4)
