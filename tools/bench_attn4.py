"""Joint-attention micro-benchmark at the sampler's shapes (run on the GPU box): R rows x 16 heads x 640 queries against self 640 |
text 436 | speaker 640 keys with the CFG rows' segment switches.  ECHO_ATTN=4 / 5 time attn4_kernel / attn5_kernel (4 waves x 64 queries) instead of the 8-wave attn_kernel;
both variants are checked against each other by tests/test_gpu_kernels.py, this tool only times."""
import ctypes as C, math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import gpu_util as U
from echo_tts_amd import _lib as L


def run(R, S=640, H=16, Lt=436, Ls=640, iters=30):
    dev = "cuda"
    D = H * 128
    qkvg = (torch.randn((R * S + 256, 4 * D), device=dev) * 0.5).bfloat16()
    pS, pT, pSp = (S + 63) // 64 * 64, (Lt + 63) // 64 * 64, (Ls + 63) // 64 * 64
    vt_self = torch.randn((R, H, 128, pS), device=dev).bfloat16()
    kt = torch.randn((Lt + 128, 48 * D), device=dev).bfloat16(); vt_t = torch.randn((1, H, 128, pT), device=dev).bfloat16()
    ksp = torch.randn((Ls + 128, 48 * D), device=dev).bfloat16(); vt_s = torch.randn((1, H, 128, pSp), device=dev).bfloat16()
    out = torch.zeros((R * S, D), dtype=torch.bfloat16, device=dev)
    B = R // 3 if R % 3 == 0 else R
    rows = [[S] * R, ([Lt] * B + [0] * B + [Lt] * B) if R % 3 == 0 else [Lt] * R, ([Ls] * B + [Ls] * B + [0] * B) if R % 3 == 0 else [Ls] * R]
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
    ts = []
    for _ in range(3):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(iters):
            lib.echo_op_attention_bf16(C.byref(d), U.stream())
        e.record(); torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) / iters)
    ms = sorted(ts)[1]
    keys = sum(sum(r[i] for r in rows) for i in range(R))
    fl = 4.0 * S * keys * D
    if os.environ.get("ECHO_ATTN") == "5" and os.environ.get("PROF"):
        nw = ((S + 255) // 256) * H * R * 4
        prof = torch.zeros((nw, 8), dtype=torch.int64, device=dev)
        d.prof = prof.data_ptr()
        L.check(lib.echo_op_attention_bf16(C.byref(d), U.stream()))
        torch.cuda.synchronize()
        d.prof = None
        raw = prof.cpu()
        nb = (S + 255) // 256
        blk = (torch.arange(raw.shape[0]) // 4) % nb
        for bsel in range(nb):
            rb = raw[(blk == bsel) & (raw[:, 3] > 0)]
            if rb.shape[0]:
                x = rb[:, 7]
                print("   block %d of a (row, head): %d waves, loop start %.0f | loop end %.0f | wave end %.0f cycles, steps %.1f" % (bsel, rb.shape[0], (x & 0xFFFFFFFF).double().mean(), (x >> 32).double().mean(), rb[:, 5].double().mean(), (rb[:, 3] & ((1 << 40) - 1)).double().mean()))
        raw = raw[raw[:, 3] > 0]
        pa, pb, pc = (raw[:, 1] >> 40).double().mean(), (raw[:, 2] >> 40).double().mean(), (raw[:, 4] >> 40).double().mean()
        pr = (raw & ((1 << 40) - 1)).double()
        pr[:, 5:] = raw[:, 5:].double()
        n = pr[:, 3]
        x = raw[:, 7]
        pro, loop_end = (x & 0xFFFFFFFF).double(), (x >> 32).double()
        print("   attn5 per-step cycles (mean over waves): wait+barrier %.0f  phase A %.0f  phase B %.0f  step tail (walk) %.0f | steps/wave %.1f" % ((pr[:, 0] / n).mean(), (pr[:, 1] / n).mean(), (pr[:, 2] / n).mean(), (pr[:, 4] / n).mean(), n.mean()))
        print("   prologue: setup done %.0f | DMA issued %.0f | Q scaled %.0f" % (pa, pb, pc))
        print("   cycles per step incl. everything between the loop's first and last stamp: %.0f" % ((loop_end - pro) / n).mean())
        print("   per wave: setup + first tiles landed %.0f | + A(0), early(0) = loop start %.0f | loop end %.0f | wave end (O stored) %.0f cycles" % (pr[:, 6].mean(), pro.mean(), loop_end.mean(), pr[:, 5].mean()))
    print(f"attn {('v' + os.environ['ECHO_ATTN']) if os.environ.get('ECHO_ATTN') else 'default'} R={R:2d} S={S}: {ms*1e3:7.1f} us  {fl/ms/1e9:7.1f} TFLOP/s  finite={bool(torch.isfinite(out.float()).all())}", flush=True)


if __name__ == "__main__":
    for R in ((24, 1) if os.environ.get("ECHO_ATTN_DIAG") else (24, 12, 8, 6, 4, 3, 1)):
        run(R)
