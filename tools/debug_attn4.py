"""Where does attn4_kernel / attn5_kernel (ECHO_ATTN=4 / 5) differ from an fp32 reference?  Per row / head / 64-query block error maxima at the C2 shape
with the CFG rows' empty segments (debug aid, run on the GPU box)."""
import ctypes as C, math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import gpu_util as U
from echo_tts_amd import _lib as L

def main(R=3, S=640, H=16, Lt=436, Ls=640):
    dev = "cuda"; D = H * 128
    g = torch.Generator(device=dev); g.manual_seed(0)
    rn = lambda *s: torch.randn(s, device=dev, generator=g)
    qkvg = (rn(R * S + 256, 4 * D) * 0.5).bfloat16()
    pS, pT, pSp = (S + 63) // 64 * 64, (Lt + 63) // 64 * 64, (Ls + 63) // 64 * 64
    vt_self = rn(R, H, 128, pS).bfloat16()
    kt, vt_t = rn(Lt + 128, 4 * D).bfloat16(), rn(1, H, 128, pT).bfloat16()
    ksp, vt_s = rn(Ls + 128, 4 * D).bfloat16(), rn(1, H, 128, pSp).bfloat16()
    rows = [[S] * R, [Lt, 0, Lt][:R], [Ls, Ls, 0][:R]] if not os.environ.get('FULLROWS') else [[S] * R, [Lt] * R, [Ls] * R]
    print('tiles per row:', [sum((rows[i][r] + 63) // 64 for i in range(3)) for r in range(R)])
    nk = torch.tensor(rows, dtype=torch.int32, device=dev)
    outs = []
    for _ in range(3):
        out = torch.zeros((R * S, D), dtype=torch.bfloat16, device=dev)
        d = L.EchoAttnDesc()
        d.Q, d.q_ld, d.q_row_stride = qkvg.data_ptr(), 4 * D, S * 4 * D
        d.O, d.o_ld, d.o_row_stride = out.data_ptr(), D, S * D
        d.G, d.g_ld, d.g_row_stride = qkvg.data_ptr() + 3 * D * 2, 4 * D, S * 4 * D
        d.S, d.H, d.rows, d.nseg, d.causal, d.scale = S, H, R, 3, 0, 1 / math.sqrt(128)
        for i, (kp, kld, krs, vt, pitch, shared) in enumerate(((qkvg.data_ptr() + D * 2, 4 * D, S * 4 * D, vt_self, pS, False),
                                                               (kt.data_ptr(), 4 * D, 0, vt_t, pT, True), (ksp.data_ptr(), 4 * D, 0, vt_s, pSp, True))):
            sg = d.seg[i]
            sg.K, sg.k_ld, sg.k_head_stride, sg.k_row_stride = kp, kld, 128, krs
            sg.Vt, sg.vt_ld, sg.vt_head_stride = vt.data_ptr(), pitch, 128 * pitch
            sg.vt_row_stride = 0 if shared else H * 128 * pitch
            sg.nkeys = nk[i].data_ptr(); sg.kv_mod = 1 if shared else 0
        L.check(U.lib().echo_op_attention_bf16(C.byref(d), U.stream()))
        torch.cuda.synchronize()
        outs.append(out.clone())
    print("bit-identical runs:", torch.equal(outs[0], outs[1]), torch.equal(outs[0], outs[2]))
    for r in range(R):
        q = qkvg[r * S:(r + 1) * S, :D].float().view(S, H, 128).transpose(0, 1)
        ks = [qkvg[r * S:(r + 1) * S, D:2 * D].float().view(S, H, 128).transpose(0, 1)]
        vs = [vt_self[r, :, :, :S].float().transpose(1, 2)]
        if rows[1][r]: ks.append(kt[:Lt, :D].float().view(Lt, H, 128).transpose(0, 1)); vs.append(vt_t[0, :, :, :Lt].float().transpose(1, 2))
        if rows[2][r]: ks.append(ksp[:Ls, :D].float().view(Ls, H, 128).transpose(0, 1)); vs.append(vt_s[0, :, :, :Ls].float().transpose(1, 2))
        k, v = torch.cat(ks, 1), torch.cat(vs, 1)
        ref = torch.softmax(q @ k.transpose(1, 2) / math.sqrt(128), -1) @ v           # H, S, 128
        gate = torch.sigmoid(qkvg[r * S:(r + 1) * S, 3 * D:].float()).bfloat16().float().view(S, H, 128).transpose(0, 1)
        ref = ref.bfloat16().float() * gate
        got = outs[0][r * S:(r + 1) * S].float().view(S, H, 128).transpose(0, 1)
        err = (got - ref).abs()                                                        # H, S, 128
        eb = err.view(H, S // 32, 32, 128).amax((2, 3))                                # H, S/32
        print(f"row {r}: max err {float(err.max()):.4f}; per 32-query block (max over heads):", [round(float(x), 3) for x in eb.amax(0)])
        print("   per head:", [round(float(x), 3) for x in eb.amax(1)])
        dd = (outs[0] != outs[1])[r * S:(r + 1) * S].view(S // 32, 32, D).any(2).any(1)
        print("   32-query blocks differing between runs:", [int(i) for i in dd.nonzero().flatten()])

main(Lt=int(os.environ.get('LT', '436')), Ls=int(os.environ.get('LS', '640')))
