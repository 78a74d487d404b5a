"""GPU: each hand-written HIP kernel against a plain PyTorch fp32/fp64 evaluation of the same op,
called through the C ABI (echo_op_*).  bf16 kernels are compared with the fp32 result rounded at
the reference's rounding points; fp32 kernels with an fp64 evaluation."""
import ctypes as C
import math
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

from tests import gpu_util as U  # noqa: E402
from echo_tts_amd import _lib as L  # noqa: E402

DEV = U.DEV


def rnd(*shape, dtype=torch.float32, scale=1.0, seed=0):
    g = torch.Generator().manual_seed(seed + sum(shape))
    return (torch.randn(shape, generator=g) * scale).to(dtype).to(DEV)


# --------------------------------------------------------------------------- GEMM
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (200, 256, 192), (640, 384, 2048), (37, 80, 128), (1920, 256, 512)])
def test_gemm_bf16_plain(M, N, K):
    A = rnd(M, K, dtype=torch.bfloat16)
    W = U.pad_rows(rnd(N, K, dtype=torch.bfloat16, seed=1))
    out = torch.full((M, N), 7.0, dtype=torch.bfloat16, device=DEV)
    U.gemm(A, W, out, M=M, N=N, K=K, lda=K, ldw=K, ldc=N)
    ref = (A.float() @ W[:N].float().T).bfloat16()
    U.bf16_close(out, ref, ulps=1.01, atol=1e-3, frac_exact=0.98)


@pytest.mark.parametrize("M,N,K", [(128, 128, 32), (200, 256, 160), (77, 96, 96), (640, 1024, 1024)])
def test_gemm_f32_matches_fp64(M, N, K):
    A = rnd(M, K)
    W = U.pad_rows(rnd(N, K, seed=1))
    out = torch.zeros((M, N), device=DEV)
    U.gemm(A, W, out, M=M, N=N, K=K, lda=K, ldw=K, ldc=N)
    ref = (A.double() @ W[:N].double().T)
    err = (out.double() - ref).abs().max().item()
    assert err < 2e-5 * math.sqrt(K), err


def test_gemm_bf16_epilogue_chain():
    """bias -> round -> /6 -> silu -> colscale -> +res, each rounded to bf16 (model.py:461-462, 385-388)."""
    M, N, K = 300, 256, 128
    A, W = rnd(M, K, dtype=torch.bfloat16), U.pad_rows(rnd(N, K, dtype=torch.bfloat16, seed=1))
    bias, cs = rnd(N, dtype=torch.bfloat16, seed=2), rnd(N, dtype=torch.bfloat16, seed=3)
    res = rnd(M, N, dtype=torch.bfloat16, seed=4)
    out = res.clone()
    U.gemm(A, W, out, M=M, N=N, K=K, lda=K, ldw=K, ldc=N, bias=bias, div=6.0, act=1, colscale=cs, res=out, ldres=N)
    y = (A.float() @ W[:N].float().T + bias.float()).bfloat16()
    y = (y.float() / 6.0).bfloat16()
    y = torch.nn.functional.silu(y.float()).bfloat16()
    y = (y.float() * cs.float()).bfloat16()
    y = (y.float() + res.float()).bfloat16()
    U.bf16_close(out, y, ulps=2.0, atol=2e-3)


def test_gemm_bf16_swiglu():
    M, F, K = 256, 192, 256
    A = rnd(M, K, dtype=torch.bfloat16)
    w1, w3 = rnd(F, K, dtype=torch.bfloat16, seed=1, scale=0.1), rnd(F, K, dtype=torch.bfloat16, seed=2, scale=0.1)
    Wp = U.pack_swiglu(w1, w3)
    out = torch.zeros((M, F), dtype=torch.bfloat16, device=DEV)
    U.gemm(A, Wp, out, M=M, N=2 * F, K=K, lda=K, ldw=K, ldc=F, swiglu=1, Npad=Wp.shape[0])
    a = (A.float() @ w1.float().T).bfloat16()
    b = (A.float() @ w3.float().T).bfloat16()
    ref = (torch.nn.functional.silu(a.float()).bfloat16().float() * b.float()).bfloat16()
    U.bf16_close(out, ref, ulps=2.0, atol=2e-3)


def test_gemm_f32_batched_two_level():
    """z -> (zo, zi) strides, as used for the AdaLN low-rank refinements and per-head attention GEMMs."""
    nbo, nbi, M, N, K = 3, 2, 70, 64, 64
    A = rnd(nbo, nbi, M, K)
    W = rnd(nbo, nbi, 128, K, seed=1)
    out = torch.zeros((nbo, nbi, M, N), device=DEV)
    U.gemm(A, W, out, M=M, N=N, K=K, lda=K, ldw=K, ldc=N, nbatch=nbo * nbi, nbi=nbi, a_bo=nbi * M * K, a_bi=M * K,
           w_bo=nbi * 128 * K, w_bi=128 * K, c_bo=nbi * M * N, c_bi=M * N, acc_scale=0.5)
    ref = 0.5 * torch.einsum("abmk,abnk->abmn", A.double(), W[:, :, :N].double())
    assert (out.double() - ref).abs().max().item() < 1e-4


@pytest.mark.parametrize("k,dil", [(7, 1), (7, 3), (7, 9), (1, 1)])
def test_gemm_f32_causal_conv_taps(k, dil):
    """taps loop == causal Conv1d on channels-last rows (autoencoder.py:264-289), incl. Snake second output."""
    T, Ci, Co = 300, 64, 96
    x = rnd(1, Ci, T, scale=1.0)
    w = rnd(Co, Ci, k, seed=1, scale=0.1)
    b = rnd(Co, seed=2, scale=0.1)
    alpha = 1.0 + 0.2 * rnd(Co, seed=3)
    pad = (k - 1) * dil
    ref = torch.nn.functional.conv1d(torch.nn.functional.pad(x.double(), (pad, 0)), w.double(), b.double(), dilation=dil)
    ref_cl = ref[0].T  # (T, Co)
    xcl = torch.zeros((64 + T, Ci), device=DEV)
    xcl[64:] = x[0].T
    Wg = U.pad_rows(w.permute(0, 2, 1).reshape(Co, k * Ci).contiguous())
    out = torch.zeros((T, Co), device=DEV)
    out2 = torch.zeros((T, Co), device=DEV)
    U.gemm(xcl, Wg, out, C2=out2, M=T, N=Co, K=Ci, lda=Ci, ldw=k * Ci, ldc=Co, taps=k, tap_base=-(k - 1) * dil, tap_shift=dil,
           bias=b, snake_alpha=alpha, a_offset_elems=64 * Ci)
    assert (out.double() - ref_cl).abs().max().item() < 1e-4
    sref = ref_cl + (1.0 / (alpha.double() + 1e-9)) * torch.sin(alpha.double() * ref_cl) ** 2
    assert (out2.double() - sref).abs().max().item() < 1e-4


@pytest.mark.parametrize("r", [2, 8])
def test_gemm_f32_causal_conv_transpose(r):
    """2-tap GEMM with N = r*Co == causal ConvTranspose1d k=2r, stride r (autoencoder.py:300-316)."""
    from echo_tts_amd.autoencoder import _convT_as_gemm
    T, Ci, Co = 200, 64, 32
    x = rnd(1, Ci, T)
    w = rnd(Ci, Co, 2 * r, seed=1, scale=0.1)
    b = rnd(Co, seed=2, scale=0.1)
    ref = torch.nn.functional.conv_transpose1d(x.double(), w.double(), b.double(), stride=r)[..., : T * r]
    ref_cl = ref[0].T  # (T*r, Co)
    xcl = torch.zeros((64 + T, Ci), device=DEV)
    xcl[64:] = x[0].T
    Wg = U.pad_rows(_convT_as_gemm(w, r))
    out = torch.zeros((T * r, Co), device=DEV)
    U.gemm(xcl, Wg, out, M=T, N=r * Co, K=Ci, lda=Ci, ldw=2 * Ci, ldc=r * Co, taps=2, tap_base=-1, tap_shift=1, bias=b,
           vec_mod=Co, a_offset_elems=64 * Ci)
    assert (out.double() - ref_cl).abs().max().item() < 1e-4


# --------------------------------------------------------------------------- attention
def _ref_attention(q, segs, scale, causal, gate):
    """q (R,S,H,128) fp32; segs: list of (K (R,Lk,H,128), V, mask (R,Lk) bool).  fp32 softmax attention."""
    K = torch.cat([s[0] for s in segs], 1)
    V = torch.cat([s[1] for s in segs], 1)
    m = torch.cat([s[2] for s in segs], 1)[:, None, None]
    if causal:
        S = q.shape[1]
        m = m & torch.tril(torch.ones(S, K.shape[1], dtype=torch.bool, device=q.device))[None, None]
    o = torch.nn.functional.scaled_dot_product_attention(q.transpose(1, 2), K.transpose(1, 2), V.transpose(1, 2), attn_mask=m,
                                                         scale=scale).transpose(1, 2)
    o = o.bfloat16().float()
    if gate is not None:
        o = (o * torch.sigmoid(gate.float()).bfloat16().float()).bfloat16().float()
    return o


def _to_vt(v, pitch):
    """(Rk, Lk, H, 128) -> (Rk, H, 128, pitch) zero padded."""
    Rk, Lk, H, D = v.shape
    vt = torch.zeros((Rk, H, D, pitch), dtype=v.dtype, device=v.device)
    vt[..., :Lk] = v.permute(0, 2, 3, 1)
    return vt.contiguous()


@pytest.mark.parametrize("S,Lt,Ls,use_bias,spike", [(200, 70, 33, False, 0), (128, 64, 128, True, 0), (640, 436, 640, False, 0),
                                                     (640, 436, 640, False, 2), (640, 436, 640, False, 20), (300, 64, 7, False, 20)])
def test_attention_bf16_joint_segments(S, Lt, Ls, use_bias, spike):
    """Joint attention over self | text | speaker segments with the CFG rows' segment switches, ragged segment ends and (use_bias)
    an irregular key mask, against fp32 SDPA.  Without a bias the fast kernel (attn4_kernel: fixed softmax reference per stream)
    runs; `spike` plants keys in a LATER tile whose score with one query is `spike` x |q|^2 (cdna guide rule 26: force the rare
    branch): spike 2 stays inside the fast kernel's range (P up to 2^32 against the first tile's reference), spike 20 leaves it, so
    the workgroup must be flagged and redone by attn_kernel with per-tile rescaling - that query's output is then one value row.
    The default build runs attn_kernel (deferred rescale: the spikes force its rescale branch); tests/test_gpu_kernels.py::
    test_attention_fast_kernel_variant repeats the spike cases in child processes with ECHO_ATTN=4 and 5."""
    R, H = 3, 2
    q = rnd(R, S, H, 128, dtype=torch.bfloat16)
    k_self, v_self = rnd(R, S, H, 128, dtype=torch.bfloat16, seed=1), rnd(R, S, H, 128, dtype=torch.bfloat16, seed=2)
    if spike:
        k_self[0, min(S - 1, 150), 0] = (spike * q[0, 7, 0].float()).bfloat16()           # row 0, head 0: query 7 meets it in its third tile
        k_self[1, S - 1, 1] = (spike * q[1, S - 2, 1].float()).bfloat16()                # row 1, head 1: in the last tile of the self segment
    k_t, v_t = rnd(1, Lt, H, 128, dtype=torch.bfloat16, seed=3), rnd(1, Lt, H, 128, dtype=torch.bfloat16, seed=4)
    k_s, v_s = rnd(1, Ls, H, 128, dtype=torch.bfloat16, seed=5), rnd(1, Ls, H, 128, dtype=torch.bfloat16, seed=6)
    gate = rnd(R, S, H * 128, dtype=torch.bfloat16, seed=7)
    # CFG rows: row 1 drops text, row 2 drops speaker; text uses a shorter valid prefix
    nt = Lt - 5
    tmask = torch.zeros((R, Lt), dtype=torch.bool, device=DEV)
    tmask[0, :nt] = True
    tmask[2, :nt] = True
    if use_bias:
        tmask[0, 3] = False
        tmask[2, 3] = False
    smask = torch.ones((R, Ls), dtype=torch.bool, device=DEV)
    smask[2] = False
    ones = torch.ones((R, S), dtype=torch.bool, device=DEV)
    ref = _ref_attention(q.float(), [(k_self.float(), v_self.float(), ones),
                                     (k_t.float().expand(R, -1, -1, -1), v_t.float().expand(R, -1, -1, -1), tmask),
                                     (k_s.float().expand(R, -1, -1, -1), v_s.float().expand(R, -1, -1, -1), smask)],
                         1 / math.sqrt(128), False, gate.view(R, S, H, 128))
    out = torch.zeros((R, S, H * 128), dtype=torch.bfloat16, device=DEV)
    pS, pT, pSp = (S + 63) // 64 * 64, (Lt + 63) // 64 * 64, (Ls + 63) // 64 * 64
    vt_self, vt_t, vt_s = _to_vt(v_self, pS), _to_vt(v_t, pT), _to_vt(v_s, pSp)
    nk = torch.tensor([[S, S, S], [nt, 0, nt], [Ls, Ls, 0]], dtype=torch.int32, device=DEV)
    bias = None
    if use_bias:
        bias = torch.zeros((1, Lt), device=DEV)
        bias[0, 3] = float("-inf")
    d = L.EchoAttnDesc()
    d.Q, d.q_ld, d.q_row_stride = q.data_ptr(), H * 128, S * H * 128
    d.O, d.o_ld, d.o_row_stride = out.data_ptr(), H * 128, S * H * 128
    d.G, d.g_ld, d.g_row_stride = gate.data_ptr(), H * 128, S * H * 128
    d.S, d.H, d.rows, d.nseg, d.causal, d.scale = S, H, R, 3, 0, 1 / math.sqrt(128)
    for i, (kk, vt, pitch, shared) in enumerate(((k_self, vt_self, pS, False), (k_t, vt_t, pT, True), (k_s, vt_s, pSp, True))):
        sg = d.seg[i]
        sg.K, sg.k_ld, sg.k_head_stride = kk.data_ptr(), H * 128, 128
        sg.k_row_stride = 0 if shared else kk.shape[1] * H * 128
        sg.Vt, sg.vt_ld, sg.vt_head_stride = vt.data_ptr(), pitch, 128 * pitch
        sg.vt_row_stride = 0 if shared else H * 128 * pitch
        sg.nkeys = nk[i].data_ptr()
        sg.kv_mod = 1 if shared else 0
        if i == 1 and bias is not None:
            sg.bias, sg.bias_row_stride = bias.data_ptr(), Lt
    L.check(U.lib().echo_op_attention_bf16(C.byref(d), U.stream()))
    torch.cuda.synchronize()
    refr = ref.reshape(R, S, H * 128)
    err = (out.float() - refr).abs()
    # bf16 outputs: one ulp of a value in [4, 8) is 2^-5 - the bound is 3e-2 plus one bf16 ulp of the reference's magnitude
    assert float((err - refr.abs() * 2.0 ** -7).max()) < 3e-2, float(err.max())
    assert float(err.mean()) < 2e-3, float(err.mean())
    # e4m3 output (ABI 6, fp8 engine with a static activation scale): the same launch with O8 writes e4m3(bf16(out) * inv), saturating
    # ABI 6, g_activated: the gate tensor already holds bf16(sigmoid(gate)) (what the QKVG tail stores with qkv_gate_act): same output up to
    # the rounding of torch's sigmoid against the kernels' v_exp + v_rcp form
    act = torch.sigmoid(gate.float()).bfloat16()
    out_act = torch.zeros_like(out)
    d.G, d.O, d.g_activated = act.data_ptr(), out_act.data_ptr(), 1
    L.check(U.lib().echo_op_attention_bf16(C.byref(d), U.stream()))
    torch.cuda.synchronize()
    d.G, d.O, d.g_activated = gate.data_ptr(), out.data_ptr(), 0
    U.bf16_close(out_act, out, ulps=2.0, atol=1e-6, frac_exact=0.98)
    _check_attention_fp8_output(d, out, R * S, H * 128, inv=448.0 / float(out.float().abs().max()) * 1.5)      # the largest values saturate at 448


def _check_attention_fp8_output(d, out_bf16, rows, width, inv):
    """Re-launches the attention described by `d` with an e4m3 destination and compares the bytes with a torch quantisation of the
    bf16 output the first launch produced (v_cvt_pk_fp8_f32 vs torch: near-ties may land on the neighbouring code, see
    test_quant_rows_fp8_matches_torch_e4m3); the bf16 destination must stay untouched."""
    import os
    if os.environ.get("ECHO_ATTN") == "4":
        return      # attn4_kernel has no e4m3 output: the launcher takes attn5_kernel for it, whose bf16 roundings differ from attn4's in places
    o8 = torch.full((rows, width + 4), 0x55, dtype=torch.uint8, device=DEV)
    keep = out_bf16.clone()
    d.O8, d.o8_ld, d.o8_row_stride, d.o8_inv = o8.data_ptr(), width + 4, (rows // d.rows) * (width + 4), inv
    L.check(U.lib().echo_op_attention_bf16(C.byref(d), U.stream()))
    torch.cuda.synchronize()
    d.O8 = None
    assert torch.equal(out_bf16, keep), "the bf16 output was written although an e4m3 destination was given"
    assert bool((o8[:, width:] == 0x55).all()), "bytes beyond the row were written"
    want = (keep.float().reshape(rows, width) * inv).clamp(-448.0, 448.0)
    got, ref = _deq(o8[:, :width].contiguous()), _deq(_to_e4m3_bytes(want))
    assert bool(torch.isfinite(got).all())
    bad = got != ref
    assert float(bad.float().mean()) < 2e-3, float(bad.float().mean())
    step = ref.abs().clamp_min(2.0 ** -6) * 0.126
    assert bool(((got - ref).abs()[bad] <= step[bad] * 1.0001).all())
    assert float((got.abs() == 448.0).float().mean()) > 0, "the saturation branch was not exercised"


def test_attention_fp8_output_two_stream_blocks():
    """The 256-query workgroups of attn5_kernel (two query streams per wave) + its short last block with the e4m3 output: a grid of more
    than one round (6 rows x 16 heads x 3 blocks of 128 > 256 workgroups), S = 320 = one 256-query block + one of 64."""
    R, S, H, Lt = 6, 320, 16, 70
    D = H * 128
    qkvg = rnd(R * S + 256, 4 * D, dtype=torch.bfloat16, scale=0.5)
    pS, pT = (S + 63) // 64 * 64, (Lt + 63) // 64 * 64
    vt_self = rnd(R, H, 128, pS, dtype=torch.bfloat16, seed=1)
    kt, vt_t = rnd(Lt + 128, 4 * D, dtype=torch.bfloat16, seed=2), rnd(1, H, 128, pT, dtype=torch.bfloat16, seed=3)
    nk = torch.tensor([[S] * R, [Lt, 0, Lt, Lt, 0, Lt]], dtype=torch.int32, device=DEV)
    out = torch.zeros((R * S, D), dtype=torch.bfloat16, device=DEV)
    d = L.EchoAttnDesc()
    d.Q, d.q_ld, d.q_row_stride = qkvg.data_ptr(), 4 * D, S * 4 * D
    d.O, d.o_ld, d.o_row_stride = out.data_ptr(), D, S * D
    d.G, d.g_ld, d.g_row_stride = qkvg.data_ptr() + 3 * D * 2, 4 * D, S * 4 * D
    d.S, d.H, d.rows, d.nseg, d.causal, d.scale = S, H, R, 2, 0, 1 / math.sqrt(128)
    for i, (kp, kld, krs, vt, pitch, shared) in enumerate(((qkvg.data_ptr() + D * 2, 4 * D, S * 4 * D, vt_self, pS, False),
                                                           (kt.data_ptr(), 4 * D, 0, vt_t, pT, True))):
        sg = d.seg[i]
        sg.K, sg.k_ld, sg.k_head_stride, sg.k_row_stride = kp, kld, 128, krs
        sg.Vt, sg.vt_ld, sg.vt_head_stride = vt.data_ptr(), pitch, 128 * pitch
        sg.vt_row_stride = 0 if shared else H * 128 * pitch
        sg.nkeys = nk[i].data_ptr()
        sg.kv_mod = 1 if shared else 0
    L.check(U.lib().echo_op_attention_bf16(C.byref(d), U.stream()))
    torch.cuda.synchronize()
    assert bool(out.any())
    _check_attention_fp8_output(d, out, R * S, D, inv=448.0 / float(out.float().abs().max()) * 1.5)


def test_attention_range_fallback_many_blocks():
    """attn_redo_kernel with several report words per workgroup: 6 rows x 16 heads x 2 blocks of 256 queries = 192 words over 32 workgroups
    (6 each).  Keys whose score with one query is 20 x |q|^2 are planted so that flagged words share a scan chunk, sit in its first and
    last word, cover a full 256-query block (two 128-query passes of attn_kernel's body) and the short last block; every query of the
    flagged (row, head) pairs and of two untouched ones is checked against fp32."""
    R, S, H, Lt = 6, 320, 16, 70
    D = H * 128
    qkvg = rnd(R * S + 256, 4 * D, dtype=torch.bfloat16, scale=0.5)
    pS, pT = (S + 63) // 64 * 64, (Lt + 63) // 64 * 64
    v_self = rnd(R, S, H, 128, dtype=torch.bfloat16, seed=1)
    vt_self = v_self.permute(0, 2, 3, 1).contiguous()
    vt_self = torch.nn.functional.pad(vt_self, (0, pS - S)).contiguous()
    kt, v_t = rnd(Lt + 128, 4 * D, dtype=torch.bfloat16, seed=2), rnd(1, Lt, H, 128, dtype=torch.bfloat16, seed=3)
    vt_t = torch.nn.functional.pad(v_t.permute(0, 2, 3, 1).contiguous(), (0, pT - Lt)).contiguous()
    # (row, head, query, key position): words (row * 16 + head) * 2 + block; words 0 and 5 share workgroup 0's chunk, word 191 is the last
    plants = [(0, 0, 7, 150), (0, 2, 300, 310), (0, 2, 40, 100), (3, 7, 200, 20), (5, 15, 319, 318)]
    qv = qkvg[:R * S].view(R, S, 4 * D)
    for (r, h, qi, kj) in plants:
        qv[r, kj, D + h * 128:D + (h + 1) * 128] = (20.0 * qv[r, qi, h * 128:(h + 1) * 128].float()).bfloat16()
    tx = [Lt, 0, Lt, Lt, 0, Lt]
    nk = torch.tensor([[S] * R, tx], dtype=torch.int32, device=DEV)
    out = torch.zeros((R * S, D), dtype=torch.bfloat16, device=DEV)
    d = L.EchoAttnDesc()
    d.Q, d.q_ld, d.q_row_stride = qkvg.data_ptr(), 4 * D, S * 4 * D
    d.O, d.o_ld, d.o_row_stride = out.data_ptr(), D, S * D
    d.S, d.H, d.rows, d.nseg, d.causal, d.scale = S, H, R, 2, 0, 1 / math.sqrt(128)
    for i, (kp, kld, krs, vt, pitch, shared) in enumerate(((qkvg.data_ptr() + D * 2, 4 * D, S * 4 * D, vt_self, pS, False),
                                                           (kt.data_ptr(), 4 * D, 0, vt_t, pT, True))):
        sg = d.seg[i]
        sg.K, sg.k_ld, sg.k_head_stride, sg.k_row_stride = kp, kld, 128, krs
        sg.Vt, sg.vt_ld, sg.vt_head_stride = vt.data_ptr(), pitch, 128 * pitch
        sg.vt_row_stride = 0 if shared else H * 128 * pitch
        sg.nkeys = nk[i].data_ptr()
        sg.kv_mod = 1 if shared else 0
    redo = torch.zeros((R * H * ((S + 127) // 128),), dtype=torch.int32, device=DEV)
    d.redo = redo.data_ptr()
    L.check(U.lib().echo_op_attention_bf16(C.byref(d), U.stream()))
    torch.cuda.synchronize()
    assert bool(torch.isfinite(out.float()).all())
    import os
    if os.environ.get("ECHO_ATTN", "5") in ("4", "5"):
        words = redo[:R * H * 2].view(R, H, 2)        # the fast kernels' report: one word per 256-query block
        flagged = {(r, h, qi // 256) for (r, h, qi, _) in plants}
        seen = {tuple(int(x) for x in ix) for ix in torch.nonzero(words).tolist()}
        # attn5_kernel (no reference point) flags exactly the planted blocks; attn4_kernel's reference is the first tile's maximum, so a
        # key planted IN the first tile (3, 7) raises the reference instead of leaving the range
        assert seen == flagged if os.environ.get("ECHO_ATTN", "5") == "5" else (seen <= flagged and len(seen) >= 3), sorted(seen)
    for (r, h) in sorted({(r, h) for (r, h, _, _) in plants} | {(1, 3), (4, 9)}):
        q = qv[r, :, h * 128:(h + 1) * 128].float()
        k = qv[r, :, D + h * 128:D + (h + 1) * 128].float()
        v = v_self[r, :, h].float()
        if tx[r]:
            k = torch.cat([k, kt[:Lt, h * 128:(h + 1) * 128].float()])
            v = torch.cat([v, v_t[0, :, h].float()])
        ref = torch.softmax(q @ k.T / math.sqrt(128), -1) @ v
        got = out.view(R, S, D)[r, :, h * 128:(h + 1) * 128].float()
        err = (got - ref).abs()
        assert float((err - ref.abs() * 2.0 ** -7).max()) < 3e-2 and float(err.mean()) < 2e-3, (r, h, float(err.max()), float(err.mean()))
    for (r, h, qi, kj) in plants:      # the planted key dominates its query: the output row is that key's value row
        got = out.view(R, S, D)[r, qi, h * 128:(h + 1) * 128].float()
        assert float((got - v_self[r, kj, h].float()).abs().max()) < 2e-2, (r, h, qi)


@pytest.mark.parametrize("variant", ["1", "4", "5"])
def test_attention_fast_kernel_variant(variant):
    """attn_kernel / attn4_kernel / attn5_kernel forced (ECHO_ATTN=1 / 4 / 5, read once per process; the default picks by grid size): the
    joint-segment cases incl. the range-fallback spikes and the tile counts of every remainder class of their unrolled loops, in a child process."""
    import os, subprocess, sys
    if os.environ.get("ECHO_ATTN"):
        pytest.skip("already a variant run")
    env = dict(os.environ, ECHO_ATTN=variant)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gpu_kernels.py"), "-m", "gpu", "-q", "-x", "-k",
                        "joint_segments or deterministic_at_full_size or attention_tile_count_classes or range_fallback_many"], env=env, cwd=root, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


@pytest.mark.parametrize("S", [100, 256])
def test_attention_bf16_causal_self(S):
    R, H = 2, 3
    q, k, v = (rnd(R, S, H, 128, dtype=torch.bfloat16, seed=i) for i in range(3))
    ones = torch.ones((R, S), dtype=torch.bool, device=DEV)
    ref = _ref_attention(q.float(), [(k.float(), v.float(), ones)], 1 / math.sqrt(128), True, None)
    out = torch.zeros((R, S, H * 128), dtype=torch.bfloat16, device=DEV)
    pS = (S + 63) // 64 * 64
    vt = _to_vt(v, pS)
    nk = torch.full((R,), S, dtype=torch.int32, device=DEV)
    d = L.EchoAttnDesc()
    d.Q, d.q_ld, d.q_row_stride = q.data_ptr(), H * 128, S * H * 128
    d.O, d.o_ld, d.o_row_stride = out.data_ptr(), H * 128, S * H * 128
    d.S, d.H, d.rows, d.nseg, d.causal, d.scale = S, H, R, 1, 1, 1 / math.sqrt(128)
    sg = d.seg[0]
    sg.K, sg.k_ld, sg.k_head_stride, sg.k_row_stride = k.data_ptr(), H * 128, 128, S * H * 128
    sg.Vt, sg.vt_ld, sg.vt_head_stride, sg.vt_row_stride = vt.data_ptr(), pS, 128 * pS, H * 128 * pS
    sg.nkeys = nk.data_ptr()
    L.check(U.lib().echo_op_attention_bf16(C.byref(d), U.stream()))
    torch.cuda.synchronize()
    err = (out.float() - ref.reshape(R, S, H * 128)).abs()
    assert float(err.max()) < 3e-2 and float(err.mean()) < 2e-3, (float(err.max()), float(err.mean()))


# --------------------------------------------------------------------------- row kernels
@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("D", [256, 1280, 2048])
def test_norm_modes(dt, D):
    rows = 37
    x = rnd(rows, D, dtype=dt)
    w0, w1 = (1 + 0.1 * rnd(D, seed=1)).to(dt), (0.1 * rnd(D, seed=2)).to(dt)
    xf = x.float()
    rs = torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + 1e-5)
    refs = {
        0: (xf * rs * w0.float() + w1.float()).to(dt),
        1: (xf * rs * w0.float()).to(dt),
        2: ((xf * rs).to(dt) * w0).to(dt),
        3: torch.nn.functional.layer_norm(xf, (D,), w0.float(), w1.float(), 1e-5).to(dt),
    }
    for mode, ref in refs.items():
        y = torch.zeros_like(x)
        L.check(U.lib().echo_op_norm(U.code(x), mode, x.data_ptr(), D, y.data_ptr(), D, rows, D, 1e-5, w0.data_ptr(), w1.data_ptr(),
                                     U.stream()))
        if dt == torch.bfloat16:
            U.bf16_close(y, ref, ulps=1.01, atol=1e-3)
        else:
            assert (y - ref).abs().max().item() < 2e-5, mode


@pytest.mark.parametrize("rows", [4096, 5003, 15360])
def test_norm_adaln_rows_per_wave_kernel_equals_the_generic_one(rows):
    """AdaLN-apply at the EchoDiT width with >= 4096 rows runs norm_adaln_rows_kernel (two rows per wave, the modulation vectors in
    registers; +35 % bandwidth, tools/bench_norm.py).  Same arithmetic and summation order as the generic kernel: its output must be BIT-equal
    to the generic kernel's on the same rows (launched in slices of < 4096 rows, which take the generic path), incl. a row count that is
    not a multiple of the 8 rows of a workgroup, strided inputs, and it must stay within 1 bf16 ulp of the torch evaluation."""
    D = 2048
    xbig = rnd(rows, D + 64, dtype=torch.bfloat16, seed=7)
    x = xbig[:, :D]                                        # row pitch D + 64: the kernels take ldx
    w0, w1 = (1 + 0.1 * rnd(D, seed=1)).bfloat16(), (0.1 * rnd(D, seed=2)).bfloat16()
    y = torch.zeros((rows, D), dtype=torch.bfloat16, device=DEV)
    L.check(U.lib().echo_op_norm(L.ECHO_BF16, 0, x.data_ptr(), D + 64, y.data_ptr(), D, rows, D, 1e-5, w0.data_ptr(), w1.data_ptr(), U.stream()))
    yg = torch.zeros_like(y)
    for r0 in range(0, rows, 2048):
        n = min(2048, rows - r0)
        L.check(U.lib().echo_op_norm(L.ECHO_BF16, 0, x[r0:].data_ptr(), D + 64, yg[r0:].data_ptr(), D, n, D, 1e-5, w0.data_ptr(), w1.data_ptr(), U.stream()))
    assert torch.equal(y, yg)
    xf = x.float()
    ref = (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + 1e-5) * w0.float() + w1.float()).bfloat16()
    U.bf16_close(y, ref, ulps=1.01, atol=1e-3)
    # the fp8 engine's fused variant (norm + e4m3 row quantisation), same slicing: codes and scales bit-equal
    q, sc = torch.zeros((rows, D), dtype=torch.uint8, device=DEV), torch.zeros((rows,), dtype=torch.float32, device=DEV)
    qg, sg = torch.zeros_like(q), torch.zeros_like(sc)
    L.check(U.lib().echo_op_norm_adaln_fp8(x.data_ptr(), D + 64, q.data_ptr(), D, sc.data_ptr(), rows, D, 1e-5, w0.data_ptr(), w1.data_ptr(), U.stream()))
    for r0 in range(0, rows, 2048):
        n = min(2048, rows - r0)
        L.check(U.lib().echo_op_norm_adaln_fp8(x[r0:].data_ptr(), D + 64, qg[r0:].data_ptr(), D, sg[r0:].data_ptr(), n, D, 1e-5, w0.data_ptr(),
                                               w1.data_ptr(), U.stream()))
    assert torch.equal(q, qg) and torch.equal(sc, sg)
    q2, s2 = U.quant_rows_fp8(y)                          # = norm kernel followed by the row quantiser
    assert torch.equal(q, q2) and torch.equal(sc, s2)


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float32])
def test_headnorm_rope_half_heads(dt):
    """q_norm/k_norm + RoPE on heads [0, H/2) at positions start_pos + s (model.py:199-232)."""
    from oracle import echo_ref as R
    rows, S, H, start = 2, 50, 4, 7
    D = H * 128
    x = rnd(rows * S, 2 * D, dtype=dt)           # [q | k] side by side, like the QKVG buffer
    w = (1 + 0.1 * rnd(2, H, 128, seed=1)).to(dt)
    fc = R.rope_table(128, start + S)
    ref = []
    for ti in range(2):
        xt = x[:, ti * D:(ti + 1) * D].reshape(rows, S, H, 128).cpu()
        xn = R.rms_norm(xt, w[ti].cpu(), 1e-5)
        ref.append(R.rotate_first_half_of_heads(xn, fc[start:start + S]).reshape(rows * S, D))
    ref = torch.cat(ref, 1)
    table = torch.view_as_real(fc).contiguous().to(DEV)
    L.check(U.lib().echo_op_headnorm_rope(U.code(x), x.data_ptr(), 2 * D, D, 2, rows * S, S, H, w.data_ptr(), D, 1e-5, 1, H // 2,
                                          table.data_ptr(), start, 1, U.stream()))
    if dt == torch.bfloat16:
        U.bf16_close(x.cpu(), ref, ulps=1.01, atol=1e-3)
    else:
        assert (x.cpu() - ref).abs().max().item() < 1e-5


@pytest.mark.parametrize("dt,HD", [(torch.bfloat16, 128), (torch.float32, 128), (torch.float32, 64)])
def test_transpose_heads(dt, HD):
    B, S, H = 2, 150, 3
    v = rnd(B * S, H * HD + 16, dtype=dt)
    pitch = (S + 63) // 64 * 64
    vt = torch.full((B, H, HD, pitch), 5.0, dtype=dt, device=DEV)
    L.check(U.lib().echo_op_transpose_heads(U.code(v), v.data_ptr(), H * HD + 16, vt.data_ptr(), pitch, H * HD * pitch, B, S, H, HD,
                                            U.stream()))
    ref = v[:, :H * HD].reshape(B, S, H, HD).permute(0, 2, 3, 1)
    assert torch.equal(vt[..., :S], ref)
    assert bool((vt[..., S:] == 0).all())


@pytest.mark.parametrize("cfg", [0, 1, 2, 3, 4, 6, 7, 8, 9])
@pytest.mark.parametrize("ksplit", [1, 2, 4])
def test_gemm_all_tile_configs_and_splitk(cfg, ksplit):
    """Every tile configuration / pipeline depth / split-K factor gives the same result (bf16 epilogue, fp32 taps, SwiGLU)."""
    M, N, K = 300, 384, 512
    A, W = rnd(M, K, dtype=torch.bfloat16), U.pad_rows(rnd(N, K, dtype=torch.bfloat16, seed=1))
    bias, cs = rnd(N, dtype=torch.bfloat16, seed=2), rnd(N, dtype=torch.bfloat16, seed=3)
    res = rnd(M, N, dtype=torch.bfloat16, seed=4)
    out = res.clone()
    U.gemm(A, W, out, M=M, N=N, K=K, lda=K, ldw=K, ldc=N, bias=bias, colscale=cs, res=out, ldres=N, cfg=cfg, ksplit=ksplit)
    y = (A.float() @ W[:N].float().T + bias.float()).bfloat16()
    y = ((y.float() * cs.float()).bfloat16().float() + res.float()).bfloat16()
    U.bf16_close(out, y, ulps=2.0, atol=2e-3)
    # SwiGLU
    F = 192
    w1, w3 = rnd(F, K, dtype=torch.bfloat16, seed=5, scale=0.1), rnd(F, K, dtype=torch.bfloat16, seed=6, scale=0.1)
    Wp = U.pack_swiglu(w1, w3)
    o2 = torch.zeros((M, F), dtype=torch.bfloat16, device=DEV)
    U.gemm(A, Wp, o2, M=M, N=2 * F, K=K, lda=K, ldw=K, ldc=F, swiglu=1, Npad=Wp.shape[0], cfg=cfg, ksplit=ksplit)
    a = (A.float() @ w1.float().T).bfloat16()
    b = (A.float() @ w3.float().T).bfloat16()
    U.bf16_close(o2, (torch.nn.functional.silu(a.float()).bfloat16().float() * b.float()).bfloat16(), ulps=2.0, atol=2e-3)
    # fp32 dilated conv through the taps loop
    T_, Ci, Co, k, dil = 333, 64, 96, 7, 3
    x = rnd(1, Ci, T_)
    w = rnd(Co, Ci, k, seed=7, scale=0.1)
    ref = torch.nn.functional.conv1d(torch.nn.functional.pad(x.double(), ((k - 1) * dil, 0)), w.double(), dilation=dil)[0].T
    xcl = torch.zeros((64 + T_, Ci), device=DEV)
    xcl[64:] = x[0].T
    Wg = U.pad_rows(w.permute(0, 2, 1).reshape(Co, k * Ci).contiguous())
    o3 = torch.zeros((T_, Co), device=DEV)
    U.gemm(xcl, Wg, o3, M=T_, N=Co, K=Ci, lda=Ci, ldw=k * Ci, ldc=Co, taps=k, tap_base=-(k - 1) * dil, tap_shift=dil,
           a_offset_elems=64 * Ci, cfg=cfg, ksplit=ksplit)
    assert (o3.double() - ref).abs().max().item() < 1e-4


@pytest.mark.parametrize("ksplit", [1, 2, 3, 4])
def test_gemm_pingpong_matches_torch(ksplit):
    """gemm_pp_kernel (cfg 5: 256x256 tile, 16x16x32 MFMA, staggered wave groups): ragged M / N, short and long K,
    bf16 epilogue, SwiGLU, and a dilated bf16 conv through the taps loop, against fp32 torch + the same rounding points."""
    for (M, N, K) in ((300, 384, 512), (1000, 640, 2048), (256, 256, 64 * ksplit)):
        A, W = rnd(M, K, dtype=torch.bfloat16), U.pad_rows(rnd(N, K, dtype=torch.bfloat16, seed=1))
        bias, cs = rnd(N, dtype=torch.bfloat16, seed=2), rnd(N, dtype=torch.bfloat16, seed=3)
        res = rnd(M, N, dtype=torch.bfloat16, seed=4)
        out = res.clone()
        U.gemm(A, W, out, M=M, N=N, K=K, lda=K, ldw=K, ldc=N, bias=bias, colscale=cs, res=out, ldres=N, cfg=5, ksplit=ksplit)
        y1 = (A.float() @ W[:N].float().T + bias.float()).bfloat16()
        y = ((y1.float() * cs.float()).bfloat16().float() + res.float()).bfloat16()
        # a different summation order may flip the bf16 rounding of the intermediate y1: allow 2 ulp of every rounding point
        tol = 2.0 * 2.0 ** -7 * ((y1.float() * cs.float()).abs() + y.float().abs()) + 2e-3
        err = (out.float() - y.float()).abs()
        assert bool((err <= tol).all()), (M, N, K, float((err - tol).max()))
    M, K, F = 300, 512, 192
    A = rnd(M, K, dtype=torch.bfloat16)
    w1, w3 = rnd(F, K, dtype=torch.bfloat16, seed=5, scale=0.1), rnd(F, K, dtype=torch.bfloat16, seed=6, scale=0.1)
    Wp = U.pack_swiglu(w1, w3)
    o2 = torch.zeros((M, F), dtype=torch.bfloat16, device=DEV)
    U.gemm(A, Wp, o2, M=M, N=2 * F, K=K, lda=K, ldw=K, ldc=F, swiglu=1, Npad=Wp.shape[0], cfg=5, ksplit=ksplit)
    a = (A.float() @ w1.float().T).bfloat16()
    b = (A.float() @ w3.float().T).bfloat16()
    U.bf16_close(o2, (torch.nn.functional.silu(a.float()).bfloat16().float() * b.float()).bfloat16(), ulps=2.0, atol=2e-3)
    # bf16 dilated causal conv through the taps loop (tap boundaries move the A cursor)
    T_, Ci, Co, k, dil = 333, 128, 96, 7, 3
    x = rnd(1, Ci, T_, dtype=torch.bfloat16)
    w = rnd(Co, Ci, k, seed=7, scale=0.1, dtype=torch.bfloat16)
    ref = torch.nn.functional.conv1d(torch.nn.functional.pad(x.double(), ((k - 1) * dil, 0)), w.double(), dilation=dil)[0].T
    xcl = torch.zeros((64 + T_, Ci), device=DEV, dtype=torch.bfloat16)
    xcl[64:] = x[0].T
    Wg = U.pad_rows(w.permute(0, 2, 1).reshape(Co, k * Ci).contiguous())
    o3 = torch.zeros((T_, Co), device=DEV, dtype=torch.bfloat16)
    U.gemm(xcl, Wg, o3, M=T_, N=Co, K=Ci, lda=Ci, ldw=k * Ci, ldc=Co, taps=k, tap_base=-(k - 1) * dil, tap_shift=dil,
           a_offset_elems=64 * Ci, cfg=5, ksplit=ksplit)
    U.bf16_close(o3, ref.float().bfloat16(), ulps=2.0, atol=2e-3)


def test_gemm_pingpong_deterministic_and_exact_on_integers():
    """Small-integer operands make every product and sum exact in fp32: the ping-pong kernel must reproduce the integer
    result bit for bit at a sampler-sized shape, three launches in a row (guards the LDS ring synchronisation)."""
    M, N, K = 1920, 2048, 2048
    g = torch.Generator(device="cpu").manual_seed(11)
    A = torch.randint(-3, 4, (M, K), generator=g).to(DEV, torch.bfloat16)
    W = torch.randint(-3, 4, (N, K), generator=g).to(DEV, torch.bfloat16)
    ref = (A.float() @ W.float().T)
    outs = []
    for _ in range(3):
        out = torch.zeros((M, N), dtype=torch.float32, device=DEV).bfloat16()
        U.gemm(A, W, out, M=M, N=N, K=K, lda=K, ldw=K, ldc=N, cfg=5)
        outs.append(out)
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    assert torch.equal(outs[0], ref.bfloat16())


def _int_operands(M, N, K, seed, wscale=1.0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    A = torch.randint(-3, 4, (M, K), generator=g).to(DEV, torch.bfloat16)
    W = (torch.randint(-3, 4, (N, K), generator=g).float() * wscale).to(DEV, torch.bfloat16)
    return A, W


@pytest.mark.parametrize("M,N,K", [(7680, 8192, 2048), (5120, 2048, 5888), (15360, 2048, 2048), (5000, 2304, 2048)])
def test_gemm_pingpong_walks_several_tiles_per_workgroup_exactly(M, N, K):
    """The regime bench.py runs the dominant kernel in: more 256x256 tiles than CUs (up to 7.5 per workgroup), so every
    workgroup goes through the tile hand-over (new per-lane offsets mid-stream, cursor reset, ring parity carried across
    tiles).  Integer operands make every fp32 sum exact: the bf16 output must equal torch's bit for bit, with a column
    scale + residual tail and with a ragged last row / column tile, twice in a row."""
    A, W = _int_operands(M, N, K, seed=21)
    Wp = U.pad_rows(W, 256)
    ref = A.float() @ W.float().T
    out = torch.zeros((M, N), dtype=torch.bfloat16, device=DEV)
    U.gemm(A, Wp, out, M=M, N=N, K=K, lda=K, ldw=K, ldc=N, cfg=5)
    assert torch.equal(out, ref.bfloat16())
    cs = (torch.randint(-2, 3, (N,), generator=torch.Generator().manual_seed(5)).float() * 0.5).to(DEV, torch.bfloat16)
    res = torch.randint(-8, 9, (M, N), generator=torch.Generator().manual_seed(6)).to(DEV, torch.bfloat16)
    want = ((ref.bfloat16().float() * cs.float()).bfloat16().float() + res.float()).bfloat16()
    for _ in range(2):
        o2 = res.clone()
        U.gemm(A, Wp, o2, M=M, N=N, K=K, lda=K, ldw=K, ldc=N, colscale=cs, res=o2, ldres=N, cfg=5)
        assert torch.equal(o2, want)


def test_gemm_pingpong_multi_tile_swiglu_and_fast_silu_rate():
    """SwiGLU tail at a sampler-sized multi-tile shape (M = 5120, w1|w3 of 5888 columns: 460 tiles on 256 CUs).  w1 / w3 are
    integers x 2^-6, so a = x.w1 and b = x.w3 are exact in fp32 and rounded identically by kernel and reference; what is
    left is silu: the kernel uses v_exp_f32 + v_rcp_f32 (common.h silu_fast) where torch uses expf + IEEE division.  A silu value
    that lands on the other side of a bf16 rounding boundary moves the rounded product by up to two ulp.  Stated rate: every
    output within two bf16 ulp, at least 99.9 % bit-identical."""
    M, K, F = 5120, 2048, 5888
    A, _ = _int_operands(M, 8, K, seed=31)
    g = torch.Generator(device="cpu").manual_seed(32)
    w1 = (torch.randint(-3, 4, (F, K), generator=g).float() * 2.0 ** -6).to(DEV, torch.bfloat16)
    w3 = (torch.randint(-3, 4, (F, K), generator=g).float() * 2.0 ** -6).to(DEV, torch.bfloat16)
    Wp = U.pack_swiglu(w1, w3)
    out = torch.zeros((M, F), dtype=torch.bfloat16, device=DEV)
    U.gemm(A, Wp, out, M=M, N=2 * F, K=K, lda=K, ldw=K, ldc=F, swiglu=1, Npad=Wp.shape[0], cfg=5)
    a = (A.float() @ w1.float().T).bfloat16()
    b = (A.float() @ w3.float().T).bfloat16()
    ref = (torch.nn.functional.silu(a.float()).bfloat16().float() * b.float()).bfloat16()
    U.bf16_close(out, ref, ulps=2.0, atol=1e-30, frac_exact=0.999)
    print(f"fast-silu SwiGLU tail: {float((out == ref).float().mean()):.6f} of {out.numel()} outputs bit-identical to expf + division")


def _qkv_reference(A, W, qk_w, rope, D, S, rope_heads, pos0, eps):
    """model.py:217-232 with the rounding points of SURVEY.md §A.2: bf16 linear outputs, fp32 per-head RMSNorm x weight -> bf16,
    fp32 interleaved-pair RoPE on heads < rope_heads -> bf16; V and gate are the rounded projections."""
    y = (A.float() @ W.float().T).bfloat16().float()                  # (M, 4D)
    M = y.shape[0]
    H = D // 128
    outs = []
    for sec in range(2):
        t = y[:, sec * D:(sec + 1) * D].view(M, H, 128)
        n = (t * torch.rsqrt(t.pow(2).mean(-1, keepdim=True) + eps)) * qk_w[sec * D:(sec + 1) * D].float().view(1, H, 128)
        n = n.bfloat16().float()
        pos = pos0 + (torch.arange(M, device=A.device) % S)
        cs = rope[pos]                                                # (M, 64, 2)
        x0, x1 = n[..., 0::2], n[..., 1::2]
        c, s_ = cs[:, None, :, 0], cs[:, None, :, 1]
        r = torch.stack([x0 * c - x1 * s_, x0 * s_ + x1 * c], dim=-1).reshape(M, H, 128)
        n = torch.where((torch.arange(H, device=A.device) < rope_heads).view(1, H, 1), r, n)
        outs.append(n.reshape(M, D).bfloat16())
    return outs[0], outs[1], y[:, 2 * D:3 * D].bfloat16(), y[:, 3 * D:].bfloat16()


@pytest.mark.parametrize("rows,S", [(12, 640), (5, 333)])
def test_gemm_pingpong_fused_qkv_tail_multi_tile(rows, S):
    """The fused QKV(G) tail of the ping-pong kernel (per-head RMSNorm + half-head RoPE on q / k, transposed V, gate) at a
    sampler shape with several tiles per workgroup (rows x S tokens x 8192 columns; S = 333 makes token tiles straddle rows
    and leaves a ragged last tile), against a torch evaluation with the reference's rounding points.  Integer operands make
    the projection exact, so only rsqrt / the norm product can differ: >= 99 % bit-identical, and within 2 bf16 ulp of the value
    or of the rotation's inputs (a RoPE output is a difference of two products: a one-ulp flip of a normalised input of magnitude
    ~4 moves a small output by that input's ulp, 2^-5)."""
    from echo_tts_amd.model import rope_table
    D, K, H = 2048, 2048, 16
    M = rows * S
    A, W = _int_operands(M, 4 * D, K, seed=41, wscale=2.0 ** -5)
    qk_w = (1.0 + 0.25 * torch.randn((2 * D,), generator=torch.Generator().manual_seed(42))).to(DEV, torch.bfloat16)
    rope = rope_table(128, 1024).to(DEV)
    Sp = (S + 63) // 64 * 64
    vt = torch.zeros((rows, D, Sp), dtype=torch.bfloat16, device=DEV)
    out = torch.zeros((M, 4 * D), dtype=torch.bfloat16, device=DEV)
    U.gemm(A, W, out, M=M, N=4 * D, K=K, lda=K, ldw=K, ldc=4 * D, cfg=5,
           qkv=dict(D=D, S=S, rope_heads=H // 2, pos0=3, eps=1e-5, qk_w=qk_w, rope=rope, vt=vt, vt_ld=Sp, vt_row_stride=D * Sp))
    q, k, v, gte = _qkv_reference(A, W, qk_w, rope, D, S, H // 2, 3, 1e-5)
    U.bf16_close(out[:, :D], q, ulps=2.0, atol=2.0 ** -4, frac_exact=0.99)
    U.bf16_close(out[:, D:2 * D], k, ulps=2.0, atol=2.0 ** -4, frac_exact=0.99)
    assert torch.equal(out[:, 3 * D:], gte)
    assert torch.equal(vt[:, :, :S], v.view(rows, S, D).transpose(1, 2))
    # ABI 6, qkv_gate_act: the gate section leaves as bf16(sigmoid(bf16(acc))) (the attention epilogue's v_exp + v_rcp form, moved into
    # this tail), everything else bit for bit as before - the ping-pong kernel and the tile kernel's tail (cfg 4)
    for cfg in (5, 4):
        vt2 = torch.zeros_like(vt)
        out2 = torch.zeros_like(out)
        U.gemm(A, W, out2, M=M, N=4 * D, K=K, lda=K, ldw=K, ldc=4 * D, cfg=cfg,
               qkv=dict(D=D, S=S, rope_heads=H // 2, pos0=3, eps=1e-5, qk_w=qk_w, rope=rope, vt=vt2, vt_ld=Sp, vt_row_stride=D * Sp, gate_act=1))
        if cfg == 5:
            assert torch.equal(out2[:, :3 * D], out[:, :3 * D]) and torch.equal(vt2, vt)
        U.bf16_close(out2[:, 3 * D:], torch.sigmoid(gte.float()).bfloat16(), ulps=1.0, frac_exact=0.995)


def test_gemm_pingpong_tail_specialised_instantiations_equal_the_all_tails_build(tmp_path):
    """The tail-specialised, branch-free instantiations of gemm_pp_kernel (plain / column scale + residual / fused QKV, interior and edge forms,
    bf16 and fp8) against the one-kernel-with-every-tail instantiation (ECHO_PP_TAILS=0, read once per process: two child processes) on the
    same seeded random operands: every output bit for bit - an interior-only shape and a ragged one (M = 5000, S = 200), in-place residual."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "tails.py"
    script.write_text(
        "import sys, torch\n"
        f"sys.path.insert(0, {root!r})\n"
        "from tests import gpu_util as U\n"
        "g = torch.Generator().manual_seed(5)\n"
        "out = {}\n"
        "D, K = 2048, 2048\n"
        "for (B, S) in ((4, 640), (25, 200)):\n"
        "    M = B * S\n"
        "    A = (torch.rand((M + 256, K), generator=g) * 2 - 1).to(torch.bfloat16).cuda()\n"
        "    W = ((torch.rand((4 * D, K), generator=g) * 2 - 1) * 0.05).to(torch.bfloat16).cuda()\n"
        "    cs = torch.rand((D,), generator=g).to(torch.bfloat16).cuda()\n"
        "    res = torch.randn((M, D), generator=g).to(torch.bfloat16).cuda()\n"
        "    qk_w = (1 + 0.1 * torch.randn((2 * D,), generator=g)).to(torch.bfloat16).cuda()\n"
        "    ang = torch.rand((S, 64), generator=g).cuda(); rope = torch.stack([torch.cos(ang), torch.sin(ang)], -1).contiguous()\n"
        "    A8, sa = U.quant_rows_fp8(A); W8, sw = U.quant_rows_fp8(W)\n"
        "    for name, a, w, extra in (('bf16', A, W, {}), ('fp8', A8, W8, dict(a_scale=sa, w_scale=sw))):\n"
        "        c = torch.zeros((M, 4 * D), dtype=torch.bfloat16, device='cuda')\n"
        "        U.gemm(a, w, c, M=M, N=4 * D, K=K, lda=K, ldw=K, ldc=4 * D, Npad=4 * D, cfg=5, **extra)\n"
        "        out[f'plain_{name}_{M}'] = c.cpu()\n"
        "        r = res.clone()\n"
        "        U.gemm(a, w, r, M=M, N=D, K=K, lda=K, ldw=K, ldc=D, Npad=D, cfg=5, colscale=cs, res=r, ldres=D, **extra)\n"
        "        out[f'csres_{name}_{M}'] = r.cpu()\n"
        "        c = torch.zeros((M, 4 * D), dtype=torch.bfloat16, device='cuda'); vt = torch.zeros((B, D, S), dtype=torch.bfloat16, device='cuda')\n"
        "        U.gemm(a, w, c, M=M, N=4 * D, K=K, lda=K, ldw=K, ldc=4 * D, Npad=4 * D, cfg=5, **extra,\n"
        "               qkv=dict(D=D, S=S, rope_heads=8, pos0=0, eps=1e-6, qk_w=qk_w, rope=rope, vt=vt, vt_ld=S, vt_row_stride=D * S, gate_act=1))\n"
        "        out[f'qkv_{name}_{M}'] = c.cpu(); out[f'vt_{name}_{M}'] = vt.cpu()\n"
        "torch.cuda.synchronize(); torch.save(out, sys.argv[1])\n")
    outs = {}
    for tails in ("1", "0"):
        o = tmp_path / f"tails{tails}.pt"
        r = subprocess.run([sys.executable, str(script), str(o)], env=dict(os.environ, ECHO_PP_TAILS=tails), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs[tails] = torch.load(o, weights_only=True)
    assert outs["1"].keys() == outs["0"].keys() and len(outs["1"]) == 16
    for k in outs["1"]:
        assert torch.equal(outs["1"][k], outs["0"][k]), k


@pytest.mark.parametrize("B,S", [(24, 640), (5, 200), (7, 100), (3, 37)])
def test_gemm_pingpong_fused_qkv_tail_is_equivariant_under_row_block_permutation(B, S):
    """A token's q / k / v / gate must not depend on WHERE in the launch it sits: the same B sequences in another order give the same bits,
    with random (non-integer) operands, RoPE on half the heads, and sequence lengths that leave a ragged last tile and make 16-row blocks
    straddle sequences.  Round 3 finding: the interior-tile and the edge-tile copy of the fused tail are compiled separately, and with hipcc's
    default floating-point contraction each copy fused a * c - b * s into an fma in its own way - 1-4 RoPE outputs per launch came out one
    bf16 ulp apart depending on the tile a token was in (the library is built with -ffp-contract=off since)."""
    D, K = 2048, 2048
    M, N = B * S, 4 * D
    g = torch.Generator().manual_seed(3)
    A = (torch.rand((M + 256, K), generator=g) * 2 - 1).to(torch.bfloat16).to(DEV)
    W = ((torch.rand((N, K), generator=g) * 2 - 1) * 0.05).to(torch.bfloat16).to(DEV)
    qk_w = (1 + 0.1 * torch.randn((2 * D,), generator=g)).to(torch.bfloat16).to(DEV)
    ang = torch.rand((S, 64), generator=g).to(DEV)
    rope = torch.stack([torch.cos(ang), torch.sin(ang)], -1).contiguous()
    S8 = (S + 7) // 8 * 8

    def go(a):
        out = torch.zeros((M, N), dtype=torch.bfloat16, device=DEV)
        vt = torch.zeros((B, D, S8), dtype=torch.bfloat16, device=DEV)
        U.gemm(a, W, out, M=M, N=N, K=K, lda=K, ldw=K, ldc=N, Npad=N, cfg=5,
               qkv=dict(D=D, S=S, rope_heads=8, pos0=0, eps=1e-6, qk_w=qk_w, rope=rope, vt=vt, vt_ld=S8, vt_row_stride=D * S8, gate_act=1))
        return out, vt

    out0, vt0 = go(A)
    perm = torch.randperm(B, generator=g)
    Ap = A.clone()
    Ap[:M] = A[:M].view(B, S, K)[perm.to(DEV)].reshape(M, K)
    out1, vt1 = go(Ap)
    assert torch.equal(out1, out0.view(B, S, N)[perm.to(DEV)].reshape(M, N))
    assert torch.equal(vt1[:, :, :S], vt0[perm.to(DEV)][:, :, :S])


@pytest.mark.parametrize("cfg", [5, 4, 2, 0, 104, 106, 107])
@pytest.mark.parametrize("M,N,K", [(4096, 4096, 4096), (2560, 2048, 2048), (1000, 640, 2048)])
def test_gemm_operands_flush_against_the_end_of_their_allocation(cfg, M, N, K):
    """Round-1 finding: one run of a 5,104,4 sweep faulted at 4096^3 on a 2 MiB-aligned address, i.e. just past an operand whose
    size is a 2 MiB multiple (A, W, C are 32 MiB there), and a later run passed.  Here A, W, the residual and C each live in
    their own hipMalloc and END exactly at its last byte, so the last look-ahead units of the LDS-DMA ring (which re-stage the
    final K-tile once the cursors stop), the clamped rows of ragged tiles and the epilogue's 16-byte accesses fault
    deterministically if they step over.  cfg 104 / 106 / 107: the diagnostic builds of that sweep and the LEAD 6 / 4 variants
    (results not checked: timing builds).  Runs once; a fault here is a bug to read, not to retry."""
    if cfg >= 100 and (N & 7):
        pytest.skip("diagnostic builds share the ping-pong kernel's alignment rules")
    A, W = _int_operands(M, N, K, seed=51)
    res = torch.randint(-8, 9, (M, N), generator=torch.Generator().manual_seed(7)).to(DEV, torch.bfloat16)
    fa, fw = U.FlushAlloc(A), U.FlushAlloc(U.pad_rows(W, 128))
    fc = U.FlushAlloc(res)
    try:
        U.gemm(fa, fw, fc, M=M, N=N, K=K, lda=K, ldw=K, ldc=N, res=fc, ldres=N, cfg=cfg)
        torch.cuda.synchronize()
        if cfg < 100:
            want = ((A.float() @ W.float().T).bfloat16().float() + res.float()).bfloat16()
            assert torch.equal(fc.to_tensor(), want)
    finally:
        fa.free(); fw.free(); fc.free()


def test_gemm_fp8_operands_flush_against_the_end_of_their_allocation():
    """Same guard for the e4m3 variant of the ping-pong kernel (1-byte elements: the K-tile is 128 elements) and its row scales."""
    M, N, K = 2560, 2048, 2048
    a, w = rnd(M, K, dtype=torch.bfloat16), rnd(N, K, dtype=torch.bfloat16, seed=1)
    qa, sa = U.quant_rows_fp8(a)
    qw, sw = U.quant_rows_fp8(w)
    fa, fw, fsa, fsw = U.FlushAlloc(qa), U.FlushAlloc(qw), U.FlushAlloc(sa), U.FlushAlloc(sw)
    fc = U.FlushAlloc(torch.zeros((M, N), dtype=torch.bfloat16, device=DEV))
    try:
        plain = torch.zeros((M, N), dtype=torch.bfloat16, device=DEV)
        U.gemm(qa, qw, plain, M=M, N=N, K=K, lda=K, ldw=K, ldc=N, cfg=5, a_scale=sa, w_scale=sw)
        U.gemm(fa, fw, fc, M=M, N=N, K=K, lda=K, ldw=K, ldc=N, cfg=5, a_scale=fsa, w_scale=fsw)
        torch.cuda.synchronize()
        assert torch.equal(fc.to_tensor(), plain)
    finally:
        for f in (fa, fw, fsa, fsw, fc):
            f.free()


@pytest.mark.parametrize("cfg", [0, 2, 4, 6, 7, 8, 9])
def test_gemm_f32_split3_accuracy(cfg):
    """fp32 GEMM on 3 bf16 MFMAs per product: relative error ~1e-5 of the exact result (used by the DAC decoder)."""
    M, N, K = 300, 256, 1024
    A = rnd(M, K)
    W = U.pad_rows(rnd(N, K, seed=1))
    out = torch.zeros((M, N), device=DEV)
    U.gemm(A, W, out, M=M, N=N, K=K, lda=K, ldw=K, ldc=N, cfg=cfg, split3=1)
    ref = A.double() @ W[:N].double().T
    rel = ((out.double() - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item()
    assert rel < 3e-5, rel


@pytest.mark.parametrize("cfg,taps", [(0, 1), (2, 7), (4, 1), (6, 7), (7, 7), (8, 7), (9, 1), (8, 2)])
def test_gemm_f32_split3_presplit_weights_bit_identical(cfg, taps):
    """echo_op_presplit_weights + w_presplit: the weight operand of a split3 GEMM reformatted once ([32 hi | 32 lo] bf16 per 32-float
    block) instead of being split per fragment by every wave.  Same hi / lo values, same MFMAs: bit-identical results, with taps
    (the DAC decoder's causal dilated convs), ragged M, and N below the tile width."""
    M, N, K = 333, 192, 96 * 2
    rows = M + 8 * taps
    A = rnd(rows, K)
    W = U.pad_rows(rnd(N, taps * K, seed=1))
    kw = dict(M=M, N=N, K=K, lda=K, ldw=taps * K, ldc=N, cfg=cfg, split3=1, taps=taps, tap_base=-(taps - 1), tap_shift=1, a_offset_elems=(taps - 1) * K)
    ref = torch.zeros((M, N), device=DEV)
    U.gemm(A, W, ref, **kw)
    Wp = W.clone()
    L.check(U.lib().echo_op_presplit_weights(Wp.data_ptr(), Wp.shape[0], Wp.shape[1], U.stream()))
    out = torch.zeros((M, N), device=DEV)
    U.gemm(A, Wp, out, w_presplit=1, **kw)
    torch.cuda.synchronize()
    assert not torch.equal(Wp, W)
    assert torch.equal(out, ref)
    # and it is the fp32 product to split3 accuracy
    acc = torch.zeros((M, N), dtype=torch.float64, device=DEV)
    for t in range(taps):
        acc += A[t:t + M].double() @ W[:N, t * K:(t + 1) * K].double().T
    rel = ((out.double() - acc).pow(2).mean().sqrt() / acc.pow(2).mean().sqrt()).item()
    assert rel < 3e-5, rel


@pytest.mark.parametrize("R", [3, 1])
def test_attention_is_deterministic_at_full_size(R):
    """Same launch three times at the C2 shape (S=640, 16 heads, 436 text + 640 speaker keys): bit-identical and finite.
    Guards the LDS-DMA ring synchronisation (a missing vmcnt wait before the tile barrier showed up only at this size)."""
    S, H, Lt, Ls = 640, 16, 436, 640
    D = H * 128
    qkvg = rnd(R * S + 256, 4 * D, dtype=torch.bfloat16, scale=0.5)
    pS, pT, pSp = (S + 63) // 64 * 64, (Lt + 63) // 64 * 64, (Ls + 63) // 64 * 64
    vt_self = rnd(R, H, 128, pS, dtype=torch.bfloat16, seed=1)
    kt, vt_t = rnd(Lt + 128, 4 * D, dtype=torch.bfloat16, seed=2), rnd(1, H, 128, pT, dtype=torch.bfloat16, seed=3)
    ksp, vt_s = rnd(Ls + 128, 4 * D, dtype=torch.bfloat16, seed=4), rnd(1, H, 128, pSp, dtype=torch.bfloat16, seed=5)
    rows = [[S] * R, [Lt, 0, Lt][:R], [Ls, Ls, 0][:R]]
    nk = torch.tensor(rows, dtype=torch.int32, device=DEV)
    outs = []
    for _ in range(3):
        out = torch.zeros((R * S, D), dtype=torch.bfloat16, device=DEV)
        d = L.EchoAttnDesc()
        d.Q, d.q_ld, d.q_row_stride = qkvg.data_ptr(), 4 * D, S * 4 * D
        d.O, d.o_ld, d.o_row_stride = out.data_ptr(), D, S * D
        d.G, d.g_ld, d.g_row_stride = qkvg.data_ptr() + 3 * D * 2, 4 * D, S * 4 * D
        d.S, d.H, d.rows, d.nseg, d.causal, d.scale = S, H, R, 3, 0, 1 / math.sqrt(128)
        for i, (kp, kld, krs, vt, pitch, shared) in enumerate(((qkvg.data_ptr() + D * 2, 4 * D, S * 4 * D, vt_self, pS, False),
                                                               (kt.data_ptr(), 4 * D, 0, vt_t, pT, True),
                                                               (ksp.data_ptr(), 4 * D, 0, vt_s, pSp, True))):
            sg = d.seg[i]
            sg.K, sg.k_ld, sg.k_head_stride, sg.k_row_stride = kp, kld, 128, krs
            sg.Vt, sg.vt_ld, sg.vt_head_stride = vt.data_ptr(), pitch, 128 * pitch
            sg.vt_row_stride = 0 if shared else H * 128 * pitch
            sg.nkeys = nk[i].data_ptr()
            sg.kv_mod = 1 if shared else 0
        L.check(U.lib().echo_op_attention_bf16(C.byref(d), U.stream()))
        torch.cuda.synchronize()
        outs.append(out.clone())
    assert bool(torch.isfinite(outs[0].float()).all())
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    # and it is right: fp32 reference of row 0, head 0
    q = qkvg[:S, :128].float()
    k = torch.cat([qkvg[:S, D:D + 128], kt[:Lt, :128], ksp[:Ls, :128]]).float()
    v = torch.cat([vt_self[0, 0, :, :S].T, vt_t[0, 0, :, :Lt].T, vt_s[0, 0, :, :Ls].T]).float()
    ref = torch.softmax(q @ k.T / math.sqrt(128), -1) @ v
    ref = (ref.bfloat16().float() * torch.sigmoid(qkvg[:S, 3 * D:3 * D + 128].float()).bfloat16().float())
    err = (outs[0][:S, :128].float() - ref).abs()
    assert float(err.max()) < 3e-2 and float(err.mean()) < 2e-3


@pytest.mark.parametrize("Lt,Ls", [(436, 640), (372, 640), (308, 640), (500, 600), (64, 0), (1, 0), (0, 0)])
def test_attention_tile_count_classes(Lt, Ls):
    """Every row, head and query against fp32 at S = 640 for key-tile counts in every remainder class of the kernels' unrolled loops
    (27 / 20 / 17, 26 / 20 / 16, 25 / 20 / 15, 28 / 20 / 18, 11 / 10, 10 tiles per row) incl. ragged last tiles and one-key segments;
    three launches bit-identical (a value placed in the destination of an in-flight LDS read showed up only at total_tiles % 3 == 2)."""
    R, S, H = 3, 640, 4
    D = H * 128
    qkvg = rnd(R * S + 256, 4 * D, dtype=torch.bfloat16, scale=0.5)
    pS, pT, pSp = (S + 63) // 64 * 64, max(64, (Lt + 63) // 64 * 64), max(64, (Ls + 63) // 64 * 64)
    vt_self = rnd(R, H, 128, pS, dtype=torch.bfloat16, seed=1)
    kt, vt_t = rnd(Lt + 128, 4 * D, dtype=torch.bfloat16, seed=2), rnd(1, H, 128, pT, dtype=torch.bfloat16, seed=3)
    ksp, vt_s = rnd(Ls + 128, 4 * D, dtype=torch.bfloat16, seed=4), rnd(1, H, 128, pSp, dtype=torch.bfloat16, seed=5)
    rows = [[S] * R, [Lt, 0, Lt], [Ls, Ls, 0]]
    nk = torch.tensor(rows, dtype=torch.int32, device=DEV)
    outs = []
    for _ in range(3):
        out = torch.zeros((R * S, D), dtype=torch.bfloat16, device=DEV)
        d = L.EchoAttnDesc()
        d.Q, d.q_ld, d.q_row_stride = qkvg.data_ptr(), 4 * D, S * 4 * D
        d.O, d.o_ld, d.o_row_stride = out.data_ptr(), D, S * D
        d.G, d.g_ld, d.g_row_stride = qkvg.data_ptr() + 3 * D * 2, 4 * D, S * 4 * D
        d.S, d.H, d.rows, d.nseg, d.causal, d.scale = S, H, R, 3, 0, 1 / math.sqrt(128)
        for i, (kp, kld, krs, vt, pitch, shared) in enumerate(((qkvg.data_ptr() + D * 2, 4 * D, S * 4 * D, vt_self, pS, False),
                                                               (kt.data_ptr(), 4 * D, 0, vt_t, pT, True),
                                                               (ksp.data_ptr(), 4 * D, 0, vt_s, pSp, True))):
            sg = d.seg[i]
            sg.K, sg.k_ld, sg.k_head_stride, sg.k_row_stride = kp, kld, 128, krs
            sg.Vt, sg.vt_ld, sg.vt_head_stride = vt.data_ptr(), pitch, 128 * pitch
            sg.vt_row_stride = 0 if shared else H * 128 * pitch
            sg.nkeys = nk[i].data_ptr()
            sg.kv_mod = 1 if shared else 0
        L.check(U.lib().echo_op_attention_bf16(C.byref(d), U.stream()))
        torch.cuda.synchronize()
        outs.append(out.clone())
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    for r in range(R):
        q = qkvg[r * S:(r + 1) * S, :D].float().view(S, H, 128).transpose(0, 1)
        ks = [qkvg[r * S:(r + 1) * S, D:2 * D].float().view(S, H, 128).transpose(0, 1)]
        vs = [vt_self[r, :, :, :S].float().transpose(1, 2)]
        if rows[1][r]:
            ks.append(kt[:Lt, :D].float().view(Lt, H, 128).transpose(0, 1)); vs.append(vt_t[0, :, :, :Lt].float().transpose(1, 2))
        if rows[2][r]:
            ks.append(ksp[:Ls, :D].float().view(Ls, H, 128).transpose(0, 1)); vs.append(vt_s[0, :, :, :Ls].float().transpose(1, 2))
        k, v = torch.cat(ks, 1), torch.cat(vs, 1)
        ref = torch.softmax(q @ k.transpose(1, 2) / math.sqrt(128), -1) @ v
        gate = torch.sigmoid(qkvg[r * S:(r + 1) * S, 3 * D:].float()).bfloat16().float().view(S, H, 128).transpose(0, 1)
        ref = ref.bfloat16().float() * gate
        got = outs[0][r * S:(r + 1) * S].float().view(S, H, 128).transpose(0, 1)
        err = (got - ref).abs()
        assert float(err.max()) < 3e-2 and float(err.mean()) < 2e-3, (r, float(err.max()), float(err.mean()))


def test_plain_c_caller_runs_through_the_abi(tmp_path):
    """tests/c_abi/abi_smoke.c: a C11 program (no Python, no torch) drives echo_op_gemm through include/echo_hip.h; exact
    integer GEMM, refused bad descriptor, error text.  One short-lived child process on the GPU."""
    import subprocess
    from tests.test_host_cpu import _build_c_caller
    exe = _build_c_caller(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    assert "abi_smoke: ok" in r.stdout


def _deq(q8: torch.Tensor) -> torch.Tensor:
    """uint8 e4m3 bytes -> fp32 values (decoded on the CPU: no dependence on the GPU build's float8 kernels)."""
    return q8.cpu().view(torch.float8_e4m3fn).float().to(q8.device)


def _to_e4m3_bytes(x: torch.Tensor) -> torch.Tensor:
    return x.float().cpu().to(torch.float8_e4m3fn).view(torch.uint8).to(x.device)


def _quant_ref(x: torch.Tensor):
    """The kernel's arithmetic in torch: scale = amax / 448, q = e4m3(x * (448 / amax)) (fp32 ops, round to nearest even)."""
    xf = x.float().cpu()                                  # IEEE fp32 division on the host = __fdiv_rn in the kernel
    amax = xf.abs().amax(dim=1, keepdim=True)
    nz = amax > 0
    inv = torch.where(nz, 448.0 / amax, torch.ones_like(amax))
    sc = torch.where(nz, amax / 448.0, torch.ones_like(amax))
    return _to_e4m3_bytes(xf * inv).to(x.device), sc[:, 0].to(x.device)


def test_quant_rows_fp8_matches_torch_e4m3():
    x = rnd(300, 2048, dtype=torch.bfloat16)
    x[7] = 0                                             # an all-zero row keeps scale 1
    x[11, 5] = 37.5                                      # an outlier sets its row's scale
    q, s = U.quant_rows_fp8(x)
    q_ref, s_ref = _quant_ref(x)
    assert torch.equal(s, s_ref)
    # v_cvt_pk_fp8_f32 resolves NEAR-ties to even (a product a hair above the midpoint of two e4m3 codes goes to the even
    # one), torch rounds on the full fp32 value: measured 0.05 % of the elements land on the neighbouring code.  Everything
    # else is bit-identical; -0 / +0 aside.
    d, dr = _deq(q), _deq(q_ref)
    bad = d != dr
    assert float(bad.float().mean()) < 2e-3, float(bad.float().mean())
    step = dr.abs().clamp_min(2.0 ** -6) * 0.126            # one e4m3 step is 1/8 of the power of two below |value|
    assert bool(((d - dr).abs()[bad] <= step[bad] * 1.0001).all())
    scaled = (x.float() * (448.0 / x.float().abs().amax(dim=1, keepdim=True).clamp_min(1e-30)))
    mid = 0.5 * (d + dr)
    assert bool(((scaled - mid).abs()[bad] <= 2.0 ** -9 * mid.abs()[bad]).all())   # only products at a rounding boundary differ


@pytest.mark.parametrize("M,N,K", [(300, 384, 512), (1000, 640, 2048), (2560, 2048, 5888)])
def test_gemm_pingpong_fp8_matches_dequantised_product(M, N, K):
    """gemm_pp_kernel<FP8>: C = bf16((A8 . W8^T) * a_scale[m] * w_scale[n]) against the fp32 product of the dequantised
    operands (every e4m3 x e4m3 product is exact in fp32; only the summation order differs)."""
    A, W = rnd(M, K, dtype=torch.bfloat16), U.pad_rows(rnd(N, K, dtype=torch.bfloat16, seed=1))
    A8, sa = U.quant_rows_fp8(A)
    W8, sw = U.quant_rows_fp8(W)
    out = torch.zeros((M, N), dtype=torch.bfloat16, device=DEV)
    U.gemm(A8, W8, out, M=M, N=N, K=K, lda=K, ldw=K, ldc=N, cfg=5, a_scale=sa, w_scale=sw)
    ref = (_deq(A8) @ _deq(W8)[:N].T) * sa[:, None] * sw[None, :N]
    # the block-scaled MFMA does not add its 128 products in an fp32 chain: deviations of up to ~1e-4 of the output rms were
    # measured on outputs that cancel to ~0 (invisible after the bf16 rounding of any output of ordinary size)
    U.bf16_close(out, ref.bfloat16(), ulps=2.0, atol=5e-4 * float(ref.pow(2).mean().sqrt()) + 2e-3)
    # and close to the unquantised bf16 product: two e4m3 roundings per product, ~3 % relative rms
    exact = A.float() @ W[:N].float().T
    rel = ((out.float() - exact).pow(2).mean().sqrt() / exact.pow(2).mean().sqrt()).item()
    assert rel < 0.06, rel


def test_gemm_pingpong_fp8_exact_on_integers_and_swiglu():
    """Small integers are exact in e4m3 and in the fp32 accumulator: bit-exact result at a sampler-sized shape (guards the
    fragment addressing of the 32-byte K runs and the LDS ring), plus the SwiGLU register tail on fp8 operands."""
    M, N, K = 1920, 2048, 2048
    g = torch.Generator(device="cpu").manual_seed(5)
    A = torch.randint(-3, 4, (M, K), generator=g).to(DEV, torch.float32)
    W = torch.randint(-3, 4, (N, K), generator=g).to(DEV, torch.float32)
    A8, W8 = _to_e4m3_bytes(A), _to_e4m3_bytes(W)
    ones_m, ones_n = torch.ones(M, device=DEV), torch.ones(N, device=DEV)
    out = torch.zeros((M, N), dtype=torch.bfloat16, device=DEV)
    U.gemm(A8, W8, out, M=M, N=N, K=K, lda=K, ldw=K, ldc=N, cfg=5, a_scale=ones_m, w_scale=ones_n)
    assert torch.equal(out, (A @ W.T).bfloat16())
    F, K2, M2 = 192, 512, 300
    a = rnd(M2, K2, dtype=torch.bfloat16)
    w1, w3 = rnd(F, K2, dtype=torch.bfloat16, seed=5, scale=0.1), rnd(F, K2, dtype=torch.bfloat16, seed=6, scale=0.1)
    Wp = U.pack_swiglu(w1, w3)
    a8, sa = U.quant_rows_fp8(a)
    w8, sw = U.quant_rows_fp8(Wp)
    o2 = torch.zeros((M2, F), dtype=torch.bfloat16, device=DEV)
    U.gemm(a8, w8, o2, M=M2, N=2 * F, K=K2, lda=K2, ldw=K2, ldc=F, swiglu=1, Npad=Wp.shape[0], cfg=5, a_scale=sa, w_scale=sw)
    full = (_deq(a8) @ _deq(w8).T) * sa[:, None] * sw[None, :]
    # undo the [16 x w1 | 16 x w3] row packing of pack_swiglu
    blk = full[:, :2 * F].reshape(M2, F // 16, 2, 16)
    ya, yb = blk[:, :, 0].reshape(M2, F).bfloat16(), blk[:, :, 1].reshape(M2, F).bfloat16()
    U.bf16_close(o2, (torch.nn.functional.silu(ya.float()).bfloat16().float() * yb.float()).bfloat16(), ulps=2.0, atol=2e-3)
    # ABI 6, static activation scales: a_scale == NULL + a_scale_const equals a constant scale vector bit for bit, and the SwiGLU
    # tail's e4m3 output (c8) is the quantisation of the bf16 output of the same launch (saturating)
    cs = float(sa.mean())
    o3, o4 = torch.zeros_like(o2), torch.zeros_like(o2)
    U.gemm(a8, w8, o3, M=M2, N=2 * F, K=K2, lda=K2, ldw=K2, ldc=F, swiglu=1, Npad=Wp.shape[0], cfg=5, a_scale=torch.full_like(sa, cs), w_scale=sw)
    U.gemm(a8, w8, o4, M=M2, N=2 * F, K=K2, lda=K2, ldw=K2, ldc=F, swiglu=1, Npad=Wp.shape[0], cfg=5, a_scale_const=cs, w_scale=sw)
    assert torch.equal(o3, o4) and bool(o3.any())
    inv = 448.0 / float(o3.float().abs().max()) * 1.5
    c8 = torch.full((M2, F + 8), 0x55, dtype=torch.uint8, device=DEV)
    o5 = torch.zeros_like(o2)
    U.gemm(a8, w8, o5, M=M2, N=2 * F, K=K2, lda=K2, ldw=K2, ldc=F, swiglu=1, Npad=Wp.shape[0], cfg=5, a_scale_const=cs, w_scale=sw, c8=c8, c8_inv=inv)
    assert not bool(o5.any()) and bool((c8[:, F:] == 0x55).all())
    got, ref = _deq(c8[:, :F].contiguous()), _deq(_to_e4m3_bytes((o3.float() * inv).clamp(-448.0, 448.0)))
    bad = got != ref
    assert float(bad.float().mean()) < 2e-3 and bool(((got - ref).abs()[bad] <= ref.abs().clamp_min(2.0 ** -6)[bad] * 0.1261).all())
    assert float((got.abs() == 448.0).float().mean()) > 0


def test_fp8_block_linears_teacher_forced_against_the_restatement():
    """BASELINE config C5, pinned PER LINEAR.  The reference has no fp8 path (parity with the reference: unpinned by nature); the
    arithmetic the fp8 engine states is restated in the oracle (`set_fp8_block_linears`: e4m3 operands, one scale per token row and
    per weight row = amax / 448, fp32 accumulation, bf16 result and bf16 tails).  End to end the two can only agree loosely, and not
    because of a disagreement about that arithmetic: re-quantising to e4m3 turns every legitimate bf16-level difference upstream (the
    fp32 summation order of a GEMM, flash attention rounding P to bf16 where SDPA keeps fp32) into 6-12 % steps of the affected
    codes - the restatement itself moves by 0.32 of fp8's whole effect when its SDPA is replaced by a flash-style evaluation (measured
    on the CPU, d = 2048; the same swap moves the plain bf16 forward 5x less).  So here every block linear of one full-width layer is
    TEACHER-FORCED: the oracle's own bf16 operand of that linear (tapped from its fp8 forward) goes through the engine's kernels -
    `norm_adaln_fp8` / `quant_rows_fp8`, then `gemm_pp_kernel<FP8>` with the tail the engine uses - and is compared with the oracle's
    own result for that single linear:
      * row scales bit-equal, e4m3 codes >= 99.8 % identical and never more than one code apart (the converter's near-tie rule);
      * the bf16 output at least 4x closer to the restatement's output of that linear than that output is to the plain bf16
        linear (fp8's own effect on this linear; the 0.05 % of operand codes that sit one step away because v_cvt_pk_fp8_f32
        resolves near-ties to even account for ~0.07 of it), and within 2 bf16 ulps of the fp64 product of the engine's own codes."""
    from oracle import echo_ref as R
    from tests.golden_defs import WIDE1
    cfg, S, T = WIDE1, 320, 40
    D, F = cfg.model_size, cfg.intermediate_size
    wb = {k: v.bfloat16() for k, v in R.make_dit_weights(cfg, seed=0).items()}
    gen = torch.Generator().manual_seed(3)
    ids = torch.randint(1, 256, (1, T), generator=gen, dtype=torch.int32)
    tmask = torch.ones((1, T), dtype=torch.bool)
    spk, smask = torch.randn((1, 64, 80), generator=gen).bfloat16(), torch.ones((1, 64), dtype=torch.bool)
    x = torch.randn((1, S, 80), generator=gen).bfloat16()
    t = torch.full((1,), 0.75).bfloat16()
    kvt, kvs = R.kv_cache_text(wb, cfg, ids, tmask), R.kv_cache_speaker(wb, cfg, spk)
    taps = {}
    R.set_linear_taps(taps)
    R.set_fp8_block_linears(True)
    try:
        R.dit_forward(wb, cfg, x, t, tmask, smask, kvt, kvs)
    finally:
        R.set_fp8_block_linears(False)
        R.set_linear_taps(None)
    rms = lambda a, b=None: float(((a.float().cpu() - b.float().cpu()) if b is not None else a.float().cpu()).pow(2).mean().sqrt())

    def codes_close(q, q_ref, what):
        d, dr = _deq(q), _deq(q_ref.to(DEV))
        bad = d != dr
        frac = float(bad.float().mean())
        assert frac < 2e-3, (what, frac)
        assert bool(((d - dr).abs()[bad] <= dr.abs().clamp_min(2.0 ** -6)[bad] * 0.1261).all()), what
        return frac

    def oracle_quant(xin):          # (rows, K) bf16 on the CPU -> the restatement's codes and scales (fake_quant_rows_e4m3, undone)
        xf = xin.float()
        s = xf.abs().amax(dim=-1, keepdim=True).clamp_min(1e-30) / 448.0
        return (xf / s).to(torch.float8_e4m3fn).view(torch.uint8), s[:, 0]

    report = []
    # (1) AdaLN-apply + quantisation in one pass: the operands of QKVG and of w1 | w3
    for ad, lin in (("blocks.0.attention_adaln", "blocks.0.attention.wq"), ("blocks.0.mlp_adaln", "blocks.0.mlp.w1")):
        xin = taps[f"{ad}.in"][0].to(DEV)                                   # (S, D) bf16 residual stream
        s1p, sh = taps[f"{ad}.scale1p"][0, 0].to(DEV).contiguous(), taps[f"{ad}.shift"][0, 0].to(DEV).contiguous()
        q = torch.empty((S, D), dtype=torch.uint8, device=DEV)
        sc = torch.empty((S,), dtype=torch.float32, device=DEV)
        L.check(U.lib().echo_op_norm_adaln_fp8(xin.data_ptr(), D, q.data_ptr(), D, sc.data_ptr(), S, D, cfg.norm_eps, s1p.data_ptr(),
                                               sh.data_ptr(), U.stream()))
        q_ref, s_ref = oracle_quant(taps[f"{lin}.in"][0])                   # the oracle's bf16 AdaLN output, quantised by the restatement
        # the norm itself may differ from torch's by one bf16 ulp in a few elements (rsqrt, fp32 product order): such a row's maximum,
        # hence its scale, can move by one bf16 step - everywhere else the scales are bit-equal
        same = sc.cpu() == s_ref
        assert float(same.float().mean()) > 0.97, float(same.float().mean())
        assert bool(((sc.cpu() - s_ref).abs() <= s_ref * 2.0 ** -7).all())
        frac = float((_deq(q)[same.to(DEV)] != _deq(q_ref.to(DEV))[same.to(DEV)]).float().mean())
        assert frac < 1e-2, (ad, frac)
        report.append(f"{ad.split('.')[-1]}: scales equal {float(same.float().mean()):.4f}, codes differing {frac:.2e}")
    # (2) every linear: quantise the oracle's operand, run the fp8 GEMM with the engine's tail, compare with the oracle's result
    def w_of(name):
        return wb[f"blocks.0.{name}.weight"].to(DEV)
    cases = [("attention.wq", D, D, 0), ("attention.wk", D, D, 0), ("attention.wv", D, D, 0), ("attention.gate", D, D, 0),
             ("attention.wo", D, D, 0), ("mlp.w1", F, D, 1), ("mlp.w2", D, F, 0)]
    for name, N, K, swiglu in cases:
        a = taps[f"blocks.0.{name}.in"][0].to(DEV).contiguous()              # (S, K) bf16: the oracle's operand
        a8, sa = U.quant_rows_fp8(a)
        q_ref, s_ref = oracle_quant(a.cpu())
        assert torch.equal(sa.cpu(), s_ref), name
        fa = codes_close(a8, q_ref, name)
        if swiglu:
            Wp = U.pack_swiglu(w_of("mlp.w1"), w_of("mlp.w3"))
            w8, sw = U.quant_rows_fp8(Wp)
            out = torch.zeros((S, F), dtype=torch.bfloat16, device=DEV)
            U.gemm(a8, w8, out, M=S, N=2 * F, K=K, lda=K, ldw=K, ldc=F, swiglu=1, Npad=Wp.shape[0], cfg=5, a_scale=sa, w_scale=sw)
            want = taps["blocks.0.mlp.w2.in"][0]                             # silu(w1 x) * w3 x, the restatement's w2 operand
            plain = (torch.nn.functional.silu(torch.nn.functional.linear(a.cpu(), wb["blocks.0.mlp.w1.weight"])) *
                     torch.nn.functional.linear(a.cpu(), wb["blocks.0.mlp.w3.weight"]))
        else:
            W = U.pad_rows(w_of(name))
            w8, sw = U.quant_rows_fp8(W)
            wq_ref, ws_ref = oracle_quant(w_of(name).cpu())
            assert torch.equal(sw[:N].cpu(), ws_ref), name
            codes_close(w8[:N].contiguous(), wq_ref, name + " (weight)")
            out = torch.zeros((S, N), dtype=torch.bfloat16, device=DEV)
            U.gemm(a8, w8, out, M=S, N=N, K=K, lda=K, ldw=K, ldc=N, cfg=5, a_scale=sa, w_scale=sw)
            want = taps[f"blocks.0.{name}.out"][0]
            plain = torch.nn.functional.linear(a.cpu(), wb[f"blocks.0.{name}.weight"])
            ref = (_deq(a8).double() @ _deq(w8)[:N].double().T) * sa.double()[:, None] * sw.double()[None, :N]
            U.bf16_close(out, ref.float().bfloat16(), ulps=2.0, atol=5e-4 * float(ref.pow(2).mean().sqrt()) + 2e-3)
        e, effect = rms(out, want), rms(want, plain)
        report.append(f"{name}: operand codes differing {fa:.2e}; out vs restatement {e:.3e}, fp8's effect on this linear {effect:.3e} ({e / effect:.3f})")
        assert effect > 0 and e < 0.25 * effect, (name, e, effect)
    # (3) static activation scale (calibrated): the SwiGLU tail writes w2's e4m3 operand itself
    a = taps["blocks.0.mlp.w1.in"][0].to(DEV).contiguous()
    a8, sa = U.quant_rows_fp8(a)
    Wp = U.pack_swiglu(w_of("mlp.w1"), w_of("mlp.w3"))
    w8, sw = U.quant_rows_fp8(Wp)
    h = taps["blocks.0.mlp.w2.in"][0]
    s_static = float(h.float().abs().max()) / 448.0 * 0.8                    # 0.8: some values saturate
    c8 = torch.zeros((S, F), dtype=torch.uint8, device=DEV)
    dummy = torch.zeros((S, F), dtype=torch.bfloat16, device=DEV)
    U.gemm(a8, w8, dummy, M=S, N=2 * F, K=D, lda=D, ldw=D, ldc=F, swiglu=1, Npad=Wp.shape[0], cfg=5, a_scale=sa, w_scale=sw, c8=c8, c8_inv=1.0 / s_static)
    ref8 = (h.float() / s_static).clamp(-448.0, 448.0).to(torch.float8_e4m3fn).view(torch.uint8)      # fake_quant_static_e4m3's codes
    d, dr = _deq(c8), _deq(ref8.to(DEV))
    frac = float((d != dr).float().mean())
    report.append(f"static SwiGLU tail: codes differing {frac:.2e}, saturated {float((d.abs() == 448.0).float().mean()):.2e}")
    assert frac < 2e-2, frac                       # the bf16 h differs from the restatement's in ~1 % of the elements (previous case), each a code apart at most
    assert rms(d, dr) < 0.05 * rms(dr)
    print("\n".join(["fp8 block linears, teacher-forced (d = 2048, one layer):"] + report))
