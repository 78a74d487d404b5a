"""Helpers for the -m gpu tests: call libechohip's single-kernel entry points on torch (ROCm) tensors."""
from __future__ import annotations

import ctypes as C

import torch

import echo_tts_amd  # noqa: F401  (package alias)
from echo_tts_amd import _lib as L

DEV = "cuda:0"


def lib():
    return L.load_library()


def stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def code(t: torch.Tensor) -> int:
    return L.ECHO_BF16 if t.dtype == torch.bfloat16 else L.ECHO_F32


def ptr(t):
    return None if t is None else t.data_ptr()


def pad_rows(w: torch.Tensor, mult: int = 128) -> torch.Tensor:
    n = w.shape[0]
    npad = (n + mult - 1) // mult * mult
    out = torch.zeros((npad,) + tuple(w.shape[1:]), dtype=w.dtype, device=w.device)
    out[:n] = w
    return out


def gemm(A, W, C_out, *, M=None, N=None, K=None, lda=None, ldw=None, ldc=None, C2=None, taps=1, tap_base=0, tap_shift=0,
         nbatch=1, nbi=1, a_bo=0, a_bi=0, w_bo=0, w_bi=0, c_bo=0, c_bi=0, acc_scale=1.0, bias=None, bias_bo=0, bias_bi=0,
         vec_mod=0, div=0.0, act=0, colscale=None, res=None, ldres=0, res_bo=0, res_bi=0, snake_alpha=None, store_main=1,
         swiglu=0, Npad=None, a_offset_elems=0, cfg=0, ksplit=1, split3=0, ws=None, a_scale=None, w_scale=None, qkv=None, w_presplit=0, a_scale_const=0.0, c8=None, c8_inv=0.0):
    d = L.EchoGemmDesc()
    es = A.element_size()
    d.A = A.data_ptr() + a_offset_elems * es
    d.W, d.C, d.C2 = W.data_ptr(), C_out.data_ptr(), ptr(C2)
    d.M, d.N, d.K = M, N, K
    d.Npad = Npad if Npad is not None else (N + 127) // 128 * 128
    d.lda, d.ldw, d.ldc = lda, ldw, ldc
    d.taps, d.tap_base, d.tap_shift = taps, tap_base, tap_shift
    d.nbatch, d.nbi = nbatch, nbi
    d.a_bo, d.a_bi, d.w_bo, d.w_bi, d.c_bo, d.c_bi = a_bo, a_bi, w_bo, w_bi, c_bo, c_bi
    d.acc_scale = acc_scale
    d.bias, d.bias_bo, d.bias_bi, d.vec_mod = ptr(bias), bias_bo, bias_bi, vec_mod
    d.div, d.act = div, act
    d.colscale = ptr(colscale)
    d.res, d.ldres, d.res_bo, d.res_bi = ptr(res), ldres, res_bo, res_bi
    d.snake_alpha = ptr(snake_alpha)
    d.store_main, d.swiglu = store_main, swiglu
    d.cfg, d.ksplit, d.split3, d.w_presplit = cfg, ksplit, split3, w_presplit
    if a_scale is not None or a_scale_const:   # e4m3 operands: A / W are uint8 tensors, the output C decides the dtype code
        d.fp8, d.a_scale, d.w_scale, d.a_scale_const = 1, ptr(a_scale), w_scale.data_ptr(), a_scale_const
    if c8 is not None:        # SwiGLU tail of the fp8 kernel writing e4m3 bytes (static activation scale)
        d.c8, d.c8_ld, d.c8_inv = c8.data_ptr(), c8.stride(0), c8_inv
    if qkv is not None:       # fused QKV(G) tail: dict(D, S, rope_heads, pos0, eps, qk_w, rope, vt, vt_ld, vt_row_stride)
        d.qkv_mode, d.qkv_D, d.qkv_S, d.rope_heads, d.pos0, d.qk_eps = 1, qkv["D"], qkv["S"], qkv["rope_heads"], qkv.get("pos0", 0), qkv["eps"]
        d.qk_w, d.rope, d.vt, d.vt_ld, d.vt_row_stride = qkv["qk_w"].data_ptr(), qkv["rope"].data_ptr(), qkv["vt"].data_ptr(), qkv["vt_ld"], qkv["vt_row_stride"]
        d.qkv_gate_act = int(qkv.get("gate_act", 0))
    if ws is not None:
        d.ws, d.ws_bytes = ws.data_ptr(), ws.numel() * ws.element_size()
    elif ksplit > 1:
        need = ksplit * ((M + 767) // 768 * 768) * d.Npad * 4   # rows padded for every tile height (128 / 256 / 384)
        ws = torch.empty((need,), dtype=torch.uint8, device=A.device)
        d.ws, d.ws_bytes = ws.data_ptr(), need
    L.check(lib().echo_op_gemm(code(C_out if (a_scale is not None or a_scale_const) else A), C.byref(d), stream()))


def pack_swiglu(w1: torch.Tensor, w3: torch.Tensor) -> torch.Tensor:
    """[16 rows of w1 | 16 rows of w3] blocks, the layout gemm.hip's SWIGLU epilogue expects."""
    f, k = w1.shape
    out = torch.zeros(((2 * f + 127) // 128 * 128, k), dtype=w1.dtype, device=w1.device)
    for half, w in ((0, w1), (1, w3)):
        L.check(lib().echo_op_pack_rows(w.data_ptr(), code(w), k, out.data_ptr(), code(out), k, f, k, 0, half, stream()))
    return out


def bf16_close(out: torch.Tensor, ref: torch.Tensor, ulps: float = 2.0, atol: float = 0.0, frac_exact: float = 0.0):
    """out, ref as fp32 values of bf16 numbers: |out-ref| <= ulps * 2^-7 * |ref| + atol everywhere."""
    o, r = out.float(), ref.float()
    err = (o - r).abs()
    tol = ulps * (2.0 ** -7) * r.abs() + atol
    bad = err > tol
    assert not bool(bad.any()), f"{int(bad.sum())} / {bad.numel()} elements off; max err {float(err.max()):.4g}, worst ref {float(r[bad].abs().max()):.4g}"
    if frac_exact:
        fe = float((o == r).float().mean())
        assert fe >= frac_exact, f"only {fe:.4f} of elements match the reference rounding exactly"


def rms(x: torch.Tensor) -> float:
    return float(x.float().pow(2).mean().sqrt())


def quant_rows_fp8(x: torch.Tensor):
    """bf16 (rows, K) -> (uint8 e4m3 bytes (rows, K), fp32 scale (rows,)) through echo_op_quant_rows_fp8."""
    rows, K = x.shape
    q = torch.empty((rows, K), dtype=torch.uint8, device=x.device)
    s = torch.empty((rows,), dtype=torch.float32, device=x.device)
    L.check(lib().echo_op_quant_rows_fp8(x.data_ptr(), x.stride(0), q.data_ptr(), K, s.data_ptr(), rows, K, stream()))
    return q, s


class FlushAlloc:
    """A dedicated hipMalloc whose payload ENDS exactly at the end of the allocation (sizes rounded up to 2 MiB, the granule the
    driver maps): a kernel that reads or writes even one byte past the operand touches the next, normally unmapped, page and
    faults deterministically instead of silently reading a neighbour.  Duck-types the two tensor methods gemm() uses."""

    GRAN = 2 << 20

    def __init__(self, src: torch.Tensor):
        self._hip = C.CDLL("libamdhip64.so")
        self._hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        self._hip.hipFree.argtypes = [C.c_void_p]
        self._hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        self._hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
        src = src.contiguous()
        self.nbytes = src.numel() * src.element_size()
        self.alloc = (self.nbytes + self.GRAN - 1) // self.GRAN * self.GRAN
        base = C.c_void_p()
        assert self._hip.hipMalloc(C.byref(base), self.alloc) == 0
        self._base = base.value
        self._ptr = self._base + self.alloc - self.nbytes
        assert self._ptr % 16 == 0, "payload must stay 16-byte aligned"
        torch.cuda.synchronize()
        assert self._hip.hipMemset(self._base, 0, self.alloc) == 0
        assert self._hip.hipMemcpy(self._ptr, src.data_ptr(), self.nbytes, 3) == 0     # hipMemcpyDeviceToDevice
        self.shape, self.dtype, self._es = tuple(src.shape), src.dtype, src.element_size()

    def data_ptr(self) -> int:
        return self._ptr

    def element_size(self) -> int:
        return self._es

    def to_tensor(self) -> torch.Tensor:
        out = torch.empty(self.shape, dtype=self.dtype, device=DEV)
        torch.cuda.synchronize()
        assert self._hip.hipMemcpy(out.data_ptr(), self._ptr, self.nbytes, 3) == 0
        return out

    def free(self):
        if self._base:
            torch.cuda.synchronize()
            self._hip.hipFree(self._base)
            self._base = 0

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass
