// Shared by gemm.hip (tile kernels) and gemm_pp.hip (the persistent ping-pong kernel): LDS-DMA helpers, the fused row tails and the
// split-K reduce kernel.  Everything lives in an unnamed namespace: each translation unit gets its own copy, and the two files compile in parallel.
#pragma once
#include "common.h"
#include <cstdlib>
#include <cstring>

namespace {


constexpr int KBYTES = 128;   // K step: 128 bytes per row (64 bf16 / 32 fp32)
#ifndef DMA_SPREAD_DEN
#define DMA_SPREAD_DEN 2
#endif

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __forceinline__ void glds16(const char* src, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)lds_wave_base, 16, 0, 0);
}

// s_waitcnt vmcnt(N) as the BUILTIN, not as inline asm: the compiler's wait-count pass then knows what has completed behind it.  As opaque asm it
// did not, and where a spilled value was reloaded in front of the K loop (the fp8 QKV instantiation: the DMA offsets) it put its own vmcnt(0) in
// front of the first use INSIDE the loop - every K-tile, draining the LDS-DMA ring this counted wait exists to keep full.
// simm16 (gfx9): vmcnt[3:0] | expcnt[6:4] = 7 | lgkmcnt[11:8] = 15 | vmcnt[5:4] << 14
template <int N> __device__ __forceinline__ void wait_vmcnt() { __builtin_amdgcn_s_waitcnt((N & 15) | (7 << 4) | (15 << 8) | ((N >> 4) << 14)); }

template <typename T>
__device__ __forceinline__ float vec_at(const void* p, long i) { return Num<T>::ld(((const T*)p)[i]); }

// ---- fused tail (one rounding to T wherever eager PyTorch materialises a tensor)
// Split into the loads (per-column vectors and the residual row) and the arithmetic + stores: CDNA4 has ONE in-order vmcnt
// for loads and stores, so a load issued behind a store waits for that store's round trip too.  The epilogues request the
// operands of output chunk k + 1 before they store chunk k.
struct TailCols { float bias[4], cs[4], al[4], ial[4]; };   // per-column operands of 4 consecutive output columns (ial = 1 / (alpha + 1e-9))

template <typename T>
__device__ __forceinline__ void gemm_tail_cols(const GemmArgs& p, int n0, int zo, int zi, TailCols& t) {
  const int nv = p.vec_mod ? n0 % p.vec_mod : n0;
  if (p.bias) {
    const long bo = zo * p.bias_bo + zi * p.bias_bi;
#pragma unroll
    for (int i = 0; i < 4; ++i) t.bias[i] = vec_at<T>(p.bias, bo + nv + i);
  }
  if (p.colscale) {
#pragma unroll
    for (int i = 0; i < 4; ++i) t.cs[i] = vec_at<T>(p.colscale, nv + i);
  }
  if (p.snake_alpha) {
#pragma unroll
    for (int i = 0; i < 4; ++i) { t.al[i] = vec_at<T>(p.snake_alpha, nv + i); t.ial[i] = 1.0f / (t.al[i] + 1e-9f); }
  }
}

template <typename T>
__device__ __forceinline__ void gemm_tail_res(const GemmArgs& p, int m, int n0, int zo, int zi, float (&r)[4]) {
  typedef Vec4<T> V;
  if (p.res) V::unpack(*(const typename V::raw*)((const T*)p.res + zo * p.res_bo + zi * p.res_bi + (long)m * p.ldres + n0), r);
}

template <typename T>
__device__ __forceinline__ void gemm_tail_apply(const GemmArgs& p, int m, int n0, float (&y)[4], const TailCols& t, const float (&res)[4], T* C,
                                                T* C2) {
  typedef Vec4<T> V;
  if (p.acc_scale != 1.0f) {
#pragma unroll
    for (int i = 0; i < 4; ++i) y[i] *= p.acc_scale;
  }
  if (p.bias) {
#pragma unroll
    for (int i = 0; i < 4; ++i) y[i] += t.bias[i];
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) y[i] = Num<T>::rnd(y[i]);
  if (p.div != 0.0f) {
#pragma unroll
    for (int i = 0; i < 4; ++i) y[i] = Num<T>::rnd(y[i] / p.div);
  }
  if (p.act == 1) {
#pragma unroll
    for (int i = 0; i < 4; ++i) y[i] = Num<T>::rnd(silu_f(y[i]));
  } else if (p.act == 2) {
#pragma unroll
    for (int i = 0; i < 4; ++i) y[i] = Num<T>::rnd(gelu_erf_f(y[i]));
  }
  if (p.colscale) {
#pragma unroll
    for (int i = 0; i < 4; ++i) y[i] = Num<T>::rnd(y[i] * t.cs[i]);
  }
  if (p.res) {
#pragma unroll
    for (int i = 0; i < 4; ++i) y[i] = Num<T>::rnd(y[i] + res[i]);
  }
  if (p.store_main) *(typename V::raw*)(C + (long)m * p.ldc + n0) = V::pack(y);
  if (p.snake_alpha) {
    float sn4[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float al = t.al[i];
      const float sn = sinf(al * y[i]);
      sn4[i] = Num<T>::rnd(y[i] + t.ial[i] * (sn * sn));      // the reciprocal is a per-column constant: same value, divided once per tile
    }
    *(typename V::raw*)(C2 + (long)m * p.ldc + n0) = V::pack(sn4);
  }
}

template <typename T>
__device__ __forceinline__ void gemm_tail(const GemmArgs& p, int m, int n0, float (&y)[4], int zo, int zi, T* C, T* C2) {
  TailCols t;
  float res[4] = {0.f, 0.f, 0.f, 0.f};
  gemm_tail_cols<T>(p, n0, zo, zi, t);
  gemm_tail_res<T>(p, m, n0, zo, zi, res);
  gemm_tail_apply<T>(p, m, n0, y, t, res, C, C2);
}

template <typename T>
__device__ __forceinline__ void swiglu_tail(const GemmArgs& p, int m, int j0, const f32x4& a4, const f32x4& b4, T* C) {
  float o[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float a = Num<T>::rnd(a4[i]);
    const float bb = Num<T>::rnd(b4[i]);
    o[i] = Num<T>::rnd(Num<T>::rnd(Num<T>::is_bf16 ? silu_fast(a) : silu_f(a)) * bb);
  }
  *(typename Vec4<T>::raw*)(C + (long)m * p.ldc + j0) = Vec4<T>::pack(o);
}

// sums the split-K partial slabs in split order (deterministic) and applies the fused tail
template <typename T, bool SWIGLU>
__global__ void __launch_bounds__(256) splitk_reduce_kernel(const GemmArgs p, int Mpad) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  const float* ws = (const float*)p.ws;
  const long slab = (long)Mpad * p.Npad;
  if constexpr (SWIGLU) {
    const int cpr = p.Npad / 8;                      // output chunks per row
    const int m = (int)(idx / cpr), pc = (int)(idx % cpr);
    const int blk = pc >> 2, q = pc & 3;             // 32-row packed block, 4-column group
    const int j0 = blk * 16 + q * 4;
    if (m >= p.M || j0 >= (p.N >> 1)) return;
    f32x4 a4 = {0, 0, 0, 0}, b4 = {0, 0, 0, 0};
    for (int s = 0; s < p.ksplit; ++s) {
      const float* r = ws + s * slab + (long)m * p.Npad + blk * 32 + q * 4;
      const f32x4 x = *(const f32x4*)r, y = *(const f32x4*)(r + 16);
      a4 += x; b4 += y;
    }
    swiglu_tail<T>(p, m, j0, a4, b4, (T*)p.C);
  } else {
    const int cpr = p.Npad / 4;
    const int m = (int)(idx / cpr), n0 = (int)(idx % cpr) * 4;
    if (m >= p.M || n0 >= p.N) return;
    f32x4 a4 = {0, 0, 0, 0};
    for (int s = 0; s < p.ksplit; ++s) a4 += *(const f32x4*)(ws + s * slab + (long)m * p.Npad + n0);
    float y[4] = {a4[0], a4[1], a4[2], a4[3]};
    gemm_tail<T>(p, m, n0, y, 0, 0, (T*)p.C, (T*)p.C2);
  }
}

}  // namespace

// gemm_pp.hip: the ping-pong kernel behind plan cfg 5 and its diagnostic builds (cfg 101-111); bf16 activations / outputs only
hipError_t launch_gemm_pp(const GemmArgs& g, hipStream_t st);
