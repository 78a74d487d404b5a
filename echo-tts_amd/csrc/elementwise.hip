// HBM-bound row / elementwise kernels of the hot path (wave64 shuffles, 16-byte vector access).
// Each kernel cites the reference expression whose rounding points it reproduces
// (SURVEY.md §A.2): values are rounded to the activation type T exactly where eager PyTorch
// materialises a tensor; for T = float `rnd` is the identity.
#include "common.h"

namespace {

// ------------------------------------------------------------------ 8-element chunks
template <typename T> struct Chunk8;
template <> struct Chunk8<bf16_t> {
  __device__ __forceinline__ static void load(const bf16_t* p, float* f) {
    const uint4 r = *(const uint4*)p;
    const uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) { f[2 * i] = __uint_as_float(w[i] << 16); f[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u); }
  }
  __device__ __forceinline__ static void store(bf16_t* p, const float* f) {
    uint4 r;
    r.x = pack_bf16x2(f[0], f[1]);
    r.y = pack_bf16x2(f[2], f[3]);
    r.z = pack_bf16x2(f[4], f[5]);
    r.w = pack_bf16x2(f[6], f[7]);
    *(uint4*)p = r;
  }
};
template <> struct Chunk8<float> {
  __device__ __forceinline__ static void load(const float* p, float* f) {
    const float4 a = *(const float4*)p, b = *(const float4*)(p + 4);
    f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
  }
  __device__ __forceinline__ static void store(float* p, const float* f) {
    *(float4*)p = make_float4(f[0], f[1], f[2], f[3]);
    *(float4*)(p + 4) = make_float4(f[4], f[5], f[6], f[7]);
  }
};

// ------------------------------------------------------------------ row norms
// NORM_ADALN : model.py:76-83  y = T( (x * rsqrt(mean x^2 + eps)) * scale1p + shift ), scale1p = T(scale + 1)
// NORM_RMS_W : model.py:99-104 y = T( (x * rsqrt(...)) * w )
// NORM_AE_RMS: autoencoder.py:726-731 y = T( T(x * rsqrt(...)) * w )
// NORM_LAYER : F.layer_norm (autoencoder.py:364)
template <typename T, int MODE>
__global__ void __launch_bounds__(256) norm_kernel(const T* __restrict__ x, long ldx, T* __restrict__ y, long ldy, int rows,
                                                   int D, float eps, const T* __restrict__ w0, const T* __restrict__ w1) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  constexpr int MAXC = 8;  // up to 8 chunks of 8 per lane: D <= 4096
  const int nch = D >> 3;
  float v[MAXC][8];
  float s1 = 0.f, s2 = 0.f;
  const T* xr = x + (long)row * ldx;
#pragma unroll
  for (int c = 0; c < MAXC; ++c) {
    const int ch = lane + 64 * c;
    if (ch < nch) {
      Chunk8<T>::load(xr + ch * 8, v[c]);
#pragma unroll
      for (int i = 0; i < 8; ++i) { s2 += v[c][i] * v[c][i]; s1 += v[c][i]; }
    }
  }
  s2 = wave_sum(s2);
  float mean = 0.f, rs;
  if (MODE == NORM_LAYER) {
    s1 = wave_sum(s1);
    mean = s1 / (float)D;
    float var = 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
      const int ch = lane + 64 * c;
      if (ch < nch) {
#pragma unroll
        for (int i = 0; i < 8; ++i) { const float d = v[c][i] - mean; var += d * d; }
      }
    }
    var = wave_sum(var) / (float)D;
    rs = rsqrtf(var + eps);
  } else {
    rs = rsqrtf(s2 / (float)D + eps);
  }
  T* yr = y + (long)row * ldy;
#pragma unroll
  for (int c = 0; c < MAXC; ++c) {
    const int ch = lane + 64 * c;
    if (ch < nch) {
      float a[8], b[8], o[8];
      Chunk8<T>::load(w0 + ch * 8, a);
      if (MODE == NORM_ADALN || MODE == NORM_LAYER) Chunk8<T>::load(w1 + ch * 8, b);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (MODE == NORM_ADALN) o[i] = __fadd_rn(__fmul_rn(__fmul_rn(v[c][i], rs), a[i]), b[i]);
        else if (MODE == NORM_RMS_W) o[i] = __fmul_rn(__fmul_rn(v[c][i], rs), a[i]);
        else if (MODE == NORM_AE_RMS) o[i] = __fmul_rn(Num<T>::rnd(__fmul_rn(v[c][i], rs)), a[i]);
        else o[i] = __fadd_rn(__fmul_rn(__fmul_rn(v[c][i] - mean, rs), a[i]), b[i]);
      }
      Chunk8<T>::store(yr + ch * 8, o);
    }
  }
}

// NORM_ADALN for the EchoDiT width (bf16, D = 512 * NC <= 2048): a wave keeps its share of the two modulation vectors in registers and
// walks ROWS consecutive rows, all of whose loads are issued before the first is consumed.  The generic kernel above re-reads the 2 x D
// modulation values for every row (8 KB through the L1 per 4 KB row at D = 2048) and has one 4 KB round trip per wave; same arithmetic,
// same summation order, same bits.
template <int NC, int ROWS>
__global__ void __launch_bounds__(256) norm_adaln_rows_kernel(const bf16_t* __restrict__ x, long ldx, bf16_t* __restrict__ y, long ldy, int rows,
                                                              float eps, const bf16_t* __restrict__ w0, const bf16_t* __restrict__ w1) {
  const int lane = threadIdx.x & 63;
  const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * ROWS;
  if (row0 >= rows) return;
  constexpr int D = 512 * NC;
  uint4 xr[ROWS][NC];
#pragma unroll
  for (int r = 0; r < ROWS; ++r) {
    const int row = row0 + r < rows ? row0 + r : rows - 1;
#pragma unroll
    for (int c = 0; c < NC; ++c) xr[r][c] = *(const uint4*)(x + (long)row * ldx + (lane + 64 * c) * 8);
  }
  uint4 wa[NC], wb[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) { wa[c] = *(const uint4*)(w0 + (lane + 64 * c) * 8); wb[c] = *(const uint4*)(w1 + (lane + 64 * c) * 8); }
  auto unpack = [](const uint4& r, float* f) __attribute__((always_inline)) {
    const uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) { f[2 * i] = __uint_as_float(w[i] << 16); f[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u); }
  };
#pragma unroll
  for (int r = 0; r < ROWS; ++r) {
    if (row0 + r >= rows) break;
    float v[NC][8];
    float s2 = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      unpack(xr[r][c], v[c]);
#pragma unroll
      for (int i = 0; i < 8; ++i) s2 += v[c][i] * v[c][i];
    }
    s2 = wave_sum(s2);
    const float rs = rsqrtf(s2 / (float)D + eps);
    bf16_t* yr = y + (long)(row0 + r) * ldy;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      float a[8], b[8], o[8];
      unpack(wa[c], a);
      unpack(wb[c], b);
#pragma unroll
      for (int i = 0; i < 8; ++i) o[i] = __fadd_rn(__fmul_rn(__fmul_rn(v[c][i], rs), a[i]), b[i]);
      Chunk8<bf16_t>::store(yr + (lane + 64 * c) * 8, o);
    }
  }
}

// ------------------------------------------------------------------ per-head RMSNorm + RoPE (HD = 128)
// model.py:221-232 (q_norm/k_norm then _apply_rotary_half), model.py:138-142 (encoders: all heads),
// model.py:289-291 (latent keys, positions 4*i).  One wave per (token, head); lane j owns the
// interleaved pair (x[2j], x[2j+1]).  Rounds to T after the norm and again after the rotation.
template <typename T>
__global__ void __launch_bounds__(256) headnorm_rope_kernel(T* __restrict__ x, long ldx, long t_stride, int rows, int S, int H,
                                                            const T* __restrict__ w, long w_stride, float eps, int do_norm,
                                                            int rope_heads, const float2* __restrict__ rope, int pos0,
                                                            int pos_mul) {
  const int lane = threadIdx.x & 63;
  const long item = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (item >= (long)rows * H) return;
  const int row = (int)(item / H), h = (int)(item - (long)row * H);
  const int tsel = blockIdx.y;
  T* p = x + tsel * t_stride + (long)row * ldx + h * 128 + 2 * lane;
  float a = Num<T>::ld(p[0]), b = Num<T>::ld(p[1]);
  if (do_norm) {
    const float ss = wave_sum(a * a + b * b);
    const float rs = rsqrtf(ss / 128.0f + eps);
    const T* wp = w + tsel * w_stride + h * 128 + 2 * lane;
    a = Num<T>::rnd(__fmul_rn(__fmul_rn(a, rs), Num<T>::ld(wp[0])));
    b = Num<T>::rnd(__fmul_rn(__fmul_rn(b, rs), Num<T>::ld(wp[1])));
  }
  if (h < rope_heads) {
    const int pos = pos0 + (row % S) * pos_mul;
    const float2 cs = rope[(long)pos * 64 + lane];
    const float re = __fsub_rn(__fmul_rn(a, cs.x), __fmul_rn(b, cs.y));
    const float im = __fadd_rn(__fmul_rn(a, cs.y), __fmul_rn(b, cs.x));
    a = re; b = im;
  }
  p[0] = Num<T>::st(a);
  p[1] = Num<T>::st(b);
}

// ------------------------------------------------------------------ V -> Vᵀ per head
// Vt[b][h][d][s] = V[b*S + s][h*HD + d]; a workgroup moves a 64(s) x HD tile through LDS.  Keys
// s >= S inside the last tile are written as zeros so that the padding of Vᵀ stays finite.
template <typename T, int HD>
__global__ void __launch_bounds__(256) transpose_heads_kernel(const T* __restrict__ v, long ldv, T* __restrict__ vt, long vt_ld,
                                                              long vt_b_stride, int S, int H) {
  __shared__ T tile[64][HD + 2];
  const int s0 = blockIdx.x * 64, h = blockIdx.y, b = blockIdx.z;
  for (int i = threadIdx.x; i < 64 * HD; i += 256) {
    const int r = i / HD, d = i - r * HD;
    const int s = s0 + r;
    tile[r][d] = s < S ? v[((long)b * S + s) * ldv + h * HD + d] : (T)0;
  }
  __syncthreads();
  T* out = vt + (long)b * vt_b_stride + (long)h * HD * vt_ld;
  for (int i = threadIdx.x; i < 64 * HD; i += 256) {
    const int d = i >> 6, r = i & 63;
    out[(long)d * vt_ld + s0 + r] = tile[r][d];
  }
}

template <typename T>
__global__ void embedding_kernel(const int* __restrict__ ids, const T* __restrict__ table, T* __restrict__ out, long ldo, int n, int D) {
  const int row = blockIdx.x;
  if (row >= n) return;
  const T* src = table + (long)ids[row] * D;
  for (int c = threadIdx.x; c < D; c += blockDim.x) out[(long)row * ldo + c] = src[c];
}

template <typename T>
__global__ void silu_kernel(const T* __restrict__ x, T* __restrict__ y, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] = Num<T>::st(silu_f(Num<T>::ld(x[i])));
}

template <typename T>
__global__ void scale2d_kernel(T* __restrict__ x, long ld, int rows, int cols, float s) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)rows * cols) return;
  const long r = i / cols, c = i - r * cols;
  T* p = x + r * ld + c;
  *p = Num<T>::st(__fmul_rn(Num<T>::ld(*p), s));
}

// mod table rows = [shift | scale | gate] x D: scale -> T(scale + 1) (model.py:79), gate -> T(tanh(gate)) (model.py:81)
template <typename T>
__global__ void mod_finalize_kernel(T* __restrict__ mod, long rows, int D) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * 3 * D) return;
  const int part = (int)((i / D) % 3);
  if (part == 0) return;
  const float v = Num<T>::ld(mod[i]);
  mod[i] = Num<T>::st(part == 1 ? __fadd_rn(v, 1.0f) : tanhf(v));
}

template <typename T>
__global__ void from_f32_kernel(const float* __restrict__ x, long ldx, T* __restrict__ y, long ldy, int rows, int cols, int cols_pad) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)rows * cols_pad) return;
  const long r = i / cols_pad;
  const int c = (int)(i - r * cols_pad);
  y[r * ldy + c] = c < cols ? Num<T>::st(x[r * ldx + c]) : (T)0;
}

template <typename T>
__global__ void to_f32_kernel(const T* __restrict__ x, long ldx, float* __restrict__ y, long ldy, int rows, int cols) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)rows * cols) return;
  const long r = i / cols;
  const int c = (int)(i - r * cols);
  y[r * ldy + c] = Num<T>::ld(x[r * ldx + c]);
}

__device__ __forceinline__ float load_any(const void* p, int dt, long i) {
  return dt == 0 ? ((const float*)p)[i] : bf2f(((const bf16_t*)p)[i]);
}
__device__ __forceinline__ void store_any(void* p, int dt, long i, float v) {
  if (dt == 0) ((float*)p)[i] = v; else ((bf16_t*)p)[i] = f2bf(v);
}

// dst[dst_row0 + map(r)][c] = src[r][c] with dtype conversion (0 = f32, 1 = bf16).
// swiglu_half h in {0,1}: source row j goes to packed row (j/16)*32 + h*16 + j%16  (gemm.hip SWIGLU layout).
__global__ void pack_rows_kernel(const void* __restrict__ src, int sdt, long sld, void* __restrict__ dst, int ddt, long dld,
                                 int rows, int cols, int dst_row0, int swiglu_half) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)rows * cols) return;
  const long r = i / cols;
  const int c = (int)(i - r * cols);
  long dr = r;
  if (swiglu_half >= 0) dr = (r >> 4) * 32 + swiglu_half * 16 + (r & 15);
  store_any(dst, ddt, (dst_row0 + dr) * dld + c, load_any(src, sdt, r * sld + c));
}

// ------------------------------------------------------------------ CFG combine + rescale + Euler (inference.py:487-515)
template <typename T>
__global__ void euler_kernel(const EulerArgs e) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long n = (long)e.B * e.S * e.L;
  if (i >= n) return;
  const long tok = i / e.L;
  const int l = (int)(i - tok * e.L);
  float x = e.x[i];
  if (e.init) {
    if (e.init_scale != 1.0f) x = __fmul_rn(x, e.init_scale);
  } else {
    const T* v = (const T*)e.v;
    const long bs = (long)e.B * e.S;
    float vp = Num<T>::ld(v[tok * e.ldv + l]);
    if (e.R == 3) {
      const float vut = Num<T>::ld(v[(bs + tok) * e.ldv + l]);
      const float vus = Num<T>::ld(v[(2 * bs + tok) * e.ldv + l]);
      const float a = __fmul_rn(e.s_text, __fsub_rn(vp, vut));
      const float b = __fmul_rn(e.s_spk, __fsub_rn(vp, vus));
      vp = __fadd_rn(__fadd_rn(vp, a), b);
    }
    if (e.rescale) {
      float t = __fadd_rn(__fmul_rn(e.r_1mt, vp), x);
      t = __fsub_rn(__fmul_rn(e.r_ratio, t), x);
      vp = __fmul_rn(e.r_inv1mt, t);
    }
    x = __fadd_rn(x, __fmul_rn(vp, e.dt));
  }
  e.x[i] = x;
  T* xin = (T*)e.xin;
  const long bs = (long)e.B * e.S;
  for (int r = 0; r < e.R_next; ++r) xin[(r * bs + tok) * e.ld_xin + l] = Num<T>::st(x);
}

// ------------------------------------------------------------------ fp32 softmax with key bias / causal window
__global__ void __launch_bounds__(256) softmax_f32_kernel(float* __restrict__ s, long ld, int rows_per_batch, long total_rows, int ncols,
                                                          int ncols_pad, const float* __restrict__ bias, long bias_batch_stride,
                                                          int heads_per_bias_row, int causal, int window) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= total_rows) return;
  const long batch = row / rows_per_batch;
  const int qi = (int)(row - batch * rows_per_batch);
  float* p = s + row * ld;
  const float* bp = bias ? bias + (batch / heads_per_bias_row) * bias_batch_stride : nullptr;
  float mx = -INFINITY;
  for (int c = lane; c < ncols; c += 64) {
    float v = p[c];
    if (bp) v += bp[c];
    if (causal && (c > qi || (window > 0 && c < qi - window + 1))) v = -INFINITY;
    p[c] = v;
    mx = fmaxf(mx, v);
  }
  mx = wave_max(mx);
  float sum = 0.f;
  for (int c = lane; c < ncols; c += 64) {
    const float v = expf(p[c] - mx);
    p[c] = v;
    sum += v;
  }
  sum = wave_sum(sum);
  const float inv = 1.0f / sum;
  for (int c = lane; c < ncols_pad; c += 64) p[c] = c < ncols ? p[c] * inv : 0.0f;
}

__global__ void mask_to_bias_kernel(const uint8_t* __restrict__ m, float* __restrict__ b, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) b[i] = m[i] ? 0.0f : -INFINITY;
}

__global__ void convert_any_kernel(const void* __restrict__ src, int sdt, void* __restrict__ dst, int ddt, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) store_any(dst, ddt, i, load_any(src, sdt, i));
}

inline dim3 grid1d(long n, int bs = 256) { return dim3((unsigned)((n + bs - 1) / bs)); }

// NORM_ADALN fused with the e4m3 row quantisation (fp8 engine): the bf16 value the bf16 engine would have stored is formed in
// registers, its row maximum sets the scale, and only the e4m3 bytes + the scale leave the CU (1 byte per element instead
// of 2 written + 2 read back by quant_rows_fp8_kernel).  Same arithmetic as norm_kernel<bf16, NORM_ADALN> + quant_rows_fp8.
__global__ void __launch_bounds__(256) norm_adaln_fp8_kernel(const bf16_t* __restrict__ x, long ldx, uint8_t* __restrict__ q, long ldq,
                                                             float* __restrict__ scale, int rows, int D, float eps,
                                                             const bf16_t* __restrict__ w0, const bf16_t* __restrict__ w1) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  constexpr int MAXC = 8;
  const int nch = D >> 3;
  float v[MAXC][8];
  float s2 = 0.f;
  const bf16_t* xr = x + (long)row * ldx;
#pragma unroll
  for (int c = 0; c < MAXC; ++c) {
    const int ch = lane + 64 * c;
    if (ch < nch) {
      Chunk8<bf16_t>::load(xr + ch * 8, v[c]);
#pragma unroll
      for (int i = 0; i < 8; ++i) s2 += v[c][i] * v[c][i];
    }
  }
  s2 = wave_sum(s2);
  const float rs = rsqrtf(s2 / (float)D + eps);
  float amax = 0.f;
#pragma unroll
  for (int c = 0; c < MAXC; ++c) {
    const int ch = lane + 64 * c;
    if (ch < nch) {
      float a[8], b[8];
      Chunk8<bf16_t>::load(w0 + ch * 8, a);
      Chunk8<bf16_t>::load(w1 + ch * 8, b);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        v[c][i] = Num<bf16_t>::rnd(__fadd_rn(__fmul_rn(__fmul_rn(v[c][i], rs), a[i]), b[i]));
        amax = fmaxf(amax, fabsf(v[c][i]));
      }
    }
  }
  amax = wave_max(amax);
  const bool nz = amax > 0.0f;
  const float inv = nz ? __fdiv_rn(448.0f, amax) : 1.0f;
  if (lane == 0) scale[row] = nz ? __fdiv_rn(amax, 448.0f) : 1.0f;
  uint8_t* qr = q + (long)row * ldq;
#pragma unroll
  for (int c = 0; c < MAXC; ++c) {
    const int ch = lane + 64 * c;
    if (ch < nch) {
      int lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[c][0] * inv, v[c][1] * inv, 0, false);
      lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[c][2] * inv, v[c][3] * inv, lo, true);
      int hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[c][4] * inv, v[c][5] * inv, 0, false);
      hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[c][6] * inv, v[c][7] * inv, hi, true);
      *(uint2*)(qr + ch * 8) = uint2{(unsigned)lo, (unsigned)hi};
    }
  }
}

// the same with ROWS rows per wave and the modulation vectors in registers (D = 512 * NC; see norm_adaln_rows_kernel): same arithmetic, same bits
template <int NC, int ROWS>
__global__ void __launch_bounds__(256) norm_adaln_fp8_rows_kernel(const bf16_t* __restrict__ x, long ldx, uint8_t* __restrict__ q, long ldq,
                                                                  float* __restrict__ scale, int rows, float eps, const bf16_t* __restrict__ w0,
                                                                  const bf16_t* __restrict__ w1) {
  const int lane = threadIdx.x & 63;
  const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * ROWS;
  if (row0 >= rows) return;
  constexpr int D = 512 * NC;
  uint4 xr[ROWS][NC];
#pragma unroll
  for (int r = 0; r < ROWS; ++r) {
    const int row = row0 + r < rows ? row0 + r : rows - 1;
#pragma unroll
    for (int c = 0; c < NC; ++c) xr[r][c] = *(const uint4*)(x + (long)row * ldx + (lane + 64 * c) * 8);
  }
  uint4 wa[NC], wb[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) { wa[c] = *(const uint4*)(w0 + (lane + 64 * c) * 8); wb[c] = *(const uint4*)(w1 + (lane + 64 * c) * 8); }
  auto unpack = [](const uint4& r, float* f) __attribute__((always_inline)) {
    const uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) { f[2 * i] = __uint_as_float(w[i] << 16); f[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u); }
  };
#pragma unroll
  for (int r = 0; r < ROWS; ++r) {
    if (row0 + r >= rows) break;
    float v[NC][8];
    float s2 = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      unpack(xr[r][c], v[c]);
#pragma unroll
      for (int i = 0; i < 8; ++i) s2 += v[c][i] * v[c][i];
    }
    s2 = wave_sum(s2);
    const float rs = rsqrtf(s2 / (float)D + eps);
    float amax = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      float a[8], b[8];
      unpack(wa[c], a);
      unpack(wb[c], b);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        v[c][i] = Num<bf16_t>::rnd(__fadd_rn(__fmul_rn(__fmul_rn(v[c][i], rs), a[i]), b[i]));
        amax = fmaxf(amax, fabsf(v[c][i]));
      }
    }
    amax = wave_max(amax);
    const bool nz = amax > 0.0f;
    const float inv = nz ? __fdiv_rn(448.0f, amax) : 1.0f;
    if (lane == 0) scale[row0 + r] = nz ? __fdiv_rn(amax, 448.0f) : 1.0f;
    uint8_t* qr = q + (long)(row0 + r) * ldq;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      int lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[c][0] * inv, v[c][1] * inv, 0, false);
      lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[c][2] * inv, v[c][3] * inv, lo, true);
      int hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[c][4] * inv, v[c][5] * inv, 0, false);
      hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[c][6] * inv, v[c][7] * inv, hi, true);
      *(uint2*)(qr + (lane + 64 * c) * 8) = uint2{(unsigned)lo, (unsigned)hi};
    }
  }
}

// bf16 rows -> OCP e4m3 bytes + one fp32 scale per row (the A / W operands of the fp8 ping-pong GEMM): one wave per row,
// scale = amax / 448 (1 for an all-zero row), q = e4m3(x * (448 / amax)), round-to-nearest-even by v_cvt_pk_fp8_f32.  The row is
// read twice (the second pass hits L2); 8 elements per lane and step: 16 bytes in, 8 bytes out.  K % 8 == 0.
__global__ void __launch_bounds__(256) quant_rows_fp8_kernel(const bf16_t* __restrict__ x, long ldx, uint8_t* __restrict__ q, long ldq,
                                                             float* __restrict__ scale, int rows, int K) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const bf16_t* xr = x + (long)row * ldx;
  const int nchunk = K >> 3;
  float amax = 0.0f;
  for (int c = lane; c < nchunk; c += 64) {
    const uint4 v = *(const uint4*)(xr + 8 * c);
    float f[8];
    Vec4<bf16_t>::unpack(uint2{v.x, v.y}, f);
    Vec4<bf16_t>::unpack(uint2{v.z, v.w}, f + 4);
#pragma unroll
    for (int e = 0; e < 8; ++e) amax = fmaxf(amax, fabsf(f[e]));
  }
  amax = wave_max(amax);
  const bool nz = amax > 0.0f;
  const float inv = nz ? __fdiv_rn(448.0f, amax) : 1.0f;      // correctly rounded: the operands are reproducible on any host
  if (lane == 0) scale[row] = nz ? __fdiv_rn(amax, 448.0f) : 1.0f;
  uint8_t* qr = q + (long)row * ldq;
  for (int c = lane; c < nchunk; c += 64) {
    const uint4 v = *(const uint4*)(xr + 8 * c);
    float f[8];
    Vec4<bf16_t>::unpack(uint2{v.x, v.y}, f);
    Vec4<bf16_t>::unpack(uint2{v.z, v.w}, f + 4);
    int lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[0] * inv, f[1] * inv, 0, false);
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[2] * inv, f[3] * inv, lo, true);
    int hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[4] * inv, f[5] * inv, 0, false);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[6] * inv, f[7] * inv, hi, true);
    *(uint2*)(qr + 8 * c) = uint2{(unsigned)lo, (unsigned)hi};
  }
}

}  // namespace

template <typename T>
hipError_t launch_norm(int mode, const T* x, long ldx, T* y, long ldy, int rows, int D, float eps, const T* w0, const T* w1,
                       hipStream_t st) {
  if (D % 8 || D > 4096 || rows <= 0) return hipErrorInvalidValue;
  dim3 grid((rows + 3) / 4);
  if constexpr (Num<T>::is_bf16) {
    static const int rows_per_wave = getenv("ECHO_NORM_ROWS") ? atoi(getenv("ECHO_NORM_ROWS")) : 2;     // 1 = the generic kernel (A/B aid); measured (tools/bench_norm.py, M = 15360 / 46080): 29.1 / 88.3 us generic, 21.3 / 67.2 with 2 rows per wave, 22.6 / 70.2 with 4
    if (mode == NORM_ADALN && D == 2048 && rows >= 4096 && rows_per_wave > 1 && (ldx & 7) == 0 && (ldy & 7) == 0) {
      if (rows_per_wave != 4) hipLaunchKernelGGL((norm_adaln_rows_kernel<4, 2>), dim3((rows + 7) / 8), dim3(256), 0, st, x, ldx, y, ldy, rows, eps, w0, w1);
      else hipLaunchKernelGGL((norm_adaln_rows_kernel<4, 4>), dim3((rows + 15) / 16), dim3(256), 0, st, x, ldx, y, ldy, rows, eps, w0, w1);
      return hipGetLastError();
    }
  }
  switch (mode) {
    case NORM_ADALN: hipLaunchKernelGGL((norm_kernel<T, NORM_ADALN>), grid, dim3(256), 0, st, x, ldx, y, ldy, rows, D, eps, w0, w1); break;
    case NORM_RMS_W: hipLaunchKernelGGL((norm_kernel<T, NORM_RMS_W>), grid, dim3(256), 0, st, x, ldx, y, ldy, rows, D, eps, w0, w1); break;
    case NORM_AE_RMS: hipLaunchKernelGGL((norm_kernel<T, NORM_AE_RMS>), grid, dim3(256), 0, st, x, ldx, y, ldy, rows, D, eps, w0, w1); break;
    case NORM_LAYER: hipLaunchKernelGGL((norm_kernel<T, NORM_LAYER>), grid, dim3(256), 0, st, x, ldx, y, ldy, rows, D, eps, w0, w1); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

// x points at tensor 0; nt tensors (q, k) spaced t_stride elements apart share the launch.
template <typename T>
hipError_t launch_headnorm_rope_nt(T* x, long ldx, long t_stride, int nt, int rows, int S, int H, const T* w, long w_stride,
                                   float eps, int do_norm, int rope_heads, const float2* rope, int pos0, int pos_mul,
                                   hipStream_t st) {
  const long items = (long)rows * H;
  dim3 grid((unsigned)((items + 3) / 4), nt);
  hipLaunchKernelGGL(headnorm_rope_kernel<T>, grid, dim3(256), 0, st, x, ldx, t_stride, rows, S, H, w, w_stride, eps, do_norm,
                     rope_heads, rope, pos0, pos_mul);
  return hipGetLastError();
}
template <typename T>
hipError_t launch_headnorm_rope(T* x, long ldx, int rows, int S, int H, const T* w, float eps, int do_norm, int rope_heads,
                                const float2* rope, int pos0, int pos_mul, hipStream_t st) {
  return launch_headnorm_rope_nt<T>(x, ldx, 0, 1, rows, S, H, w, 0, eps, do_norm, rope_heads, rope, pos0, pos_mul, st);
}

template <typename T>
hipError_t launch_transpose_heads(const T* v, long ldv, T* vt, long vt_ld, long vt_b_stride, int B, int S, int H, int HD,
                                  hipStream_t st) {
  dim3 grid((S + 63) / 64, H, B);
  if (vt_ld < (long)grid.x * 64) return hipErrorInvalidValue;
  if (HD == 128) hipLaunchKernelGGL((transpose_heads_kernel<T, 128>), grid, dim3(256), 0, st, v, ldv, vt, vt_ld, vt_b_stride, S, H);
  else if (HD == 64) hipLaunchKernelGGL((transpose_heads_kernel<T, 64>), grid, dim3(256), 0, st, v, ldv, vt, vt_ld, vt_b_stride, S, H);
  else return hipErrorInvalidValue;
  return hipGetLastError();
}

template <typename T>
hipError_t launch_embedding(const int* ids, const T* table, T* out, long ldo, int n, int D, hipStream_t st) {
  hipLaunchKernelGGL(embedding_kernel<T>, dim3(n), dim3(256), 0, st, ids, table, out, ldo, n, D);
  return hipGetLastError();
}
template <typename T> hipError_t launch_silu(const T* x, T* y, long n, hipStream_t st) {
  hipLaunchKernelGGL(silu_kernel<T>, grid1d(n), dim3(256), 0, st, x, y, n);
  return hipGetLastError();
}
template <typename T> hipError_t launch_scale_2d(T* x, long ld, int rows, int cols, float s, hipStream_t st) {
  hipLaunchKernelGGL(scale2d_kernel<T>, grid1d((long)rows * cols), dim3(256), 0, st, x, ld, rows, cols, s);
  return hipGetLastError();
}
template <typename T> hipError_t launch_scale_inplace(T* x, long n, float s, hipStream_t st) {
  return launch_scale_2d<T>(x, n, 1, (int)n, s, st);
}
template <typename T> hipError_t launch_mod_finalize(T* mod, long rows, int D, hipStream_t st) {
  hipLaunchKernelGGL(mod_finalize_kernel<T>, grid1d(rows * 3 * D), dim3(256), 0, st, mod, rows, D);
  return hipGetLastError();
}
template <typename T>
hipError_t launch_convert_from_f32(const float* x, long ldx, T* y, long ldy, int rows, int cols, int cols_pad, hipStream_t st) {
  hipLaunchKernelGGL(from_f32_kernel<T>, grid1d((long)rows * cols_pad), dim3(256), 0, st, x, ldx, y, ldy, rows, cols, cols_pad);
  return hipGetLastError();
}
template <typename T>
hipError_t launch_convert_to_f32(const T* x, long ldx, float* y, long ldy, int rows, int cols, hipStream_t st) {
  hipLaunchKernelGGL(to_f32_kernel<T>, grid1d((long)rows * cols), dim3(256), 0, st, x, ldx, y, ldy, rows, cols);
  return hipGetLastError();
}
hipError_t launch_convert_any(const void* src, int sdt, void* dst, int ddt, long n, hipStream_t st) {
  hipLaunchKernelGGL(convert_any_kernel, grid1d(n), dim3(256), 0, st, src, sdt, dst, ddt, n);
  return hipGetLastError();
}
hipError_t launch_pack_rows(const void* src, int sdt, long sld, void* dst, int ddt, long dld, int rows, int cols, int dst_row0,
                            int swiglu_half, hipStream_t st) {
  hipLaunchKernelGGL(pack_rows_kernel, grid1d((long)rows * cols), dim3(256), 0, st, src, sdt, sld, dst, ddt, dld, rows, cols,
                     dst_row0, swiglu_half);
  return hipGetLastError();
}
template <typename T> hipError_t launch_euler(const EulerArgs& e, hipStream_t st) {
  hipLaunchKernelGGL(euler_kernel<T>, grid1d((long)e.B * e.S * e.L), dim3(256), 0, st, e);
  return hipGetLastError();
}
hipError_t launch_softmax_f32(float* s, long ld, int rows_per_batch, int nbatch, int ncols, int ncols_pad, const float* bias,
                              long bias_batch_stride, int heads_per_bias_row, int causal, int window, hipStream_t st) {
  const long total = (long)rows_per_batch * nbatch;
  hipLaunchKernelGGL(softmax_f32_kernel, dim3((unsigned)((total + 3) / 4)), dim3(256), 0, st, s, ld, rows_per_batch, total, ncols,
                     ncols_pad, bias, bias_batch_stride, heads_per_bias_row < 1 ? 1 : heads_per_bias_row, causal, window);
  return hipGetLastError();
}
hipError_t launch_norm_adaln_fp8(const void* x, long ldx, void* q, long ldq, float* scale, int rows, int D, float eps, const void* scale1p,
                                 const void* shift, hipStream_t st) {
  if (rows < 1 || D < 8 || (D & 7) || D > 4096 || (ldx & 7) || (ldq & 7)) return hipErrorInvalidValue;
  static const int rows_per_wave = getenv("ECHO_NORM_ROWS") ? atoi(getenv("ECHO_NORM_ROWS")) : 2;
  if (D == 2048 && rows >= 4096 && rows_per_wave > 1) {
    hipLaunchKernelGGL((norm_adaln_fp8_rows_kernel<4, 2>), dim3((unsigned)((rows + 7) / 8)), dim3(256), 0, st, (const bf16_t*)x, ldx, (uint8_t*)q, ldq, scale,
                       rows, eps, (const bf16_t*)scale1p, (const bf16_t*)shift);
    return hipGetLastError();
  }
  hipLaunchKernelGGL(norm_adaln_fp8_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, (const bf16_t*)x, ldx, (uint8_t*)q, ldq, scale,
                     rows, D, eps, (const bf16_t*)scale1p, (const bf16_t*)shift);
  return hipGetLastError();
}
__global__ void __launch_bounds__(256) max_into_kernel(const float* __restrict__ v, int n, float* dst) {
  float m = 0.f;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) m = fmaxf(m, v[i]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  // non-negative floats order like their bit patterns
  if ((threadIdx.x & 63) == 0 && m > 0.f) atomicMax((unsigned*)dst, __float_as_uint(m));
}
hipError_t launch_max_into(const float* v, int n, float* dst, hipStream_t st) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(max_into_kernel, dim3((unsigned)std::min((n + 255) / 256, 64)), dim3(256), 0, st, v, n, dst);
  return hipGetLastError();
}

hipError_t launch_quant_rows_fp8(const void* x, long ldx, void* q, long ldq, float* scale, int rows, int K, hipStream_t st) {
  if (rows < 1 || K < 8 || (K & 7) || (ldx & 7) || (ldq & 7)) return hipErrorInvalidValue;
  hipLaunchKernelGGL(quant_rows_fp8_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, (const bf16_t*)x, ldx, (uint8_t*)q, ldq, scale,
                     rows, K);
  return hipGetLastError();
}
hipError_t launch_mask_to_bias(const uint8_t* mask, float* bias, long n, hipStream_t st) {
  hipLaunchKernelGGL(mask_to_bias_kernel, grid1d(n), dim3(256), 0, st, mask, bias, n);
  return hipGetLastError();
}

#define INST(T)                                                                                                              \
  template hipError_t launch_norm<T>(int, const T*, long, T*, long, int, int, float, const T*, const T*, hipStream_t);        \
  template hipError_t launch_headnorm_rope_nt<T>(T*, long, long, int, int, int, int, const T*, long, float, int, int,         \
                                                 const float2*, int, int, hipStream_t);                                       \
  template hipError_t launch_headnorm_rope<T>(T*, long, int, int, int, const T*, float, int, int, const float2*, int, int,    \
                                              hipStream_t);                                                                   \
  template hipError_t launch_transpose_heads<T>(const T*, long, T*, long, long, int, int, int, int, hipStream_t);            \
  template hipError_t launch_embedding<T>(const int*, const T*, T*, long, int, int, hipStream_t);                            \
  template hipError_t launch_silu<T>(const T*, T*, long, hipStream_t);                                                       \
  template hipError_t launch_scale_2d<T>(T*, long, int, int, float, hipStream_t);                                            \
  template hipError_t launch_scale_inplace<T>(T*, long, float, hipStream_t);                                                 \
  template hipError_t launch_mod_finalize<T>(T*, long, int, hipStream_t);                                                    \
  template hipError_t launch_convert_from_f32<T>(const float*, long, T*, long, int, int, int, hipStream_t);                  \
  template hipError_t launch_convert_to_f32<T>(const T*, long, float*, long, int, int, hipStream_t);                         \
  template hipError_t launch_euler<T>(const EulerArgs&, hipStream_t);
INST(bf16_t)
INST(float)
