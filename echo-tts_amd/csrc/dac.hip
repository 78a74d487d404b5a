// Small fp32 kernels of the Fish S1-DAC decode path that are not GEMM-shaped (channels-last
// activations, x[t][c]).  The convolutions themselves run through gemm_nt (taps loop).
#include "common.h"

namespace {

// autoencoder.py:815-826: interleaved pairs, cos/sin cache (pos, HD/2, 2) (bf16-rounded values, held as fp32)
__global__ void ae_rope_kernel(float* __restrict__ x, long ldx, long rows, int S, int H, int HD, const float* __restrict__ cache) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int half = HD >> 1;
  const long n = rows * H * half;
  if (i >= n) return;
  const long row = i / (H * half);
  const int rem = (int)(i - row * (H * half));
  const int h = rem / half, j = rem - h * half;
  const int pos = (int)(row % S);
  float* p = x + row * ldx + h * HD + 2 * j;
  const float c = cache[((long)pos * half + j) * 2], s = cache[((long)pos * half + j) * 2 + 1];
  const float a = p[0], b = p[1];
  p[0] = __fsub_rn(__fmul_rn(a, c), __fmul_rn(b, s));
  p[1] = __fadd_rn(__fmul_rn(b, c), __fmul_rn(a, s));
}

// ConvNeXtBlock front half (autoencoder.py:362-364): causal depthwise conv k7 (+bias), then LayerNorm over C.
// One wave per time step; lane owns float4 chunks lane + 64j.
__global__ void __launch_bounds__(256) dwconv_ln_kernel(const float* __restrict__ x, long ldx, float* __restrict__ y, long ldy, int T, int S,
                                                        int C, const float* __restrict__ w, const float* __restrict__ b,
                                                        const float* __restrict__ lnw, const float* __restrict__ lnb, float eps) {
  const int lane = threadIdx.x & 63;
  const long t = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (t >= T) return;
  const int tl = (int)(t % S);  // position inside its own sequence (causal padding restarts per batch item)
  constexpr int MAXC = 4;       // C <= 1024
  const int nch = C >> 2;
  float v[MAXC][4];
  float s1 = 0.f;
#pragma unroll
  for (int c = 0; c < MAXC; ++c) {
    const int ch = lane + 64 * c;
    if (ch < nch) {
      const float4 bb = *(const float4*)(b + ch * 4);
      float acc[4] = {bb.x, bb.y, bb.z, bb.w};
#pragma unroll
      for (int k = 0; k < 7; ++k) {
        const int dt = k - 6;
        if (tl + dt < 0) continue;
        const float4 xv = *(const float4*)(x + (t + dt) * ldx + ch * 4);
        const float xs[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = fmaf(w[(ch * 4 + i) * 7 + k], xs[i], acc[i]);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) { v[c][i] = acc[i]; s1 += acc[i]; }
    }
  }
  const float mean = wave_sum(s1) / (float)C;
  float var = 0.f;
#pragma unroll
  for (int c = 0; c < MAXC; ++c) {
    const int ch = lane + 64 * c;
    if (ch < nch) {
#pragma unroll
      for (int i = 0; i < 4; ++i) { const float d = v[c][i] - mean; var += d * d; }
    }
  }
  const float rs = rsqrtf(wave_sum(var) / (float)C + eps);
#pragma unroll
  for (int c = 0; c < MAXC; ++c) {
    const int ch = lane + 64 * c;
    if (ch < nch) {
      const float4 g = *(const float4*)(lnw + ch * 4), be = *(const float4*)(lnb + ch * 4);
      float4 o;
      o.x = (v[c][0] - mean) * rs * g.x + be.x;
      o.y = (v[c][1] - mean) * rs * g.y + be.y;
      o.z = (v[c][2] - mean) * rs * g.z + be.z;
      o.w = (v[c][3] - mean) * rs * g.w + be.w;
      *(float4*)(y + t * ldy + ch * 4) = o;
    }
  }
}

// Decoder tail (autoencoder.py:994): causal conv k, C -> 1, then tanh.  x already has the Snake applied.
// HBM-bound (T x C fp32 read once): 8 consecutive lanes share one input row (coalesced 16-byte loads), every lane
// accumulates the k per-tap partial dot products of its columns, an 8-lane butterfly finishes them into LDS
// d[row][tap], and the outputs of the block are y[t] = tanh(bias + sum_tap d[t - (k-1) + tap][tap]).
constexpr int CO_ROWS = 128, CO_KMAX = 8, CO_NV = 4;
__global__ void __launch_bounds__(256) conv_out_tanh_kernel(const float* __restrict__ x, long ldx, float* __restrict__ y, long T, int S,
                                                            int C, int k, const float* __restrict__ w, float bias) {
  __shared__ float d[CO_ROWS][CO_KMAX];
  const int R = CO_ROWS - (k - 1);                       // outputs per block
  const long t0 = (long)blockIdx.x * R;
  const int g = threadIdx.x >> 3, j = threadIdx.x & 7;
  float4 wr[CO_KMAX][CO_NV];
#pragma unroll
  for (int kk = 0; kk < CO_KMAX; ++kk)
#pragma unroll
    for (int v = 0; v < CO_NV; ++v) {
      const int c = (j + 8 * v) * 4;
      wr[kk][v] = (kk < k && c < C) ? *(const float4*)(w + (long)kk * C + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll 1
  for (int it = 0; it < CO_ROWS / 32; ++it) {
    const int lr = it * 32 + g;
    const long t = t0 - (k - 1) + lr;
    float acc[CO_KMAX];
#pragma unroll
    for (int kk = 0; kk < CO_KMAX; ++kk) acc[kk] = 0.f;
    if (t >= 0 && t < T) {
#pragma unroll
      for (int v = 0; v < CO_NV; ++v) {
        const int c = (j + 8 * v) * 4;
        if (c < C) {
          const float4 xv = *(const float4*)(x + t * ldx + c);
#pragma unroll
          for (int kk = 0; kk < CO_KMAX; ++kk) {
            acc[kk] = fmaf(xv.x, wr[kk][v].x, acc[kk]); acc[kk] = fmaf(xv.y, wr[kk][v].y, acc[kk]);
            acc[kk] = fmaf(xv.z, wr[kk][v].z, acc[kk]); acc[kk] = fmaf(xv.w, wr[kk][v].w, acc[kk]);
          }
        }
      }
    }
#pragma unroll
    for (int kk = 0; kk < CO_KMAX; ++kk) {
      float a = acc[kk];
      a += __shfl_xor(a, 1, 64); a += __shfl_xor(a, 2, 64); a += __shfl_xor(a, 4, 64);
      if (j == 0) d[lr][kk] = a;
    }
  }
  __syncthreads();
  const int i = threadIdx.x;
  const long t = t0 + i;
  if (i < R && t < T) {
    const int tl = (int)(t % S);
    float a = bias;
    for (int kk = 0; kk < k; ++kk)
      if (tl + kk - (k - 1) >= 0) a += d[i + kk][kk];
    y[t] = tanhf(a);
  }
}

// inference.py:228 front: out[r][l] = lat[r][l] / latent_scale, zero padded to Lpad columns
__global__ void pca_prep_kernel(const float* __restrict__ lat, float* __restrict__ out, long ldo, long rows, int L, int Lpad, float scale) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * Lpad) return;
  const long r = i / Lpad;
  const int l = (int)(i - r * Lpad);
  out[r * ldo + l] = l < L ? lat[r * L + l] / scale : 0.0f;
}

// autoencoder.py:96-102
__global__ void snake_kernel(const float* __restrict__ x, long ldx, float* __restrict__ y, long ldy, long rows, int C, const float* __restrict__ alpha) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * C) return;
  const long r = i / C;
  const int c = (int)(i - r * C);
  const float a = alpha[c], v = x[r * ldx + c];
  const float sn = sinf(a * v);
  y[r * ldy + c] = v + (1.0f / (a + 1e-9f)) * (sn * sn);
}

inline dim3 grid1d(long n, int bs = 256) { return dim3((unsigned)((n + bs - 1) / bs)); }

}  // namespace

// ---------------------------------------------------------------- encode path (speaker reference -> latents)
// Encoder head (autoencoder.py:917): causal conv k, 1 -> C on the raw audio; writes y[t][c] and the Snake of it with
// the first ResidualUnit's alpha (the next GEMM's input).  One thread = 4 channels of one sample; HBM-bound on the stores.
__global__ void __launch_bounds__(256) conv_in_snake_kernel(const float* __restrict__ x, long T, int C, int k, const float* __restrict__ w /*(C,k)*/,
                                                            const float* __restrict__ b, const float* __restrict__ alpha,
                                                            float* __restrict__ y, float* __restrict__ s, long ld) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int cpr = C / 4;
  const long t = i / cpr;
  if (t >= T) return;
  const int c = (int)(i - t * cpr) * 4;
  float acc[4] = {b[c], b[c + 1], b[c + 2], b[c + 3]};
  for (int kk = 0; kk < k; ++kk) {
    const long ts = t - (k - 1) + kk;
    const float xv = ts >= 0 ? x[ts] : 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[e] = fmaf(w[(long)(c + e) * k + kk], xv, acc[e]);
  }
  float sn[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float a = alpha[c + e], q = sinf(a * acc[e]);
    sn[e] = acc[e] + (1.0f / (a + 1e-9f)) * (q * q);
  }
  *(float4*)(y + t * ld + c) = make_float4(acc[0], acc[1], acc[2], acc[3]);
  *(float4*)(s + t * ld + c) = make_float4(sn[0], sn[1], sn[2], sn[3]);
}

// VectorQuantize.decode_latents (autoencoder.py:145-158) for codebook_dim 8: one wave per row.  e (T, 8) projected
// latents; cbn (size, 8) L2-normalised codebook, cb the raw one.  idx = first argmax of -(|e^|^2 - 2 e^.c^ + |c^|^2);
// writes the code, the straight-through value e + (cb[idx] - e) into zst[t][0..7] (row pitch ld_zst, the out_proj GEMM
// operand) and the raw code vector into gath[t][0..7] (row pitch ld_g, the from_codes GEMM operand).
__global__ void __launch_bounds__(256) vq_argmax_kernel(const float* __restrict__ e, long lde, int T, const float* __restrict__ cbn,
                                                        const float* __restrict__ cb, int size, int* __restrict__ idx_out,
                                                        float* __restrict__ zst, long ld_zst, float* __restrict__ gath, long ld_g) {
  const int lane = threadIdx.x & 63;
  const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (t >= T) return;
  float ev[8], en[8];
  float n2 = 0.f;
#pragma unroll
  for (int d = 0; d < 8; ++d) { ev[d] = e[(long)t * lde + d]; n2 += ev[d] * ev[d]; }
  const float inv = 1.0f / fmaxf(sqrtf(n2), 1e-12f);
  float e2 = 0.f;
#pragma unroll
  for (int d = 0; d < 8; ++d) { en[d] = ev[d] * inv; e2 += en[d] * en[d]; }
  float best = -INFINITY;
  int bi = 0x7fffffff;
  for (int j = lane; j < size; j += 64) {
    const float4 c0 = *(const float4*)(cbn + (long)j * 8), c1 = *(const float4*)(cbn + (long)j * 8 + 4);
    const float cv[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
    float dot = 0.f, c2 = 0.f;
#pragma unroll
    for (int d = 0; d < 8; ++d) { dot = fmaf(en[d], cv[d], dot); c2 = fmaf(cv[d], cv[d], c2); }
    const float score = -((e2 - 2.0f * dot) + c2);
    if (score > best) { best = score; bi = j; }      // strided scan keeps the lowest index of equal scores per lane
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ob = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
  }
  if (lane < 8) {
    const float cvr = cb[(long)bi * 8 + lane];
    const float evl = e[(long)t * lde + lane];
    zst[(long)t * ld_zst + lane] = evl + (cvr - evl);
    gath[(long)t * ld_g + lane] = cvr;
  }
  if (lane == 0) idx_out[t] = bi;
}

hipError_t launch_conv_in_snake(const float* x, long T, int C, int k, const float* w, const float* b, const float* alpha, float* y,
                                float* s, long ld, hipStream_t st) {
  if (C % 4 || (ld & 3)) return hipErrorInvalidValue;
  const long n = T * (C / 4);
  hipLaunchKernelGGL(conv_in_snake_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, T, C, k, w, b, alpha, y, s, ld);
  return hipGetLastError();
}
hipError_t launch_vq_argmax(const float* e, long lde, int T, const float* cbn, const float* cb, int size, int* idx, float* zst,
                            long ld_zst, float* gath, long ld_g, hipStream_t st) {
  hipLaunchKernelGGL(vq_argmax_kernel, dim3((T + 3) / 4), dim3(256), 0, st, e, lde, T, cbn, cb, size, idx, zst, ld_zst, gath, ld_g);
  return hipGetLastError();
}

hipError_t launch_ae_rope(float* x, long ldx, int rows, int S, int H, int HD, const float* cache, hipStream_t st) {
  const long n = (long)rows * H * (HD / 2);
  hipLaunchKernelGGL(ae_rope_kernel, grid1d(n), dim3(256), 0, st, x, ldx, (long)rows, S, H, HD, cache);
  return hipGetLastError();
}
hipError_t launch_dwconv_ln(const float* x, long ldx, float* y, long ldy, int T, int S, int C, const float* w, const float* b,
                            const float* lnw, const float* lnb, float eps, hipStream_t st) {
  if (C % 4 || C > 1024) return hipErrorInvalidValue;
  hipLaunchKernelGGL(dwconv_ln_kernel, dim3((T + 3) / 4), dim3(256), 0, st, x, ldx, y, ldy, T, S, C, w, b, lnw, lnb, eps);
  return hipGetLastError();
}
hipError_t launch_conv_out_tanh(const float* x, long ldx, float* y, long T, int S, int C, int k, const float* w, float bias,
                                hipStream_t st) {
  if (C % 4 || C > 32 * CO_NV || k < 1 || k > CO_KMAX || (ldx & 3)) return hipErrorInvalidValue;
  const int R = CO_ROWS - (k - 1);
  hipLaunchKernelGGL(conv_out_tanh_kernel, dim3((unsigned)((T + R - 1) / R)), dim3(256), 0, st, x, ldx, y, T, S, C, k, w, bias);
  return hipGetLastError();
}
hipError_t launch_pca_prep(const float* lat, float* out, long ldo, long rows, int L, int Lpad, float scale, hipStream_t st) {
  hipLaunchKernelGGL(pca_prep_kernel, grid1d(rows * Lpad), dim3(256), 0, st, lat, out, ldo, rows, L, Lpad, scale);
  return hipGetLastError();
}
hipError_t launch_snake_f32(const float* x, long ldx, float* y, long ldy, long rows, int C, const float* alpha, hipStream_t st) {
  hipLaunchKernelGGL(snake_kernel, grid1d(rows * C), dim3(256), 0, st, x, ldx, y, ldy, rows, C, alpha);
  return hipGetLastError();
}
