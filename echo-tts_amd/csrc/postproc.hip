// On-device post-processing of generated latents / waveforms.  The reference does these steps with Python
// loops and one host sync per iteration: find_flattening_point (inference.py:288-296: <= 640 iterations x 2 syncs),
// normalize_chunk_boundaries (handler.py:173-240: a per-sample Python loop over up to 44 100 samples) and crossfade_chunks
// (handler.py:126-170).  Here each is one small launch; the host reads back one int per utterance / chunk.  All three are
// HBM-bound passes over at most a few MB (B x 640 x 80 floats; the last 44 100 samples of a chunk; the output waveform).
#include "common.h"

namespace {

// ---- find_flattening_point: first frame i whose window [i, i + window) of the zero-extended (T, W) latent has unbiased
// std < std_thr and |mean - target| < 0.1; T when there is none.  One workgroup per batch item.  Sums are kept in double:
// the reference evaluates torch's fp32 mean / std per window, whose rounding no independent summation reproduces; a double
// evaluation differs from it only for windows that sit on the threshold to within fp32 noise.
__global__ void __launch_bounds__(256) flatten_point_kernel(const float* __restrict__ lat, long item_stride, int T, int W, int window, float target,
                                                            float std_thr, int* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  double* fs = (double*)smem_raw;          // per-frame sum, T + window entries (the tail is the zero padding)
  double* fq = fs + (T + window);          // per-frame sum of squares
  __shared__ int first;
  const float* x = lat + (long)blockIdx.x * item_stride;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  if (tid == 0) first = T;
  for (int f = wid; f < T + window; f += 4) {
    double s = 0.0, q = 0.0;
    if (f < T) {
      for (int c = lane; c < W; c += 64) { const double v = (double)x[(long)f * W + c]; s += v; q += v * v; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o, 64); q += __shfl_xor(q, o, 64); }
    if (lane == 0) { fs[f] = s; fq[f] = q; }
  }
  __syncthreads();
  const double cnt = (double)window * W;
  for (int i = tid; i < T; i += 256) {
    double s = 0.0, q = 0.0;
    for (int j = 0; j < window; ++j) { s += fs[i + j]; q += fq[i + j]; }
    const double mean = s / cnt;
    double var = (q - s * s / cnt) / (cnt - 1.0);
    var = var > 0.0 ? var : 0.0;
    if (sqrt(var) < (double)std_thr && fabs(mean - (double)target) < 0.1) atomicMin(&first, i);
  }
  __syncthreads();
  if (tid == 0) out[blockIdx.x] = first;
}

// ---- trailing silence: number of trailing samples of the last `window` samples of chunk c whose magnitude is below thr
// (handler.py:205-211 counts them one by one from the end).  One workgroup per chunk.
struct QuietTab { const float* p[ECHO_MAX_CHUNKS]; long n[ECHO_MAX_CHUNKS]; };
__global__ void __launch_bounds__(256) trailing_quiet_kernel(const QuietTab tab, int max_window, float thr, int* __restrict__ out) {
  __shared__ int last_loud;
  const int c = blockIdx.x;
  const long n = tab.n[c];
  const int win = (int)(n < (long)max_window ? n : (long)max_window);
  if (threadIdx.x == 0) last_loud = -1;
  __syncthreads();
  const float* tail = tab.p[c] + (n - win);
  int best = -1;
  for (int i = threadIdx.x; i < win; i += 256)
    if (!(fabsf(tail[i]) < thr)) best = i;          // the reference counts x < thr as silent: NaN is loud
  if (best >= 0) atomicMax(&last_loud, best);
  __syncthreads();
  if (threadIdx.x == 0) out[c] = win - 1 - last_loud;
}

// ---- boundary normalisation + cross-fade in one pass (handler.py:126-170 applied to the trimmed / padded chunks of
// handler.py:213-232).  Chunk i contributes `len` samples starting at output position `start`: its first `valid` samples come
// from `src`, the rest are the zero padding; `ov` = samples of overlap with chunk i + 1.  An output sample is evaluated in the
// reference's order: result = result * fade_out + next * fade_in inside an overlap, next outside.
struct AsmTab {
  const float* src[ECHO_MAX_CHUNKS];
  long start[ECHO_MAX_CHUNKS], len[ECHO_MAX_CHUNKS], valid[ECHO_MAX_CHUNKS];
  int ov[ECHO_MAX_CHUNKS];
  int n;
};
// torch.linspace(a, b, steps)[k] for float32 (symmetric evaluation from both ends, ATen RangeFactories)
__device__ __forceinline__ float linspace_at(float a, float b, int steps, int k) {
  if (steps == 1) return a;
  const float step = __fdiv_rn(__fsub_rn(b, a), (float)(steps - 1));
  return k < steps / 2 ? __fadd_rn(a, __fmul_rn(step, (float)k)) : __fsub_rn(b, __fmul_rn(step, (float)(steps - k - 1)));
}
__global__ void __launch_bounds__(256) assemble_kernel(const AsmTab tab, float* __restrict__ out, long total) {
  const long o = (long)blockIdx.x * 256 + threadIdx.x;
  if (o >= total) return;
  float acc = 0.0f;
  for (int i = 0; i < tab.n; ++i) {
    const long k = o - tab.start[i];
    if (k < 0 || k >= tab.len[i]) continue;      // starts need not be monotonic: a long overlap can reach back over a short chunk
    const float v = k < tab.valid[i] ? tab.src[i][k] : 0.0f;
    const int ovp = i > 0 ? tab.ov[i - 1] : 0;
    if (k < ovp) {
      const float dn = linspace_at(1.0f, 0.0f, ovp, (int)k), up = linspace_at(0.0f, 1.0f, ovp, (int)k);
      acc = __fadd_rn(__fmul_rn(acc, dn), __fmul_rn(v, up));
    } else {
      acc = v;
    }
  }
  out[o] = acc;
}

}  // namespace

hipError_t launch_flatten_point(const float* lat, long item_stride, int B, int T, int W, int window, float target, float std_thr, int* out,
                                hipStream_t st) {
  if (B < 1 || T < 1 || W < 1 || window < 1 || (long)(T + window) * 16 > 96 * 1024) return hipErrorInvalidValue;
  hipLaunchKernelGGL(flatten_point_kernel, dim3(B), dim3(256), (size_t)(T + window) * 16, st, lat, item_stride, T, W, window, target, std_thr, out);
  return hipGetLastError();
}

hipError_t launch_trailing_quiet(const float* const* chunks, const long* lens, int n, int max_window, float thr, int* out, hipStream_t st) {
  if (n < 1 || n > ECHO_MAX_CHUNKS || max_window < 1) return hipErrorInvalidValue;
  QuietTab tab;
  memset(&tab, 0, sizeof(tab));
  for (int i = 0; i < n; ++i) { if (!chunks[i] || lens[i] < 0) return hipErrorInvalidValue; tab.p[i] = chunks[i]; tab.n[i] = lens[i]; }
  hipLaunchKernelGGL(trailing_quiet_kernel, dim3(n), dim3(256), 0, st, tab, max_window, thr, out);
  return hipGetLastError();
}

// ---- polyphase FIR resampling (inference.py:104-113 load_audio -> torchaudio.functional.resample, the sinc / Hann-window kernel bank is built by
// the host): out[f * up + p] = sum_k bank[p][k] * x[f * down + k - width], x read as zero outside [0, n).  One thread per output sample;
// a wave reads 64 * down / up consecutive inputs plus the filter span from L2: HBM-bound on at most a few MB.
__global__ void __launch_bounds__(256) resample_kernel(const float* __restrict__ x, long n, const float* __restrict__ bank, int taps, int up, int down,
                                                       int width, float* __restrict__ out, long n_out) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n_out) return;
  const long f = i / up;
  const int p = (int)(i - f * up);
  const float* b = bank + (long)p * taps;
  const long x0 = f * down - width;
  float acc = 0.f;
  for (int k = 0; k < taps; ++k) {
    const long j = x0 + k;
    if (j >= 0 && j < n) acc = __fmaf_rn(b[k], x[j], acc);
  }
  out[i] = acc;
}

hipError_t launch_resample(const float* x, long n, const float* bank, int taps, int up, int down, int width, float* out, long n_out, hipStream_t st) {
  if (!x || !bank || !out || n < 1 || n_out < 1 || taps < 1 || up < 1 || down < 1 || width < 0) return hipErrorInvalidValue;
  hipLaunchKernelGGL(resample_kernel, dim3((unsigned)((n_out + 255) / 256)), dim3(256), 0, st, x, n, bank, taps, up, down, width, out, n_out);
  return hipGetLastError();
}

hipError_t launch_assemble_chunks(const float* const* src, const long* start, const long* len, const long* valid, const int* ov, int n,
                                  float* out, long total, hipStream_t st) {
  if (n < 1 || n > ECHO_MAX_CHUNKS || total < 0) return hipErrorInvalidValue;
  if (total == 0) return hipSuccess;
  AsmTab tab;
  memset(&tab, 0, sizeof(tab));
  tab.n = n;
  for (int i = 0; i < n; ++i) {
    if (len[i] < 0 || valid[i] < 0 || valid[i] > len[i] || start[i] < 0 || start[i] + len[i] > total || (valid[i] > 0 && !src[i])) return hipErrorInvalidValue;
    tab.src[i] = src[i]; tab.start[i] = start[i]; tab.len[i] = len[i]; tab.valid[i] = valid[i]; tab.ov[i] = i + 1 < n ? ov[i] : 0;
    if (i + 1 < n && ov[i] < 0) return hipErrorInvalidValue;
  }
  hipLaunchKernelGGL(assemble_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, tab, out, total);
  return hipGetLastError();
}
