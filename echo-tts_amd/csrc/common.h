// Shared device/host helpers for libechohip (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <atomic>
#include <string>

// hipFuncSetAttribute acts on the current device's copy of a kernel: `done` remembers (one bit per device ordinal) where
// the dynamic-LDS limit of `kern` has been raised, so that a process driving several GPUs prepares each of them.
inline hipError_t ensure_dyn_lds(const void* kern, int bytes, std::atomic<unsigned long long>& done) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  const unsigned long long bit = 1ull << (dev & 63);
  if (done.load(std::memory_order_acquire) & bit) return hipSuccess;
  e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e == hipSuccess) done.fetch_or(bit, std::memory_order_release);
  return e;
}

typedef uint16_t bf16_t;  // raw bfloat16 bits
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define ECHO_OK 0
#define ECHO_ERR 1

// ---------------------------------------------------------------- numeric traits
__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) {
  // plain cast = v_cvt_pk_bf16_f32: round-to-nearest-even, NaN stays NaN (MI355X_MICROARCH.md "Correctness boundaries")
  return __builtin_bit_cast(unsigned short, (__bf16)f);
}
typedef __attribute__((ext_vector_type(8))) float f32x8;
typedef __attribute__((ext_vector_type(8))) __bf16 hbf16x8;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 hbf16x2;
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  f32x2 v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, hbf16x2));
}

template <typename T> struct Num;
template <> struct Num<bf16_t> {
  static constexpr bool is_bf16 = true;
  __device__ __forceinline__ static float ld(bf16_t v) { return bf2f(v); }
  __device__ __forceinline__ static bf16_t st(float f) { return f2bf(f); }
  __device__ __forceinline__ static float rnd(float f) { return bf2f(f2bf(f)); }
};
template <> struct Num<float> {
  static constexpr bool is_bf16 = false;
  __device__ __forceinline__ static float ld(float v) { return v; }
  __device__ __forceinline__ static float st(float f) { return f; }
  __device__ __forceinline__ static float rnd(float f) { return f; }
};

// 4 consecutive elements of T <-> 4 floats
template <typename T> struct Vec4;
template <> struct Vec4<bf16_t> {
  typedef uint2 raw;
  __device__ __forceinline__ static void unpack(raw r, float* f) {
    f[0] = __uint_as_float(r.x << 16); f[1] = __uint_as_float(r.x & 0xffff0000u);
    f[2] = __uint_as_float(r.y << 16); f[3] = __uint_as_float(r.y & 0xffff0000u);
  }
  __device__ __forceinline__ static raw pack(const float* f) {
    raw r;
    r.x = pack_bf16x2(f[0], f[1]);
    r.y = pack_bf16x2(f[2], f[3]);
    return r;
  }
};
template <> struct Vec4<float> {
  typedef float4 raw;
  __device__ __forceinline__ static void unpack(raw r, float* f) { f[0] = r.x; f[1] = r.y; f[2] = r.z; f[3] = r.w; }
  __device__ __forceinline__ static raw pack(const float* f) { return make_float4(f[0], f[1], f[2], f[3]); }
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + expf(-x)); }
// bf16 outputs only: v_exp_f32 + v_rcp_f32 (~2 ulp of fp32, far below the bf16 rounding that follows) instead of the ~25-instruction
// expf + IEEE division; the fp32 parity engine keeps silu_f
__device__ __forceinline__ float silu_fast(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + expf(-x)); }
// sin(x) on the hardware transcendental: v_sin_f32 takes revolutions, v_fract_f32 brings them into its domain.  Absolute error ~1e-6 for the
// |x| < ~100 the Snake activations see (the revolutions lose log2(|x| / 2 pi) bits to the fraction); sinf() costs ~38 VALU instructions here, this 3.
__device__ __forceinline__ float sin_fast(float x) { return __builtin_amdgcn_sinf(__builtin_amdgcn_fractf(x * 0.15915494309189532f)); }
__device__ __forceinline__ float sigmoid_fast(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }   // bf16 outputs only (see silu_fast)
__device__ __forceinline__ float gelu_erf_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

// ---------------------------------------------------------------- GEMM (gemm.hip)
// C[m][n] = epilogue( sum_{tap,k} A[m + tap_base + tap*tap_shift][k] * W[n][tap*K + k] )
// A rows and W rows are K-contiguous ("NT").  See DESIGN.md "Kernels / gemm_nt".
struct GemmArgs {
  const void* A; const void* W; void* C; void* C2;
  int M, N, K, Npad;            // K per tap (multiple of 128 bytes / sizeof(T)); Npad = rows of W (multiple of 128)
  long lda, ldw, ldc;           // in elements
  int taps, tap_base, tap_shift;
  int nbatch, nbi;              // blockIdx.y = z -> (zo = z / nbi, zi = z % nbi)
  long a_bo, a_bi, w_bo, w_bi, c_bo, c_bi;
  float acc_scale;              // y = acc * acc_scale (1.0f = none)
  const void* bias; long bias_bo, bias_bi; int vec_mod;   // per-column vectors are indexed n % vec_mod (0 = n)
  float div;                    // != 0: y = rnd(y / div)
  int act;                      // 0 none, 1 silu, 2 gelu(erf)
  const void* colscale;         // y = rnd(y * colscale[n])
  const void* res; long ldres, res_bo, res_bi;            // y = rnd(y + res[m][n])
  const void* snake_alpha;      // C2[m][n] = rnd(snake(y, alpha[n]))
  int store_main;               // write y to C
  int swiglu;                   // W rows interleaved [16 x w1 | 16 x w3]; C has N/2 columns
  int cfg;                      // tile configuration (gemm.hip TileCfg table), 0 = 128x128x2 stages
  int ksplit;                   // > 1: split K over blockIdx.z, fp32 partial slabs in ws, reduce kernel applies the tail
  void* ws; long ws_bytes;
  int split3;                   // fp32 only: bf16 hi/lo operand splitting, 3 bf16 MFMAs per product (gemm.hip SPLIT3)
  int w_presplit;               // split3 only: W was reformatted by launch_presplit_w ([32 hi | 32 lo] bf16 per 32-float block)
  // fp8 operands with a static (calibrated) activation scale: a_scale == nullptr -> every A row has the scale a_scale_const;
  // SwiGLU tail of the fp8 ping-pong kernel with c8 != nullptr: the output leaves as e4m3(bf16(out) * c8_inv) bytes at c8 (pitch c8_ld
  // bytes) instead of bf16 at C - the A operand of w2 without a quantisation pass
  float a_scale_const;
  void* c8; long c8_ld; float c8_inv;
  int pp_gn;                    // gemm_pp_kernel: tile columns per strip of the tile order (0 = default PP_GN)
  int snake_fast;               // conv tails (NTAIL >= 2): Snake with sin_fast() instead of sinf() (engine: ECHO_DAC_FAST_SIN)
  void* sink;                   // optional: >= 256 KiB of device scratch nobody reads; the branch-free conv tails of gemm_nt_kernel (NTAIL >= 2) send the
                                // stores of their idle threads there instead of predicating them (engine: one per context)
  // fused QKV(G) epilogue (model.py:217-232 / 132-142): the N axis is [q | k | v | gate] x qkv_D.  q and k sections get the
  // per-head RMSNorm (weights qk_w = [q_norm | k_norm], each qkv_D) and interleaved-pair RoPE on heads < rope_heads at
  // position pos0 + (m % qkv_S); the v section is written TRANSPOSED to vt[(m / S)][h][d][m % S]; gate is stored as is.
  int qkv_mode, qkv_D, qkv_S, rope_heads, pos0;
  int qkv_gate_act;             // bf16 only: the gate section is stored as bf16(sigmoid_fast(bf16(acc))) - what the attention epilogue would
                                // compute from it (AttnArgs.g_act = 1 then takes it as the multiplier): same bits, the exponentials ride in the GEMM tail
  float qk_eps;
  const void* qk_w; const void* rope;
  void* vt; long vt_ld, vt_row_stride;
  // fp8 operands (gemm_pp_kernel only, activations/outputs stay bf16): A and W hold OCP e4m3 bytes, one fp32 scale per A row
  // and per W row; y = acc * a_scale[m] * w_scale[n] before the tail.  K is then a multiple of 128.
  int fp8; const float* a_scale; const float* w_scale;
};
int gemm_tile_m(int cfg);
int gemm_num_cfgs();
void gemm_args_init(GemmArgs* g);
template <typename T> hipError_t launch_gemm_nt(const GemmArgs& g, hipStream_t st);

// ---------------------------------------------------------------- attention (attention.hip)
struct AttnSeg {
  const bf16_t* K; long k_ld, k_row_stride, k_head_stride;
  const bf16_t* Vt; long vt_ld, vt_row_stride, vt_head_stride;
  const int* nkeys;             // [rows] number of leading keys that may be attended (device)
  const float* bias; long bias_row_stride;  // optional additive bias (0 / -inf) per key, per row
  int kv_mod;                   // kv batch index = kv_mod ? row % kv_mod : row
};
struct AttnArgs {
  const bf16_t* Q; long q_ld, q_row_stride;
  bf16_t* O; long o_ld, o_row_stride;
  const bf16_t* G; long g_ld, g_row_stride;   // optional pre-sigmoid gate, same indexing as O
  int g_act;                    // 1: G already holds bf16(sigmoid(gate)) (GemmArgs.qkv_gate_act): the epilogue only multiplies
  int S, H, rows;
  int nseg; AttnSeg seg[4];
  int causal;                   // only with nseg == 1 (encoder self-attention)
  float scale;
  void* prof;                   // diagnostic: per-wave {barrier, issue, compute, tiles} cycle sums (s_memtime); nullptr in production
  // attn4_kernel (the fast joint-attention kernel) writes one word per 256-query workgroup into `redo` [rows][H][ceil(S / 256)]:
  // non-zero = a score left its fixed-reference range and attn_kernel must redo that block; nullptr = attn_kernel only.
  int* redo;
  const int* redo_filter;       // set by the launcher on attn_kernel's second pass
  int q_block0;                 // set by the launcher: attn_kernel's first 128-query block (blockIdx.x counts from it)
  int redo_nb;                  // set by the launcher: blocks per (row, head) in `redo`
  int q128;                     // set by the launcher: attn5_kernel runs every block as 128 queries (one stream per wave), `redo` is per 128 queries  // fp8 engine with static (calibrated) activation scales: O8 != nullptr -> the epilogue writes e4m3(bf16(out) * o8_inv) bytes there
  // (same indexing as O, pitches in bytes) INSTEAD of the bf16 output: the A operand of the wo GEMM without a quantisation pass
  uint8_t* O8; long o8_ld, o8_row_stride; float o8_inv;
};
inline long attn_redo_words(int rows, int H, int S) { return (long)rows * H * ((S + 127) / 128); }      // enough for the 128-query block mode
hipError_t launch_attention_bf16(const AttnArgs& a, hipStream_t st);

// ---------------------------------------------------------------- elementwise (elementwise.hip)
enum NormMode { NORM_ADALN = 0, NORM_RMS_W = 1, NORM_AE_RMS = 2, NORM_LAYER = 3 };
template <typename T>
hipError_t launch_norm(int mode, const T* x, long ldx, T* y, long ldy, int rows, int D, float eps,
                       const T* w0, const T* w1, hipStream_t st);
// per-head RMSNorm(weight (H,HD)) + interleaved-pair RoPE on heads < rope_heads, in place, HD = 128
template <typename T>
hipError_t launch_headnorm_rope(T* x, long ldx, int rows, int S, int H, const T* w, float eps, int do_norm,
                                int rope_heads, const float2* rope, int pos0, int pos_mul, hipStream_t st);
// Vt[b][h][d][s] = V[b*S + s][h*HD + d]
template <typename T>
hipError_t launch_transpose_heads(const T* v, long ldv, T* vt, long vt_ld, long vt_b_stride, int B, int S, int H,
                                  int HD, hipStream_t st);
template <typename T>
hipError_t launch_headnorm_rope_nt(T* x, long ldx, long t_stride, int nt, int rows, int S, int H, const T* w, long w_stride,
                                   float eps, int do_norm, int rope_heads, const float2* rope, int pos0, int pos_mul,
                                   hipStream_t st);
template <typename T> hipError_t launch_embedding(const int* ids, const T* table, T* out, long ldo, int n, int D, hipStream_t st);
template <typename T> hipError_t launch_silu(const T* x, T* y, long n, hipStream_t st);
// bf16 rows -> OCP e4m3 bytes + per-row fp32 scale (amax / 448)
hipError_t launch_quant_rows_fp8(const void* x, long ldx, void* q, long ldq, float* scale, int rows, int K, hipStream_t st);
// *dst = max(*dst, max_i v[i]) for non-negative finite v (fp8 calibration: the largest row scale a block has seen)
hipError_t launch_max_into(const float* v, int n, float* dst, hipStream_t st);
// bf16 AdaLN-apply norm (norm_kernel<bf16, NORM_ADALN>) whose output leaves as e4m3 bytes + per-row scale
hipError_t launch_norm_adaln_fp8(const void* x, long ldx, void* q, long ldq, float* scale, int rows, int D, float eps, const void* scale1p,
                                 const void* shift, hipStream_t st);
template <typename T> hipError_t launch_scale_inplace(T* x, long n, float s, hipStream_t st);
template <typename T> hipError_t launch_scale_2d(T* x, long ld, int rows, int cols, float s, hipStream_t st);
template <typename T> hipError_t launch_mod_finalize(T* mod, long rows, int D, hipStream_t st);
template <typename T> hipError_t launch_convert_from_f32(const float* x, long ldx, T* y, long ldy, int rows, int cols, int cols_pad, hipStream_t st);
template <typename T> hipError_t launch_convert_to_f32(const T* x, long ldx, float* y, long ldy, int rows, int cols, hipStream_t st);
hipError_t launch_convert_any(const void* src, int src_dtype, void* dst, int dst_dtype, long n, hipStream_t st);
// packed row copy with dtype conversion: dst[map(r)][c] = src[r][c]
hipError_t launch_pack_rows(const void* src, int src_dtype, long src_ld, void* dst, int dst_dtype, long dst_ld,
                            int rows, int cols, int dst_row0, int swiglu_half /*-1 none, 0 w1, 1 w3*/, hipStream_t st);

struct EulerArgs {
  float* x;            // (B, S, L) fp32 state, updated in place
  const void* v;       // model output, (R*B*S, ldv) of T, rows ordered [cond | uncond_text | uncond_speaker]
  long ldv;
  void* xin;           // (R*B*S, ld_xin) of T: next model input (all R copies), zero padded columns
  long ld_xin;
  int B, S, L, R;      // R = 3 (cfg) or 1;  R_next = rows to write into xin
  int R_next;
  float s_text, s_spk;
  int rescale; float r_inv1mt, r_ratio, r_1mt;
  float dt;            // t_next - t
  int init;            // 1: x = x * init_scale and no model step (truncation_factor, inference.py:478-479; 0.0 is a legal factor) + first model input
  float init_scale;
};
template <typename T> hipError_t launch_euler(const EulerArgs& e, hipStream_t st);

// fp32 softmax over rows of a score matrix with per-key bias and optional causal window
hipError_t launch_softmax_f32(float* s, long ld, int rows_per_batch, int nbatch, int ncols, int ncols_pad,
                              const float* bias, long bias_batch_stride, int heads_per_bias_row,
                              int causal, int window, hipStream_t st);
hipError_t launch_mask_to_bias(const uint8_t* mask, float* bias, long n, hipStream_t st);

// ---------------------------------------------------------------- post-processing (postproc.hip)
#define ECHO_MAX_CHUNKS 64
// out[b] = first frame of lat[b] (T, W) whose zero-extended window is flat (inference.py:288-296), T if none
hipError_t launch_flatten_point(const float* lat, long item_stride, int B, int T, int W, int window, float target, float std_thr, int* out,
                                hipStream_t st);
// out[c] = trailing samples with |x| < thr among the last min(len, max_window) samples of chunk c (handler.py:199-211); host tables
hipError_t launch_trailing_quiet(const float* const* chunks, const long* lens, int n, int max_window, float thr, int* out, hipStream_t st);
// trimmed / zero-padded chunks cross-faded into one waveform (handler.py:126-170, 213-232); host tables
hipError_t launch_assemble_chunks(const float* const* src, const long* start, const long* len, const long* valid, const int* ov, int n,
                                  float* out, long total, hipStream_t st);

// out[f * up + p] = sum_k bank[p][k] * x[f * down + k - width] (zero outside [0, n)): polyphase sinc resampling, bank (up, taps) built by the host
hipError_t launch_resample(const float* x, long n, const float* bank, int taps, int up, int down, int width, float* out, long n_out, hipStream_t st);

// ---------------------------------------------------------------- DAC helpers (dac.hip)
// All take channels-last fp32 activations x[t][c]; S = time steps per batch item (causal padding restarts there).
hipError_t launch_ae_rope(float* x, long ldx, int rows, int S, int H, int HD, const float* cache /*(pos,HD/2,2)*/, hipStream_t st);
hipError_t launch_dwconv_ln(const float* x, long ldx, float* y, long ldy, int T, int S, int C, const float* w /*(C,7)*/,
                            const float* b, const float* lnw, const float* lnb, float eps, hipStream_t st);
hipError_t launch_conv_out_tanh(const float* x, long ldx, float* y, long T, int S, int C, int k, const float* w /*(k,C)*/,
                                float bias, hipStream_t st);
hipError_t launch_pca_prep(const float* lat, float* out, long ldo, long rows, int L, int Lpad, float scale, hipStream_t st);
hipError_t launch_presplit_w(float* w, long rows, long ld, hipStream_t st);
hipError_t launch_snake_f32(const float* x, long ldx, float* y, long ldy, long rows, int C, const float* alpha, hipStream_t st);
// encode path
hipError_t launch_conv_in_snake(const float* x, long T, int C, int k, const float* w /*(C,k)*/, const float* b, const float* alpha,
                                float* y, float* s, long ld, hipStream_t st);
hipError_t launch_vq_argmax(const float* e, long lde, int T, const float* cbn, const float* cb, int size, int* idx, float* zst,
                            long ld_zst, float* gath, long ld_g, hipStream_t st);
