// Host-side engine of libechohip: weight packing, KV caches, the EchoDiT forward, the Euler/CFG
// sampler loop and the Fish S1-DAC decode, as static schedules of the HIP kernels in this directory.
// Everything is enqueued on the caller's stream from C++ (no Python in the step loop).
#include "common.h"
#include "../../include/echo_hip.h"

#include <dlfcn.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

namespace {

thread_local std::string g_create_error;

struct Plan { int cfg = 0, ksplit = 1; };
std::map<int, std::map<std::vector<long>, Plan>> g_plans;   // device -> (shape key -> plan), shared by every context of the process
std::mutex g_plans_mu;                                       // contexts may be driven from different host threads (ctypes drops the GIL)

inline long rup(long x, long m) { return (x + m - 1) / m * m; }

// GEMM plans persist in a text file (one "M N K taps swiglu nbatch esize qkv split3 : cfg ksplit" line per shape) so that every
// process, rank and box runs a shape with the same tile configuration and split-K factor: the summation order and the bf16
// rounding points - and with them the output bits for a given seed - no longer depend on the noisy timings of a first-use
// tuning pass.  ECHO_GEMM_PLANS names the file (default: gemm_plans.txt next to libechohip.so); shapes it does not list are
// tuned once and, when ECHO_GEMM_PLANS_SAVE names a file, appended there (how the shipped table was produced).
std::string plan_file_path() {
  if (const char* e = getenv("ECHO_GEMM_PLANS")) return e;
  Dl_info info;
  if (dladdr((const void*)&plan_file_path, &info) && info.dli_fname) {
    std::string p = info.dli_fname;
    const size_t k = p.find_last_of('/');
    return (k == std::string::npos ? std::string(".") : p.substr(0, k)) + "/gemm_plans.txt";
  }
  return "gemm_plans.txt";
}
void load_plan_file(std::map<std::vector<long>, Plan>& plans) {
  FILE* f = fopen(plan_file_path().c_str(), "r");
  if (!f) return;
  char line[512];
  while (fgets(line, sizeof(line), f)) {
    long k[9]; int cfg = 0, ks = 1;
    if (line[0] == '#') continue;
    if (sscanf(line, "%ld %ld %ld %ld %ld %ld %ld %ld %ld : %d %d", &k[0], &k[1], &k[2], &k[3], &k[4], &k[5], &k[6], &k[7], &k[8], &cfg, &ks) != 11) continue;
    if (cfg < 0 || cfg >= gemm_num_cfgs() || ks < 1 || ks > 64) continue;
    Plan p; p.cfg = cfg; p.ksplit = ks;
    plans.emplace(std::vector<long>(k, k + 9), p);
  }
  fclose(f);
}
void append_plan_file(const std::vector<long>& key, const Plan& p, float us) {
  const char* path = getenv("ECHO_GEMM_PLANS_SAVE");
  if (!path) return;
  FILE* f = fopen(path, "a");
  if (!f) return;
  for (long v : key) fprintf(f, "%ld ", v);
  fprintf(f, ": %d %d   # %.1f us\n", p.cfg, p.ksplit, us);
  fclose(f);
}

struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  hipError_t reserve(size_t bytes) {
    if (bytes <= cap) return hipSuccess;
    if (p) { hipError_t e = hipFree(p); if (e != hipSuccess) return e; p = nullptr; cap = 0; }
    bytes = (size_t)rup((long)bytes, 4096);
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) return e;
    cap = bytes;
    return hipMemset(p, 0, bytes);
  }
  void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
  template <typename U> U* as() const { return (U*)p; }
};

struct RawTensor {
  void* d = nullptr;
  int dtype = 0;
  std::vector<int64_t> shape;
  long numel = 0;
};

template <typename T> struct DT;
template <> struct DT<float> { static constexpr int code = ECHO_F32; };
template <> struct DT<bf16_t> { static constexpr int code = ECHO_BF16; };

// combined key bias for the fp32 (parity-mode) attention: columns are the concatenated segments
__global__ void build_bias_kernel(float* __restrict__ dst, long ld, int rows, int nseg, const int* __restrict__ seg_off,
                                  const int* __restrict__ seg_w, const int* __restrict__ nkeys /*[4][maxrows]*/, int maxrows,
                                  const float* b2, long b2_stride, int b2_mod, const float* b3, long b3_stride, int b3_mod, int ncols) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)rows * ncols) return;
  const int r = (int)(i / ncols), c = (int)(i - (long)r * ncols);
  float v = -INFINITY;
  for (int s = 0; s < nseg; ++s) {
    const int k = c - seg_off[s];
    if (k >= 0 && k < seg_w[s]) {
      if (k < nkeys[s * maxrows + r]) {
        v = 0.0f;
        if (s == 2 && b2) v = b2[(long)(b2_mod ? r % b2_mod : r) * b2_stride + k];
        if (s == 3 && b3) v = b3[(long)(b3_mod ? r % b3_mod : r) * b3_stride + k];
      }
      break;
    }
  }
  dst[(long)r * ld + c] = v;
}

// test instrument (echo_debug_corrupt_tile): C[r][c] = -C[r][c] on one tile
template <typename T>
__global__ void negate_tile_kernel(T* __restrict__ c, long ldc, int r0, int nr, int c0, int nc) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)nr * nc) return;
  const long r = r0 + i / nc;
  const int col = c0 + (int)(i % nc);
  c[r * ldc + col] = Num<T>::st(-Num<T>::ld(c[r * ldc + col]));
}

// y = T(T(o) * T(sigmoid(g)))  (model.py:157,264) for the unfused fp32 path
template <typename T>
__global__ void gate_mul_kernel(T* __restrict__ o, long ldo, const T* __restrict__ g, long ldg, long rows, int cols) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * cols) return;
  const long r = i / cols;
  const int c = (int)(i - r * cols);
  const float ov = Num<T>::ld(o[r * ldo + c]);
  const float sg = Num<T>::rnd(sigmoid_f(Num<T>::ld(g[r * ldg + c])));
  o[r * ldo + c] = Num<T>::st(ov * sg);
}

// A reference voice as the sampler needs it: the speaker encoder's output projected to the K / V (and Vᵀ) of every EchoDiT
// layer (model.py:615-621), captured from one context and bound to any context of the same model on the same device.
struct VoiceSnap {
  DevBuf kv, vt, bias;
  int device = 0, precision = 0, model_size = 0, num_layers = 0;
  int B = 0, T = 0, pad = 0, vld = 0;
  std::vector<int> nk;
  bool has_bias = false; long bias_ld = 0;
  size_t kv_bytes = 0, vt_bytes = 0, bias_bytes = 0;
  ~VoiceSnap() { kv.release(); vt.release(); bias.release(); }
};

struct EngineBase {
  echo_config cfg{};
  int device = 0;
  std::string err;
  std::map<std::string, RawTensor> raw;
  bool dit_ready = false, dac_ready = false;
  bool profiling = false;
  echo_profile prof{};
  std::vector<std::pair<hipEvent_t, hipEvent_t>> gemm_events;
  std::vector<double> gemm_event_flops;   // > 0: a gemm_pp_kernel launch (plan cfg 5) with that many algorithmic FLOPs
  std::vector<std::vector<long>> gemm_event_shape;   // ECHO_PROFILE_SHAPES=1: {M, N, K, taps, swiglu, qkv_mode, fp8, cfg, ksplit} per launch
  size_t gemm_events_used = 0;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> attn_events;
  size_t attn_events_used = 0;
  hipEvent_t ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};

  virtual ~EngineBase() {
    for (auto& kv : raw) if (kv.second.d) (void)hipFree(kv.second.d);
    for (auto& e : gemm_events) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
    for (auto& e : attn_events) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
    for (auto& e : ev) if (e) (void)hipEventDestroy(e);
  }
  int fail(const std::string& m) { err = m; return ECHO_ERR; }
  int fail(hipError_t e, const char* what) {
    err = std::string(what) + ": " + hipGetErrorString(e);
    return ECHO_ERR;
  }
  virtual int finalize_dit(hipStream_t st) = 0;
  virtual int finalize_dac(hipStream_t st) = 0;
  virtual int set_rope(const void* t, int npos) = 0;
  virtual int set_ae_rope(const void* t, int npos) = 0;
  virtual int encode_text(const int32_t* ids, const float* bias, const int32_t* nk, int B, int Tt, hipStream_t st) = 0;
  virtual int encode_speaker(const void* lat, const float* bias, const int32_t* nk, int B, int Ts, hipStream_t st) = 0;
  virtual int encode_latent(const void* lat, int B, int n, long row_stride, hipStream_t st) = 0;
  virtual int scale_speaker_kv(float s, int max_layers, hipStream_t st) = 0;
  virtual int dit_forward(const void* x, const void* temb, int n_t, const int32_t* row_t, int rows, int B, int S, int start_pos, int use_latent,
                          const int32_t* ton, const int32_t* son, float* v, hipStream_t st) = 0;
  virtual int sample_euler(const echo_sampler_params* p, const float* x0, float* out, hipStream_t st) = 0;
  virtual int dac_decode(const float* lat, int T, float scale, float* wav, hipStream_t st) = 0;
  virtual int dac_decode_zq(const float* z, int T, float* wav, hipStream_t st) = 0;
  virtual int dac_decode_batch(const float* lat, int B, int T, float scale, float* wav, long wav_stride, hipStream_t st) = 0;
  virtual int set_pca(const float* w, const float* mean, int on_device, hipStream_t st) = 0;
  virtual int finalize_dac_encoder(hipStream_t st) = 0;
  virtual int dac_encode(const float* audio, long n, float* lat, int32_t* codes, float* zq, hipStream_t st) = 0;
  virtual int set_pca_encode(const float* w, const float* bias, float scale, int on_device, hipStream_t st) = 0;
  virtual int debug_get_kv(int which, int layer, float* k, float* v, int* B, int* T) = 0;
  int corrupt_countdown = 0;      // echo_debug_corrupt_tile
  virtual int voice_capture(struct VoiceSnap** out, hipStream_t st) = 0;
  virtual int voice_bind(const struct VoiceSnap* v, hipStream_t st) = 0;
  virtual int dac_decode_tail(const float* lat, int T, int f0, float scale, float* wav, hipStream_t st) = 0;
  virtual size_t workspace_bytes() = 0;
  virtual int reserve_workspace(int B, int S, int Tt, int Ts, int T_dac) = 0;
  virtual int fp8_calibrate(int on) = 0;
  virtual int fp8_calibration(float* out, int n) = 0;
  virtual int fp8_set_static(const float* s, int n) = 0;
};

#define CK(x)                                            \
  do {                                                   \
    hipError_t e__ = (x);                                \
    if (e__ != hipSuccess) return this->fail(e__, #x);   \
  } while (0)
#define CKI(x)                          \
  do {                                  \
    int r__ = (x);                      \
    if (r__ != ECHO_OK) return r__;     \
  } while (0)

template <typename T>
struct Engine : EngineBase {
  static constexpr int KE = 128 / (int)sizeof(T);   // GEMM K granularity in elements
  std::vector<void*> owned;                           // packed weights

  ~Engine() override {
    if (nk_pinned) (void)hipHostFree(nk_pinned);
    for (auto& e : nk_ev) if (e) (void)hipEventDestroy(e);
    for (void* p : owned) (void)hipFree(p);
    for (DevBuf* b : all_bufs()) b->release();
    b_gemm_ws.release(); b_tune_c.release(); b_flush.release(); b_sink.release();
  }

  // ------------------------------------------------------------------ weights
  struct EncW {
    int d = 0, H = 0, F = 0, L = 0;
    std::vector<T*> wqkvg, wo, w13, w2, an, mn;
    T* qkn = nullptr;   // [L][2][d]
    T* in_w = nullptr; T* in_b = nullptr; int in_k = 0;
    T* out_norm = nullptr;
    T* kv_w = nullptr;  // [Ldit*2*D][d]
  };
  EncW tenc, senc, lenc;
  T* text_emb = nullptr;
  T *cond0 = nullptr, *cond2 = nullptr, *cond4 = nullptr, *in_w = nullptr, *in_b = nullptr;
  std::vector<T*> wqkvg, wo, w13, w2;
  T *qkn = nullptr, *mod_down = nullptr, *mod_up = nullptr, *mod_up_b = nullptr, *out_norm = nullptr, *out_w = nullptr,
    *out_b = nullptr;
  int rank_pad = 0, lat_pad = 0;
  const float2* rope = nullptr; int rope_npos = 0;
  const float* ae_rope = nullptr; int ae_rope_npos = 0;

  // ------------------------------------------------------------------ caches and workspaces
  DevBuf b_kv_text, b_vt_text, b_kv_spk, b_vt_spk, b_kv_lat, b_vt_lat, b_nkeys, b_bias_text, b_bias_spk;
  DevBuf b_xin, b_x, b_xn, b_qkvg, b_vt_self, b_attn, b_h, b_vout, b_scores, b_biasrows, b_segmeta, b_attn_redo;
  DevBuf b_mod, b_c1, b_c2, b_cond, b_sc, b_dn, b_xstate, b_vf32;
  DevBuf b_ex, b_exn, b_eqkvg, b_evt, b_eattn, b_eh, b_ein, b_enk;
  DevBuf b_dacA, b_dacB, b_dacC, b_dq, b_dscore, b_dvt, b_dmisc;
  std::vector<DevBuf*> all_bufs() {
    return {&b_kv_text, &b_vt_text, &b_kv_spk, &b_vt_spk, &b_kv_lat, &b_vt_lat, &b_nkeys, &b_bias_text, &b_bias_spk,
            &b_xin, &b_x, &b_xn, &b_qkvg, &b_vt_self, &b_attn, &b_h, &b_vout, &b_scores, &b_biasrows, &b_segmeta, &b_attn_redo,
            &b_mod, &b_c1, &b_c2, &b_cond, &b_sc, &b_dn, &b_xstate, &b_vf32,
            &b_ex, &b_exn, &b_eqkvg, &b_evt, &b_eattn, &b_eh, &b_ein, &b_enk,
            &b_dacA, &b_dacB, &b_dacC, &b_dq, &b_dscore, &b_dvt, &b_dmisc};
  }
  // text / speaker / latent cache geometry
  int kvB = 0;          // batch size of the text cache (= utterances per sampler call)
  int spkB = 0;         // batch size of the speaker cache: kvB (one voice per utterance) or a divisor of it (1 = one voice shared by all rows)
  int text_T = 0, text_pad = 0, text_vld = 0; std::vector<int> text_nk; bool text_has_bias = false; long text_bias_ld = 0;
  int spk_T = 0, spk_pad = 0, spk_vld = 0; std::vector<int> spk_nk; bool spk_has_bias = false; long spk_bias_ld = 0;
  int lat_T = 0, lat_pad_rows = 0, lat_vld = 0, latB = 0;
  static constexpr int MAXROWS = 96;

  // ------------------------------------------------------------------ helpers
  hipError_t alloc_zero(void** p, size_t bytes) {
    hipError_t e = hipMalloc(p, bytes);
    if (e != hipSuccess) return e;
    owned.push_back(*p);
    return hipMemset(*p, 0, bytes);
  }
  int walloc(T** p, long rows, long cols) {
    CK(alloc_zero((void**)p, (size_t)rows * cols * sizeof(T)));
    return ECHO_OK;
  }
  const RawTensor* find(const std::string& n) {
    auto it = raw.find(n);
    return it == raw.end() ? nullptr : &it->second;
  }
  // copy raw (rows, cols) tensor into dst rows starting at dst_row0
  int pack(const std::string& name, T* dst, long dst_ld, int dst_row0, int rows, int cols, hipStream_t st, int swiglu_half = -1) {
    const RawTensor* r = find(name);
    if (!r) return fail("missing tensor: " + name);
    if (r->numel != (long)rows * cols) return fail("shape mismatch for " + name);
    CK(launch_pack_rows(r->d, r->dtype, cols, dst, DT<T>::code, dst_ld, rows, cols, dst_row0, swiglu_half, st));
    return ECHO_OK;
  }
  int pack_vec(const std::string& name, T* dst, long n, hipStream_t st) { return pack(name, dst, n, 0, 1, (int)n, st); }

  GemmArgs G(const T* A, long lda, const T* W, long ldw, T* C, long ldc, int M, int N, int K) {
    GemmArgs g;
    gemm_args_init(&g);
    g.A = A; g.lda = lda; g.W = W; g.ldw = ldw; g.C = C; g.ldc = ldc; g.M = M; g.N = N; g.K = K;
    g.Npad = (int)rup(N, 128);
    return g;
  }
  // ------------------------------------------------------------------ GEMM plans (tile config + split-K per shape)
  // plans are shared by every context of the process on the same device (a second engine context of a serving process
  // must not re-tune while the first one is already running kernels: its timings would be garbage)
  std::map<std::vector<long>, Plan>& plans = plans_of_device();
  static std::map<std::vector<long>, Plan>& plans_of_device() {
    std::lock_guard<std::mutex> lk(g_plans_mu);
    int d = 0; (void)hipGetDevice(&d);
    auto it = g_plans.find(d);
    if (it == g_plans.end()) {
      it = g_plans.emplace(d, std::map<std::vector<long>, Plan>()).first;
      if (!getenv("ECHO_GEMM_FORCE")) load_plan_file(it->second);
    }
    return it->second;
  }
  int device_key() { int d = 0; (void)hipGetDevice(&d); return d; }
  DevBuf b_gemm_ws, b_tune_c, b_flush;
  // ------------------------------------------------------------------ fp8 operands (config dit_fp8, bf16 engine; BASELINE C5)
  bool fp8 = false;
  std::vector<uint8_t*> q_wqkvg, q_wo, q_w13, q_w2;     // e4m3 copies of the packed block weights
  std::vector<float*> s_wqkvg, s_wo, s_w13, s_w2;       // one scale per weight row
  DevBuf b_q8, b_qs;                                    // quantised A operand of the current GEMM + its row scales
  // static (calibrated) activation scales for the two operands no producer can quantise per row - the attention output (wo's A) and the
  // SwiGLU output (w2's A): [2 l] / [2 l + 1] = amax / 448 bounds per block.  With them the attention epilogue and the SwiGLU tail write
  // e4m3 bytes themselves (AttnArgs.O8, GemmArgs.c8) and the two quant_rows_fp8 passes of a block disappear.  Calibration: the dynamic
  // path's own row scales, max-reduced per block into b_calib while `calibrating` (echo_fp8_calibrate / echo_fp8_calibration).
  std::vector<float> fp8_static;                        // empty = dynamic row scales everywhere
  bool calibrating = false;
  DevBuf b_calib, b_h8;                                 // 2 L floats (as ordered uint bits) | e4m3 SwiGLU output rows
  static bool fp8_shape_ok(long N, long K) { return K % 128 == 0 && N % 8 == 0; }
  int quant_weight(const T* w, long rows, long K, std::vector<uint8_t*>& qv, std::vector<float*>& sv, hipStream_t st) {
    uint8_t* q = nullptr; float* sc = nullptr;
    if (fp8_shape_ok(rows, K)) {
      CK(alloc_zero((void**)&q, (size_t)rows * K));
      CK(alloc_zero((void**)&sc, (size_t)rows * sizeof(float)));
      CK(launch_quant_rows_fp8(w, K, q, K, sc, (int)rows, (int)K, st));
    }
    qv.push_back(q); sv.push_back(sc);
    return ECHO_OK;
  }
  // quantises the A rows of `g` (per token row) and points the descriptor at the e4m3 operands; no-op without an fp8 weight
  int fp8_reserve(long M, long K) {
    CK(b_q8.reserve((size_t)(M + 256) * K));
    CK(b_qs.reserve((size_t)(M + 256) * sizeof(float)));
    return ECHO_OK;
  }
  // `quantised`: b_q8 / b_qs already hold the A rows (norm_adaln_fp8 wrote them)
  int to_fp8(GemmArgs& g, const uint8_t* qw, const float* sw, hipStream_t st, bool quantised = false, int calib_slot = -1) {
    if (!fp8 || !qw) return ECHO_OK;
    if (!quantised) {
      CKI(fp8_reserve(g.M, g.K));
      CK(launch_quant_rows_fp8(g.A, g.lda, b_q8.p, g.K, b_qs.as<float>(), g.M, g.K, st));
      if (calibrating && calib_slot >= 0) CK(launch_max_into(b_qs.as<float>(), g.M, b_calib.as<float>() + calib_slot, st));
    }
    g.A = b_q8.p; g.lda = g.K; g.W = qw; g.ldw = g.K; g.fp8 = 1; g.a_scale = b_qs.as<float>(); g.w_scale = sw;
    return ECHO_OK;
  }
  // Timing-based plan tuning is a TOOL, not a serving-path feature: it runs only when a plan table is being produced
  // (ECHO_GEMM_PLANS_SAVE names the file to append to) or when ECHO_GEMM_TUNE=1 asks for it.  By default a shape the shipped table
  // does not list takes the fixed rule below: no first-use stall under the global plan mutex, no 512 MiB cache flushes, and the
  // summation order - the output bits of a seed - is the same in every process and on every box.
  bool tune_enabled = getenv("ECHO_GEMM_TUNE") ? atoi(getenv("ECHO_GEMM_TUNE")) != 0 : getenv("ECHO_GEMM_PLANS_SAVE") != nullptr;

  template <typename U>
  int plan_gemm(GemmArgs& g, hipStream_t st) {
    const int KEu = 128 / (int)sizeof(U);
    const std::vector<long> key = {g.M, g.N, g.K, g.taps, g.swiglu, g.nbatch, (long)sizeof(U), g.qkv_mode, g.split3};
    std::lock_guard<std::mutex> lk(g_plans_mu);   // lookup, tuning and insertion are one critical section
    auto it = plans.find(key);
    if (it == plans.end()) {
      Plan best;
      const int nk = g.K / KEu * g.taps;
      if (const char* f = getenv("ECHO_GEMM_FORCE")) {   // "cfg,ksplit": debugging aid, applied wherever legal
        int fc = 0, fk = 1;
        if (sscanf(f, "%d,%d", &fc, &fk) == 2 && fc >= 0 && fc < gemm_num_cfgs()) {
          best.cfg = fc;
          if (fk > 1 && g.nbatch == 1 && !g.qkv_mode && nk / fk >= 1) best.ksplit = fk;
          plans.emplace(key, best);
          it = plans.find(key);
        }
      }
      if (it == plans.end()) {
      const long t128 = (long)((g.M + 127) / 128) * (g.Npad / 128);
      if (!tune_enabled) {
        // fixed rules instead of timings (what the shipped table's entries have in common): the ping-pong kernel for the big bf16
        // linears, 256-row / 192- / 96-column tiles for long-M fp32 convolutions, 128x128 otherwise - with a split-K factor that
        // brings a small grid towards one round of the 256 CUs while every split keeps at least 4 K steps
        const bool big = (long)g.M * g.Npad >= 256L * 256 * 64;
        if (sizeof(U) == 2 && big && g.nbatch == 1 && g.taps == 1 && (g.N & 7) == 0 && (g.ldc & 7) == 0 && (!g.swiglu || (g.N & 15) == 0) &&
            !g.snake_alpha && !g.C2 && g.act != 2 && g.store_main && (!g.qkv_mode || g.qkv_D % 256 == 0) && (!g.res || (g.ldres & 7) == 0) &&
            (g.vec_mod & 7) == 0) best.cfg = 5;
        else if (big && !g.qkv_mode) best.cfg = g.N % 192 == 0 && g.N % 256 != 0 ? 8 : g.N % 96 == 0 && g.N % 128 != 0 ? 9 : 2;
        else if (sizeof(U) == 2 && g.nbatch == 1 && !g.qkv_mode && t128 <= 128) {
          // bf16 engine only: the fp32 parity engine keeps one summation chain per output for every unlisted shape, so that e.g. a voice
          // encoded with batch 1 (in_proj: nbatch 1) and the same voice replicated over a batch (nbatch B) agree bit for bit
          static const int kKs[] = {8, 6, 4, 3, 2};
          for (int ks : kKs)
            if (t128 * ks <= 256 && nk / ks >= 4) { best.ksplit = ks; break; }
          if (nk >= 8) best.cfg = 1;     // deeper LDS pipeline for latency-bound small grids
        }
      } else
      if ((long)g.M * g.N * g.K * g.taps >= (1L << 26)) {
        // time every candidate on the real operands with a scratch output (the tail is irrelevant for the ranking)
        GemmArgs t = g;
        const long out_el = ((long)g.M + 256) * (g.ldc > g.Npad ? g.ldc : g.Npad);
        CK(b_tune_c.reserve((size_t)out_el * sizeof(U) * (g.nbatch > 1 ? 1 : 1)));
        t.C = b_tune_c.p; t.C2 = nullptr; t.res = nullptr; t.snake_alpha = nullptr; t.store_main = 1; t.colscale = nullptr; t.bias = nullptr; t.qkv_mode = 0;
        if (g.nbatch > 1) { t.nbatch = 1; t.nbi = 1; }
        struct EvPair {     // destroyed on every exit path (the CK() early returns below run while g_plans_mu is held)
          hipEvent_t a = nullptr, b = nullptr;
          ~EvPair() { if (a) (void)hipEventDestroy(a); if (b) (void)hipEventDestroy(b); }
        } evp;
        CK(hipEventCreate(&evp.a)); CK(hipEventCreate(&evp.b));
        const hipEvent_t e0 = evp.a, e1 = evp.b;
        CK(b_flush.reserve((size_t)512 << 20));
        float best_ms = 1e30f;
        for (int cfg = 0; cfg < gemm_num_cfgs(); ++cfg) {
          if (cfg >= 6 && (g.qkv_mode || g.N % 96 != 0)) continue;   // 192 / 96-column tiles: only where they divide N
          // split-K factors: 3 and 6 matter for the 256-CU grid (e.g. 160 tiles x 3 = 1.875 rounds instead of 0.625)
          static const int kKsplits[] = {1, 2, 3, 4, 6, 8};
          for (int ksp : kKsplits) {
            if (ksp > 1 && (g.nbatch > 1 || g.qkv_mode || nk / ksp < 4 || t128 * ksp > 4096)) continue;
            t.cfg = cfg; t.ksplit = ksp;
            if (ksp > 1) {
              const long need = (long)ksp * (((long)g.M + 767) / 768 * 768) * g.Npad * 4   /* rows padded for every tile height (128, 256, 384) */;
              CK(b_gemm_ws.reserve((size_t)need));
              t.ws = b_gemm_ws.p; t.ws_bytes = (long)b_gemm_ws.cap;
            }
            hipError_t le = launch_gemm_nt<U>(t, st);   // warm (code, TLB)
            if (le != hipSuccess) continue;
            // time cold: in the engine every weight matrix is read once per forward from HBM (3.8 GB of weights per
            // forward against a 256 MiB Infinity Cache), so the cache is flushed with a 512 MiB fill before each launch
            float ms = 0.f;
            for (int r = 0; r < 3; ++r) {
              CK(hipMemsetAsync(b_flush.p, r, b_flush.cap, st));
              CK(hipEventRecord(e0, st));
              CK(launch_gemm_nt<U>(t, st));
              CK(hipEventRecord(e1, st));
              CK(hipEventSynchronize(e1));
              float one = 0.f;
              CK(hipEventElapsedTime(&one, e0, e1));
              ms += one;
            }
            if (ms < best_ms) { best_ms = ms; best.cfg = cfg; best.ksplit = ksp; }
          }
        }
        append_plan_file(key, best, best_ms * 1e3f / 3);
        if (getenv("ECHO_GEMM_VERBOSE"))
          fprintf(stderr, "[echo] gemm plan M=%d N=%d K=%d taps=%d swiglu=%d T=%d: cfg %d ksplit %d (%.1f us)\n", g.M, g.N, g.K, g.taps,
                  g.swiglu, (int)sizeof(U), best.cfg, best.ksplit, best_ms * 1e3f / 3);
      }
      it = plans.emplace(key, best).first;
      }
    }
    g.cfg = it->second.cfg;
    g.ksplit = it->second.ksplit;
    if (g.ksplit > 1) {
      const long need = (long)g.ksplit * (((long)g.M + 767) / 768 * 768) * g.Npad * 4   /* rows padded for every tile height (128, 256, 384) */;
      CK(b_gemm_ws.reserve((size_t)need));
      g.ws = b_gemm_ws.p; g.ws_bytes = (long)b_gemm_ws.cap;
    }
    return ECHO_OK;
  }

  // A plan comes from a table keyed by (M N K taps swiglu nbatch esize qkv split3); the tail features of a launch (residual, second
  // output, Snake, activation, leading dimensions) are not part of the key, so a listed plan can be one this launch cannot take:
  // the launcher refuses it with hipErrorInvalidValue and the generic 128 x 128 tile runs instead - on every path (profiling or
  // not, bf16 / fp32 engine and the DAC's frun).
  template <typename U>
  hipError_t launch_planned(GemmArgs& g, hipStream_t st) {
    hipError_t le = launch_gemm_nt<U>(g, st);
    if (le == hipErrorInvalidValue && !g.fp8 && (g.cfg != 0 || g.ksplit > 1)) {
      g.cfg = 0; g.ksplit = 1;
      le = launch_gemm_nt<U>(g, st);
    }
    return le;
  }
  template <typename U>
  int run_planned(GemmArgs& g, hipStream_t st, bool count_pp) {
    if (profiling) {
      if (gemm_events_used == gemm_events.size()) {
        hipEvent_t a, b;
        CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
        gemm_events.emplace_back(a, b);
      }
      if (gemm_event_flops.size() < gemm_events.size()) gemm_event_flops.resize(gemm_events.size(), 0.0);
      if (gemm_event_shape.size() < gemm_events.size()) gemm_event_shape.resize(gemm_events.size());
      const size_t slot = gemm_events_used++;
      auto& e = gemm_events[slot];
      CK(hipEventRecord(e.first, st));
      CK(launch_planned<U>(g, st));
      CK(hipEventRecord(e.second, st));
      gemm_event_flops[slot] = (count_pp && g.cfg == 5) ? 2.0 * g.M * g.N * g.K * g.taps * g.nbatch : 0.0;   // after the launch: g.cfg is what ran
      gemm_event_shape[slot] = {g.M, g.N, g.K, g.taps, g.swiglu, g.qkv_mode, g.fp8, g.cfg, g.ksplit, g.nbatch};
      return ECHO_OK;
    }
    CK(launch_planned<U>(g, st));
    if (corrupt_countdown > 0 && !g.qkv_mode && !g.swiglu && !g.c8 && g.nbatch == 1 && g.store_main && g.M >= 512 && g.N >= 512 && --corrupt_countdown == 0)
      hipLaunchKernelGGL(negate_tile_kernel<U>, dim3(256), dim3(256), 0, st, (U*)g.C, g.ldc, 256, 256, 256, 256);
    return ECHO_OK;
  }

  int run(const GemmArgs& g_in, hipStream_t st) {
    GemmArgs g = g_in;
    if (g.fp8) { g.cfg = 5; g.ksplit = 1; }   // e4m3 operands exist for the ping-pong kernel only
    else CKI(plan_gemm<T>(g, st));
    return run_planned<T>(g, st, true);
  }

  // ------------------------------------------------------------------ finalize: pack EchoDiT (SURVEY.md §A.5 names)
  int pack_encoder(const std::string& pre, EncW& e, int d, int H, int F, int L, hipStream_t st) {
    e.d = d; e.H = H; e.F = F; e.L = L;
    if (d % KE || F % 64 || d / H != 128) return fail("unsupported encoder sizes");
    CKI(walloc(&e.qkn, (long)L * 2, d));
    for (int i = 0; i < L; ++i) {
      const std::string p = pre + ".blocks." + std::to_string(i);
      T *a, *b, *c, *dd, *n1, *n2;
      CKI(walloc(&a, rup(4 * d, 128), d));
      const char* names[4] = {"wq", "wk", "wv", "gate"};
      for (int j = 0; j < 4; ++j) CKI(pack(p + ".attention." + names[j] + ".weight", a, d, j * d, d, d, st));
      CKI(walloc(&b, rup(d, 128), d));
      CKI(pack(p + ".attention.wo.weight", b, d, 0, d, d, st));
      CKI(walloc(&c, rup(2 * F, 128), d));
      CKI(pack(p + ".mlp.w1.weight", c, d, 0, F, d, st, 0));
      CKI(pack(p + ".mlp.w3.weight", c, d, 0, F, d, st, 1));
      CKI(walloc(&dd, rup(d, 128), F));
      CKI(pack(p + ".mlp.w2.weight", dd, F, 0, d, F, st));
      CKI(walloc(&n1, 1, d)); CKI(pack_vec(p + ".attention_norm.weight", n1, d, st));
      CKI(walloc(&n2, 1, d)); CKI(pack_vec(p + ".mlp_norm.weight", n2, d, st));
      CKI(pack_vec(p + ".attention.q_norm.weight", e.qkn + (long)i * 2 * d, d, st));
      CKI(pack_vec(p + ".attention.k_norm.weight", e.qkn + (long)i * 2 * d + d, d, st));
      e.wqkvg.push_back(a); e.wo.push_back(b); e.w13.push_back(c); e.w2.push_back(dd); e.an.push_back(n1); e.mn.push_back(n2);
    }
    return ECHO_OK;
  }
  int pack_kv_proj(const std::string& src, EncW& e, hipStream_t st) {
    const int D = cfg.model_size, L = cfg.num_layers;
    CKI(walloc(&e.kv_w, (long)L * 2 * D, e.d));
    for (int l = 0; l < L; ++l) {
      const std::string p = "blocks." + std::to_string(l) + ".attention.";
      CKI(pack(p + "wk_" + src + ".weight", e.kv_w, e.d, l * 2 * D, D, e.d, st));
      CKI(pack(p + "wv_" + src + ".weight", e.kv_w, e.d, l * 2 * D + D, D, e.d, st));
    }
    return ECHO_OK;
  }
  int pack_patch_in(const std::string& pre, EncW& e, hipStream_t st) {
    e.in_k = cfg.latent_size * cfg.speaker_patch_size;
    if (e.in_k % KE) return fail("latent_size * patch must be a multiple of the GEMM K step");
    CKI(walloc(&e.in_w, rup(e.d, 128), e.in_k));
    CKI(pack(pre + ".in_proj.weight", e.in_w, e.in_k, 0, e.d, e.in_k, st));
    CKI(walloc(&e.in_b, 1, e.d));
    CKI(pack_vec(pre + ".in_proj.bias", e.in_b, e.d, st));
    return ECHO_OK;
  }

  int finalize_dit(hipStream_t st) override {
    const int D = cfg.model_size, L = cfg.num_layers, H = cfg.num_heads, F = cfg.intermediate_size, E = cfg.timestep_embed_size;
    const int R = cfg.adaln_rank;
    if (D / H != 128 || D % 128 || F % 64 || E % KE || R % KE || D % KE) return fail("unsupported EchoDiT sizes (head_dim must be 128)");
    fp8 = cfg.dit_fp8 != 0 && sizeof(T) == 2;
    if (L * 2 > MAXROWS * 8) return fail("too many layers");
    lat_pad = (int)rup(cfg.latent_size, KE);
    // encoders
    CKI(walloc(&text_emb, cfg.text_vocab_size, cfg.text_model_size));
    CKI(pack("text_encoder.text_embedding.weight", text_emb, cfg.text_model_size, 0, cfg.text_vocab_size, cfg.text_model_size, st));
    CKI(pack_encoder("text_encoder", tenc, cfg.text_model_size, cfg.text_num_heads, cfg.text_intermediate_size, cfg.text_num_layers, st));
    CKI(pack_encoder("speaker_encoder", senc, cfg.speaker_model_size, cfg.speaker_num_heads, cfg.speaker_intermediate_size, cfg.speaker_num_layers, st));
    CKI(pack_patch_in("speaker_encoder", senc, st));
    CKI(walloc(&tenc.out_norm, 1, tenc.d)); CKI(pack_vec("text_norm.weight", tenc.out_norm, tenc.d, st));
    CKI(walloc(&senc.out_norm, 1, senc.d)); CKI(pack_vec("speaker_norm.weight", senc.out_norm, senc.d, st));
    CKI(pack_kv_proj("text", tenc, st));
    CKI(pack_kv_proj("speaker", senc, st));
    if (cfg.has_latent_encoder) {
      CKI(pack_encoder("latent_encoder", lenc, cfg.speaker_model_size, cfg.speaker_num_heads, cfg.speaker_intermediate_size, cfg.speaker_num_layers, st));
      CKI(pack_patch_in("latent_encoder", lenc, st));
      CKI(walloc(&lenc.out_norm, 1, lenc.d)); CKI(pack_vec("latent_norm.weight", lenc.out_norm, lenc.d, st));
      CKI(pack_kv_proj("latent", lenc, st));
    }
    // conditioning
    CKI(walloc(&cond0, D, E)); CKI(pack("cond_module.0.weight", cond0, E, 0, D, E, st));
    CKI(walloc(&cond2, D, D)); CKI(pack("cond_module.2.weight", cond2, D, 0, D, D, st));
    CKI(walloc(&cond4, 3L * D, D)); CKI(pack("cond_module.4.weight", cond4, D, 0, 3 * D, D, st));
    CKI(walloc(&in_w, D, lat_pad)); CKI(pack("in_proj.weight", in_w, lat_pad, 0, D, cfg.latent_size, st));
    CKI(walloc(&in_b, 1, D)); CKI(pack_vec("in_proj.bias", in_b, D, st));
    // blocks
    CKI(walloc(&qkn, (long)L * 2, D));
    rank_pad = (int)rup(R, 128);
    CKI(walloc(&mod_down, (long)L * 2 * 3 * rank_pad, D));
    CKI(walloc(&mod_up, (long)L * 2 * 3 * D, R));
    CKI(walloc(&mod_up_b, (long)L * 2 * 3, D));
    const char* adn[2] = {"attention_adaln", "mlp_adaln"};
    const char* parts[3] = {"shift", "scale", "gate"};
    for (int l = 0; l < L; ++l) {
      const std::string p = "blocks." + std::to_string(l);
      T *a, *b, *c, *dd;
      CKI(walloc(&a, 4L * D, D));
      const char* names[4] = {"wq", "wk", "wv", "gate"};
      for (int j = 0; j < 4; ++j) CKI(pack(p + ".attention." + names[j] + ".weight", a, D, j * D, D, D, st));
      CKI(walloc(&b, D, D)); CKI(pack(p + ".attention.wo.weight", b, D, 0, D, D, st));
      CKI(walloc(&c, rup(2 * F, 128), D));
      CKI(pack(p + ".mlp.w1.weight", c, D, 0, F, D, st, 0));
      CKI(pack(p + ".mlp.w3.weight", c, D, 0, F, D, st, 1));
      CKI(walloc(&dd, D, F)); CKI(pack(p + ".mlp.w2.weight", dd, F, 0, D, F, st));
      wqkvg.push_back(a); wo.push_back(b); w13.push_back(c); w2.push_back(dd);
      if (fp8) {
        CKI(quant_weight(a, 4L * D, D, q_wqkvg, s_wqkvg, st));
        CKI(quant_weight(b, D, D, q_wo, s_wo, st));
        CKI(quant_weight(c, rup(2 * F, 128), D, q_w13, s_w13, st));
        CKI(quant_weight(dd, D, F, q_w2, s_w2, st));
      }
      CKI(pack_vec(p + ".attention.q_norm.weight", qkn + (long)l * 2 * D, D, st));
      CKI(pack_vec(p + ".attention.k_norm.weight", qkn + (long)l * 2 * D + D, D, st));
      for (int w = 0; w < 2; ++w)
        for (int q = 0; q < 3; ++q) {
          const long z = (long)(2 * l + w) * 3 + q;
          const std::string ap = p + "." + adn[w] + "." + parts[q];
          CKI(pack(ap + "_down.weight", mod_down, D, (int)(z * rank_pad), R, D, st));
          CKI(pack(ap + "_up.weight", mod_up, R, (int)(z * D), D, R, st));
          CKI(pack_vec(ap + "_up.bias", mod_up_b + z * D, D, st));
        }
    }
    CKI(walloc(&out_norm, 1, D)); CKI(pack_vec("out_norm.weight", out_norm, D, st));
    CKI(walloc(&out_w, 128, D)); CKI(pack("out_proj.weight", out_w, D, 0, cfg.latent_size, D, st));
    CKI(walloc(&out_b, 1, 128)); CKI(pack_vec("out_proj.bias", out_b, cfg.latent_size, st));
    CK(hipStreamSynchronize(st));
    drop_raw({"text_encoder.", "speaker_encoder.", "latent_encoder.", "text_norm", "speaker_norm", "latent_norm", "cond_module.",
              "in_proj.", "blocks.", "out_norm", "out_proj."});
    CK(b_nkeys.reserve(sizeof(int) * 4 * MAXROWS));
    dit_ready = true;
    return ECHO_OK;
  }
  void drop_raw(const std::vector<std::string>& prefixes) {
    for (auto it = raw.begin(); it != raw.end();) {
      bool hit = false;
      for (auto& p : prefixes) if (it->first.compare(0, p.size(), p) == 0) { hit = true; break; }
      if (hit) { if (it->second.d) (void)hipFree(it->second.d); it = raw.erase(it); } else ++it;
    }
  }
  int set_rope(const void* t, int npos) override { rope = (const float2*)t; rope_npos = npos; return ECHO_OK; }
  int set_ae_rope(const void* t, int npos) override { ae_rope = (const float*)t; ae_rope_npos = npos; return ECHO_OK; }

  // ------------------------------------------------------------------ attention dispatch
  struct SegDesc {
    const T* K = nullptr; long k_ld = 0, k_row_stride = 0;
    const T* Vt = nullptr; long vt_ld = 0, vt_row_stride = 0;
    const float* bias = nullptr; long bias_ld = 0;
    int kv_mod = 0;       // kv row = row % kv_mod (0: row itself)
    int maxk = 0;         // max keys over rows (host knowledge)
    int which = 0;        // slot in the nkeys table (0 self, 1 latent, 2 text, 3 speaker)
  };
  // q/gate/out live in [rows*S][..] buffers with per-row stride S*ld
  int attention(const T* q, long q_ld, const T* gate, long g_ld, T* out, long o_ld, int rows, int S, int H, const SegDesc* segs,
                int nseg, bool causal, hipStream_t st, uint8_t* out8 = nullptr, long o8_ld = 0, float o8_inv = 0.f, int g_act = 0) {
    const int* nk = b_nkeys.as<int>();
    if constexpr (Num<T>::is_bf16) {
      AttnArgs a;
      memset(&a, 0, sizeof(a));
      a.Q = q; a.q_ld = q_ld; a.q_row_stride = (long)S * q_ld;
      a.O = out; a.o_ld = o_ld; a.o_row_stride = (long)S * o_ld;
      a.G = gate; a.g_ld = g_ld; a.g_row_stride = (long)S * g_ld;
      a.O8 = out8; a.o8_ld = o8_ld; a.o8_row_stride = (long)S * o8_ld; a.o8_inv = o8_inv;
      a.g_act = g_act;
      a.S = S; a.H = H; a.rows = rows; a.causal = causal ? 1 : 0; a.scale = 1.0f / sqrtf(128.0f);
      int n = 0;
      for (int s = 0; s < nseg; ++s) {
        if (segs[s].maxk <= 0) continue;
        AttnSeg& g = a.seg[n++];
        g.K = segs[s].K; g.k_ld = segs[s].k_ld; g.k_row_stride = segs[s].k_row_stride; g.k_head_stride = 128;
        g.Vt = segs[s].Vt; g.vt_ld = segs[s].vt_ld; g.vt_row_stride = segs[s].vt_row_stride; g.vt_head_stride = 128 * segs[s].vt_ld;
        g.nkeys = nk + segs[s].which * MAXROWS;
        g.bias = segs[s].bias; g.bias_row_stride = segs[s].bias_ld;
        g.kv_mod = segs[s].kv_mod;
      }
      a.nseg = n;
      if (n == 0) return fail("attention without keys");
      CK(b_attn_redo.reserve((size_t)attn_redo_words(rows, H, S) * sizeof(int)));
      a.redo = b_attn_redo.as<int>();
      if (profiling) {
        if (attn_events_used == attn_events.size()) {
          hipEvent_t e0, e1;
          CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
          attn_events.emplace_back(e0, e1);
        }
        auto& e = attn_events[attn_events_used++];
        CK(hipEventRecord(e.first, st));
        CK(launch_attention_bf16(a, st));
        CK(hipEventRecord(e.second, st));
        return ECHO_OK;
      }
      CK(launch_attention_bf16(a, st));
      return ECHO_OK;
    } else {
      return attention_f32(q, q_ld, gate, g_ld, out, o_ld, rows, S, H, segs, nseg, causal, st);
    }
  }

  // unfused exact-fp32 attention for parity mode: scores GEMM -> softmax -> PV GEMM
  int attention_f32(const T* q, long q_ld, const T* gate, long g_ld, T* out, long o_ld, int rows, int S, int H, const SegDesc* segs,
                    int nseg, bool causal, hipStream_t st) {
    if constexpr (!Num<T>::is_bf16) {
      int off[4] = {0, 0, 0, 0}, wdt[4] = {0, 0, 0, 0}, slot[4] = {0, 0, 0, 0};
      const SegDesc* act[4];
      int n = 0, tot = 0;
      for (int s = 0; s < nseg; ++s) {
        if (segs[s].maxk <= 0) continue;
        act[n] = &segs[s]; off[n] = tot; wdt[n] = (int)rup(segs[s].maxk, 32); slot[n] = segs[s].which; tot += wdt[n]; ++n;
      }
      if (n == 0) return fail("attention without keys");
      const long ld = tot;
      CK(b_scores.reserve((size_t)rows * H * S * ld * sizeof(float)));
      CK(b_biasrows.reserve((size_t)rows * ld * sizeof(float)));
      CK(b_segmeta.reserve(sizeof(int) * (8 + 4 * MAXROWS)));
      // nkeys table reordered to the active segments
      int hmeta[8];
      for (int i = 0; i < 4; ++i) { hmeta[i] = off[i]; hmeta[4 + i] = wdt[i]; }
      CK(hipMemcpyAsync(b_segmeta.p, hmeta, sizeof(hmeta), hipMemcpyHostToDevice, st));
      int* nk_act = b_segmeta.as<int>() + 8;
      for (int i = 0; i < n; ++i)
        CK(hipMemcpyAsync(nk_act + i * MAXROWS, b_nkeys.as<int>() + slot[i] * MAXROWS, sizeof(int) * MAXROWS, hipMemcpyDeviceToDevice, st));
      CK(hipStreamSynchronize(st));  // hmeta is a stack array (parity mode only)
      const float *b2 = nullptr, *b3 = nullptr; long b2s = 0, b3s = 0; int b2m = 0, b3m = 0;
      // slots 2/3 of build_bias_kernel refer to ACTIVE segment indices here; map text/speaker biases accordingly
      float* biasrows = b_biasrows.as<float>();
      {
        // generic: write validity first, then add per-key biases segment by segment
        hipLaunchKernelGGL(build_bias_kernel, dim3((unsigned)(((long)rows * tot + 255) / 256)), dim3(256), 0, st, biasrows, ld, rows, n,
                           b_segmeta.as<int>(), b_segmeta.as<int>() + 4, nk_act, MAXROWS, b2, b2s, b2m, b3, b3s, b3m, tot);
        CK(hipGetLastError());
        for (int i = 0; i < n; ++i)
          if (act[i]->bias) {
            // add bias[kvrow][k] to columns off[i].. for every row
            CKI(add_seg_bias(biasrows, ld, rows, off[i], std::min(wdt[i], (int)act[i]->bias_ld), act[i]->bias, act[i]->bias_ld, act[i]->kv_mod, st));
          }
      }
      float* sc = b_scores.as<float>();
      const float scale = 1.0f / sqrtf(128.0f);
      for (int i = 0; i < n; ++i) {
        const SegDesc& sg = *act[i];
        const bool shared1 = sg.kv_mod == 1;      // one kv row shared by every query row: a single group with batch stride 0
        const int groups = shared1 ? 1 : sg.kv_mod ? (rows + sg.kv_mod - 1) / sg.kv_mod : 1;
        const int per = shared1 ? rows : sg.kv_mod ? sg.kv_mod : rows;
        for (int gidx = 0; gidx < groups; ++gidx) {
          const int r0 = gidx * per, nr = std::min(per, rows - r0);
          GemmArgs g = G(q + (long)r0 * S * q_ld, q_ld, sg.K + (sg.kv_mod ? 0 : (long)r0 * sg.k_row_stride), sg.k_ld,
                         (T*)(sc + (long)r0 * H * S * ld + off[i]), ld, S, wdt[i], 128);
          g.nbatch = nr * H; g.nbi = H;
          g.a_bo = (long)S * q_ld; g.a_bi = 128;
          g.w_bo = shared1 ? 0 : sg.k_row_stride; g.w_bi = 128;
          g.c_bo = (long)H * S * ld; g.c_bi = (long)S * ld;
          g.acc_scale = scale;
          CKI(run(g, st));
        }
      }
      CK(launch_softmax_f32(sc, ld, S, rows * H, tot, tot, biasrows, ld, H, causal ? 1 : 0, 0, st));
      for (int i = 0; i < n; ++i) {
        const SegDesc& sg = *act[i];
        const bool shared1 = sg.kv_mod == 1;
        const int groups = shared1 ? 1 : sg.kv_mod ? (rows + sg.kv_mod - 1) / sg.kv_mod : 1;
        const int per = shared1 ? rows : sg.kv_mod ? sg.kv_mod : rows;
        for (int gidx = 0; gidx < groups; ++gidx) {
          const int r0 = gidx * per, nr = std::min(per, rows - r0);
          T* o = out + (long)r0 * S * o_ld;
          GemmArgs g = G((const T*)(sc + (long)r0 * H * S * ld + off[i]), ld, sg.Vt + (sg.kv_mod ? 0 : (long)r0 * sg.vt_row_stride),
                         sg.vt_ld, o, o_ld, S, 128, wdt[i]);
          g.nbatch = nr * H; g.nbi = H;
          g.a_bo = (long)H * S * ld; g.a_bi = (long)S * ld;
          g.w_bo = shared1 ? 0 : sg.vt_row_stride; g.w_bi = 128 * sg.vt_ld;
          g.c_bo = (long)S * o_ld; g.c_bi = 128;
          if (i > 0) { g.res = o; g.ldres = o_ld; g.res_bo = g.c_bo; g.res_bi = g.c_bi; }
          CKI(run(g, st));
        }
      }
      if (gate) {
        const long n_el = (long)rows * S * H * 128;
        hipLaunchKernelGGL(gate_mul_kernel<T>, dim3((unsigned)((n_el + 255) / 256)), dim3(256), 0, st, out, o_ld, gate, g_ld,
                           (long)rows * S, H * 128);
        CK(hipGetLastError());
      }
      return ECHO_OK;
    } else {
      return fail("attention_f32 called in bf16 mode");
    }
  }
  int add_seg_bias(float* biasrows, long ld, int rows, int off, int w, const float* bias, long bias_ld, int kv_mod, hipStream_t st);

  // ------------------------------------------------------------------ encoders (model.py:392-469)
  // x: (B*Tn, d) in b_ex.  Runs L blocks in place.
  int run_encoder(EncW& e, int B, int Tn, bool causal, const SegDesc& self_seg_proto, hipStream_t st) {
    const int d = e.d, M = B * Tn, F = e.F, H = e.H;
    const int Tpad = (int)rup(Tn, 64);
    CK(b_exn.reserve((size_t)(M + 128) * d * sizeof(T)));
    CK(b_eqkvg.reserve((size_t)(M + 128) * 4 * d * sizeof(T)));
    CK(b_evt.reserve((size_t)B * d * Tpad * sizeof(T)));
    CK(b_eattn.reserve((size_t)(M + 128) * d * sizeof(T)));
    CK(b_eh.reserve((size_t)(M + 128) * F * sizeof(T)));
    T *x = b_ex.as<T>(), *xn = b_exn.as<T>(), *qkvg = b_eqkvg.as<T>(), *vt = b_evt.as<T>(), *ao = b_eattn.as<T>(), *hh = b_eh.as<T>();
    for (int i = 0; i < e.L; ++i) {
      CK(launch_norm<T>(NORM_RMS_W, x, d, xn, d, M, d, cfg.norm_eps, e.an[i], nullptr, st));
      if (d % 256 == 0) {
        GemmArgs g = G(xn, d, e.wqkvg[i], d, qkvg, 4 * d, M, 4 * d, d);
        g.qkv_mode = 1; g.qkv_D = d; g.qkv_S = Tn; g.rope_heads = H; g.pos0 = 0; g.qk_eps = cfg.norm_eps;
        g.qk_w = e.qkn + (long)i * 2 * d; g.rope = rope; g.vt = vt; g.vt_ld = Tpad; g.vt_row_stride = (long)d * Tpad;
        CKI(run(g, st));
      } else {
        CKI(run(G(xn, d, e.wqkvg[i], d, qkvg, 4 * d, M, 4 * d, d), st));
        CK(launch_headnorm_rope_nt<T>(qkvg, 4 * d, d, 2, M, Tn, H, e.qkn + (long)i * 2 * d, d, cfg.norm_eps, 1, H, rope, 0, 1, st));
        CK(launch_transpose_heads<T>(qkvg + 2 * d, 4 * d, vt, Tpad, (long)d * Tpad, B, Tn, H, 128, st));
      }
      SegDesc sg = self_seg_proto;
      sg.K = qkvg + d; sg.k_ld = 4 * d; sg.k_row_stride = (long)Tn * 4 * d;
      sg.Vt = vt; sg.vt_ld = Tpad; sg.vt_row_stride = (long)d * Tpad;
      CKI(attention(qkvg, 4 * d, qkvg + 3 * d, 4 * d, ao, d, B, Tn, H, &sg, 1, causal, st));
      GemmArgs g = G(ao, d, e.wo[i], d, x, d, M, d, d);
      g.res = x; g.ldres = d;
      CKI(run(g, st));
      CK(launch_norm<T>(NORM_RMS_W, x, d, xn, d, M, d, cfg.norm_eps, e.mn[i], nullptr, st));
      GemmArgs g1 = G(xn, d, e.w13[i], d, hh, F, M, 2 * F, d);
      g1.swiglu = 1;
      CKI(run(g1, st));
      GemmArgs g2 = G(hh, F, e.w2[i], F, x, d, M, d, F);
      g2.res = x; g2.ldres = d;
      CKI(run(g2, st));
    }
    return ECHO_OK;
  }

  // after the encoder: final norm, K/V projection for all DiT layers, k_norm (+ optional RoPE), Vᵀ
  int project_kv(EncW& e, int B, int Tn, DevBuf& b_kv, DevBuf& b_vt, int& pad_rows, int& vld, bool rope_keys, hipStream_t st) {
    const int D = cfg.model_size, L = cfg.num_layers, H = cfg.num_heads, M = B * Tn;
    const long ld = (long)L * 2 * D;
    pad_rows = (int)rup(Tn, 128);
    vld = (int)rup(Tn, 64);
    CK(b_kv.reserve((size_t)((long)B * Tn + 256) * ld * sizeof(T)));
    CK(b_vt.reserve((size_t)L * B * D * vld * sizeof(T) + 4096));
    T *x = b_ex.as<T>(), *xn = b_exn.as<T>(), *kv = b_kv.as<T>(), *vt = b_vt.as<T>();
    CK(launch_norm<T>(NORM_RMS_W, x, e.d, xn, e.d, M, e.d, cfg.norm_eps, e.out_norm, nullptr, st));
    CKI(run(G(xn, e.d, e.kv_w, e.d, kv, ld, M, L * 2 * D, e.d), st));
    // k_norm of each DiT layer (model.py:274,281,289) on that layer's K columns; latent keys also get half-head RoPE at 4*i
    CK(launch_headnorm_rope_nt<T>(kv, ld, 2 * D, L, M, Tn, H, qkn + D, 2 * D, cfg.norm_eps, 1, rope_keys ? H / 2 : 0, rope, 0,
                                  cfg.speaker_patch_size, st));
    for (int l = 0; l < L; ++l)
      CK(launch_transpose_heads<T>(kv + (long)l * 2 * D + D, ld, vt + (long)l * B * D * vld, vld, (long)D * vld, B, Tn, H, 128, st));
    return ECHO_OK;
  }

  int upload_bias(const float* bias, int B, int Tt, DevBuf& dst, bool& has, long& ldb, hipStream_t st) {
    has = bias != nullptr;
    ldb = Tt;
    if (has) {
      CK(dst.reserve((size_t)B * Tt * sizeof(float)));
      CK(hipMemcpyAsync(dst.p, bias, (size_t)B * Tt * sizeof(float), hipMemcpyDeviceToDevice, st));
    }
    return ECHO_OK;
  }
  int encode_text(const int32_t* ids, const float* bias, const int32_t* nk, int B, int Tt, hipStream_t st) override {
    if (!dit_ready) return fail("echo_finalize_dit was not called");
    if (B < 1 || 3 * B > MAXROWS) return fail("unsupported batch size");
    if (!rope) return fail("rope table not set");
    kvB = B;
    text_nk.assign(nk, nk + B);
    int Te = 0;
    for (int b = 0; b < B; ++b) { if (nk[b] < 0 || nk[b] > Tt) return fail("bad text nkeys"); Te = std::max(Te, nk[b]); }
    text_T = Te;
    if (Te == 0) return ECHO_OK;
    if (Te > rope_npos) return fail("rope table too short");
    CKI(upload_bias(bias, B, Tt, b_bias_text, text_has_bias, text_bias_ld, st));
    const int d = tenc.d;
    CK(b_ex.reserve((size_t)((long)B * Te + 128) * d * sizeof(T)));
    CK(b_ein.reserve((size_t)B * Te * sizeof(int)));
    // gather the first Te ids of every batch row
    CK(hipMemcpy2DAsync(b_ein.p, Te * sizeof(int), ids, Tt * sizeof(int), Te * sizeof(int), B, hipMemcpyDeviceToDevice, st));
    CK(launch_embedding<T>(b_ein.as<int>(), text_emb, b_ex.as<T>(), d, B * Te, d, st));
    // encoder self-attention: keys = valid prefix (+ optional bias), rows here are the B batch items
    CKI(push_nkeys_for_encoder(text_nk, st));
    SegDesc proto;
    proto.which = 0; proto.maxk = Te; proto.kv_mod = 0;
    proto.bias = text_has_bias ? b_bias_text.as<float>() : nullptr; proto.bias_ld = text_bias_ld;
    CKI(run_encoder(tenc, B, Te, false, proto, st));
    CKI(project_kv(tenc, B, Te, b_kv_text, b_vt_text, text_pad, text_vld, false, st));
    return ECHO_OK;
  }

  std::vector<int> host_nk = std::vector<int>(4 * MAXROWS, 0);
  // The per-row key counts go to the device through a ring of pinned staging slots, without a host sync: the caller may
  // enqueue the next sampler call while this one is still running (two requests in flight on two streams keep both
  // streams fed).  A slot is reused only after the copy that read it has executed (its event).
  static constexpr int NK_SLOTS = 16;
  int* nk_pinned = nullptr;
  hipEvent_t nk_ev[NK_SLOTS] = {};
  bool nk_used[NK_SLOTS] = {};
  int nk_next = 0;
  int push_nkeys(hipStream_t st) {
    const size_t bytes = host_nk.size() * sizeof(int);
    if (!nk_pinned) {
      CK(hipHostMalloc((void**)&nk_pinned, bytes * NK_SLOTS, hipHostMallocDefault));
      for (auto& e : nk_ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    const int slot = nk_next;
    nk_next = (nk_next + 1) % NK_SLOTS;
    if (nk_used[slot]) CK(hipEventSynchronize(nk_ev[slot]));
    int* stage = nk_pinned + (size_t)slot * host_nk.size();
    memcpy(stage, host_nk.data(), bytes);
    CK(hipMemcpyAsync(b_nkeys.p, stage, bytes, hipMemcpyHostToDevice, st));
    CK(hipEventRecord(nk_ev[slot], st));
    nk_used[slot] = true;
    return ECHO_OK;
  }
  int push_nkeys_for_encoder(const std::vector<int>& nk, hipStream_t st) {
    for (int i = 0; i < MAXROWS; ++i) host_nk[i] = i < (int)nk.size() ? nk[i] : 0;
    return push_nkeys(st);
  }

  int encode_patches(EncW& e, const void* lat, long row_stride, int B, int npatch, hipStream_t st) {
    // lat: (B, npatch*patch, latent) of T with batch stride row_stride elements; x = (in_proj(patches) + b) / 6  (model.py:459-462)
    const int d = e.d, K = e.in_k;
    CK(b_ex.reserve((size_t)((long)B * npatch + 128) * d * sizeof(T)));
    GemmArgs g = G((const T*)lat, K, e.in_w, K, b_ex.as<T>(), d, npatch, d, K);
    g.nbatch = B; g.nbi = 1; g.a_bo = row_stride; g.c_bo = (long)npatch * d;
    g.bias = e.in_b; g.div = 6.0f;
    CKI(run(g, st));
    return ECHO_OK;
  }

  int encode_speaker(const void* lat, const float* bias, const int32_t* nk, int B, int Ts, hipStream_t st) override {
    if (!dit_ready) return fail("echo_finalize_dit was not called");
    if (B < 1 || 3 * B > MAXROWS) return fail("unsupported batch size");
    if (!rope) return fail("rope table not set");
    const int ps = cfg.speaker_patch_size;
    if (Ts % ps) return fail("speaker latent length must be a multiple of the patch size");
    spkB = B;
    spk_nk.assign(nk, nk + B);
    int Se = 0;
    for (int b = 0; b < B; ++b) { if (nk[b] < 0 || nk[b] > Ts / ps) return fail("bad speaker nkeys"); Se = std::max(Se, nk[b]); }
    spk_T = Se;
    if (Se == 0) return ECHO_OK;   // no speaker reference: one always-masked key in the reference, zero keys here
    if (Se > rope_npos) return fail("rope table too short");
    CKI(upload_bias(bias, B, Ts / ps, b_bias_spk, spk_has_bias, spk_bias_ld, st));
    CKI(encode_patches(senc, lat, (long)Ts * cfg.latent_size, B, Se, st));
    std::vector<int> full(B, Se);   // SpeakerEncoder attends causally with no key mask (model.py:467)
    CKI(push_nkeys_for_encoder(full, st));
    SegDesc proto;
    proto.which = 0; proto.maxk = Se;
    CKI(run_encoder(senc, B, Se, true, proto, st));
    CKI(project_kv(senc, B, Se, b_kv_spk, b_vt_spk, spk_pad, spk_vld, false, st));
    return ECHO_OK;
  }

  int encode_latent(const void* lat, int B, int n_latents, long row_stride, hipStream_t st) override {
    if (!dit_ready || !cfg.has_latent_encoder) return fail("latent encoder not available");
    const int ps = cfg.speaker_patch_size;
    if (n_latents % ps) return fail("prefix length must be a multiple of the patch size");
    latB = B;
    lat_T = n_latents / ps;
    if (lat_T == 0) return ECHO_OK;
    if (lat_T * ps > rope_npos) return fail("rope table too short");
    CKI(encode_patches(lenc, lat, row_stride, B, lat_T, st));
    std::vector<int> full(B, lat_T);
    CKI(push_nkeys_for_encoder(full, st));
    SegDesc proto;
    proto.which = 0; proto.maxk = lat_T;
    CKI(run_encoder(lenc, B, lat_T, true, proto, st));
    CKI(project_kv(lenc, B, lat_T, b_kv_lat, b_vt_lat, lat_pad_rows, lat_vld, true, st));
    return ECHO_OK;
  }

  int scale_speaker_kv(float s, int max_layers, hipStream_t st) override {
    if (spk_T == 0) return ECHO_OK;
    const int D = cfg.model_size, L = cfg.num_layers;
    const int n = max_layers < 0 ? L : std::min(max_layers, L);
    const long ld = (long)L * 2 * D;
    for (int l = 0; l < n; ++l) {
      CK(launch_scale_2d<T>(b_kv_spk.as<T>() + (long)l * 2 * D, ld, spkB * spk_T, 2 * D, s, st));
      // the transposed copy of V follows (same elementwise product, same rounding)
      CK(launch_scale_2d<T>(b_vt_spk.as<T>() + (long)l * spkB * D * spk_vld, spk_vld, spkB * D, spk_T, s, st));
    }
    return ECHO_OK;
  }

  // Per-voice cache (the reference re-runs get_kv_cache_speaker for every chunk: handler.py:750-758,
  // inference.py:333-340, model.py:615-621).  capture: device copy of the speaker K/V + Vᵀ of all layers as they stand
  // (call it right after echo_encode_speaker, before any speaker-KV scaling); bind: copy it back as this context's speaker
  // cache, bit-identical to a fresh encode, without running the 14-layer encoder or the 24 projections.
  int voice_capture(VoiceSnap** out, hipStream_t st) override {
    if (!out) return fail("null output");
    std::unique_ptr<VoiceSnap> v(new VoiceSnap());
    v->device = device; v->precision = cfg.precision; v->model_size = cfg.model_size; v->num_layers = cfg.num_layers;
    v->B = spkB; v->T = spk_T; v->pad = spk_pad; v->vld = spk_vld; v->nk = spk_nk; v->has_bias = spk_has_bias; v->bias_ld = spk_bias_ld;
    if (spkB < 1) return fail("no speaker cache to capture (call echo_encode_speaker first)");
    if (spk_T > 0) {
      const int D = cfg.model_size, L = cfg.num_layers;
      v->kv_bytes = (size_t)spkB * spk_T * L * 2 * D * sizeof(T);
      v->vt_bytes = (size_t)L * spkB * D * spk_vld * sizeof(T);
      CK(v->kv.reserve(v->kv_bytes)); CK(v->vt.reserve(v->vt_bytes));
      CK(hipMemcpyAsync(v->kv.p, b_kv_spk.p, v->kv_bytes, hipMemcpyDeviceToDevice, st));
      CK(hipMemcpyAsync(v->vt.p, b_vt_spk.p, v->vt_bytes, hipMemcpyDeviceToDevice, st));
      if (spk_has_bias) {
        v->bias_bytes = (size_t)spkB * spk_bias_ld * sizeof(float);
        CK(v->bias.reserve(v->bias_bytes));
        CK(hipMemcpyAsync(v->bias.p, b_bias_spk.p, v->bias_bytes, hipMemcpyDeviceToDevice, st));
      }
    }
    *out = v.release();
    return ECHO_OK;
  }
  int voice_bind(const VoiceSnap* v, hipStream_t st) override {
    if (!dit_ready) return fail("echo_finalize_dit was not called");
    if (!v) return fail("null voice");
    if (v->device != device || v->precision != cfg.precision || v->model_size != cfg.model_size || v->num_layers != cfg.num_layers)
      return fail("voice was captured from a different model / device / precision");
    spkB = v->B; spk_T = v->T; spk_pad = v->pad; spk_vld = v->vld; spk_nk = v->nk; spk_has_bias = v->has_bias; spk_bias_ld = v->bias_ld;
    if (spk_T > 0) {
      const int D = cfg.model_size, L = cfg.num_layers;
      CK(b_kv_spk.reserve((size_t)((long)spkB * spk_T + 256) * L * 2 * D * sizeof(T)));
      CK(b_vt_spk.reserve(v->vt_bytes + 4096));
      CK(hipMemcpyAsync(b_kv_spk.p, v->kv.p, v->kv_bytes, hipMemcpyDeviceToDevice, st));
      CK(hipMemcpyAsync(b_vt_spk.p, v->vt.p, v->vt_bytes, hipMemcpyDeviceToDevice, st));
      if (spk_has_bias) {
        CK(b_bias_spk.reserve(v->bias_bytes));
        CK(hipMemcpyAsync(b_bias_spk.p, v->bias.p, v->bias_bytes, hipMemcpyDeviceToDevice, st));
      }
    }
    return ECHO_OK;
  }

  size_t workspace_bytes() override {
    size_t n = b_gemm_ws.cap + b_tune_c.cap + b_flush.cap + b_q8.cap + b_qs.cap + b_h8.cap + b_calib.cap;
    for (DevBuf* b : all_bufs()) n += b->cap;
    return n;
  }
  // ---- fp8 activation-scale calibration (BASELINE C5).  fp8_calibrate(1) clears the per-block maxima and makes the dynamic
  // path record them (static scales are ignored while it is on); fp8_calibration() returns the 2 L maxima seen so far (row scale =
  // amax / 448: [2 l] attention output, [2 l + 1] SwiGLU output); fp8_set_static() installs 2 L scales (the caller applies its margin),
  // n = 0 goes back to dynamic row scales.
  int fp8_calibrate(int on) override {
    if (!fp8) return fail("fp8 calibration needs an engine created with dit_fp8");
    if (on) {
      CK(b_calib.reserve((size_t)2 * cfg.num_layers * sizeof(float)));
      CK(hipMemset(b_calib.p, 0, (size_t)2 * cfg.num_layers * sizeof(float)));
    }
    calibrating = on != 0;
    return ECHO_OK;
  }
  int fp8_calibration(float* out, int n) override {
    if (!fp8 || !b_calib.p || n != 2 * cfg.num_layers || !out) return fail("fp8_calibration: no calibration data / n != 2 * num_layers");
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(out, b_calib.p, (size_t)n * sizeof(float), hipMemcpyDeviceToHost));
    return ECHO_OK;
  }
  int fp8_set_static(const float* sc, int n) override {
    if (!fp8) return fail("static fp8 scales need an engine created with dit_fp8");
    if (n == 0) { fp8_static.clear(); return ECHO_OK; }
    if (n != 2 * cfg.num_layers || !sc) return fail("fp8_set_static: n != 2 * num_layers");
    for (int i = 0; i < n; ++i)
      if (!(sc[i] > 0.0f) || !std::isfinite(sc[i])) return fail("fp8_set_static: scales must be positive and finite");
    fp8_static.assign(sc, sc + n);
    return ECHO_OK;
  }
  // grows every workspace a request of this geometry touches (B utterances per sampler call, S latents, Tt text tokens,
  // Ts speaker latents, T_dac frames per DAC decode), so that the first request allocates nothing (SURVEY.md §8b "ownership")
  int reserve_workspace(int B, int S, int Tt, int Ts, int T_dac) override {
    const int D = cfg.model_size, L = cfg.num_layers;
    if (B < 1 || 3 * B > MAXROWS || S < 1) return fail("bad workspace geometry");
    if (dit_ready) {
      CKI(reserve_dit_ws(3 * B * S, 3 * B, S));
      CK(b_xstate.reserve((size_t)B * S * cfg.latent_size * sizeof(float)));
      const long ld = (long)L * 2 * D;
      if (Tt > 0) { CK(b_kv_text.reserve((size_t)((long)B * Tt + 256) * ld * sizeof(T))); CK(b_vt_text.reserve((size_t)L * B * D * rup(Tt, 64) * sizeof(T) + 4096)); }
      const int Sk = Ts / std::max(1, cfg.speaker_patch_size);
      if (Sk > 0) { CK(b_kv_spk.reserve((size_t)((long)B * Sk + 256) * ld * sizeof(T))); CK(b_vt_spk.reserve((size_t)L * B * D * rup(Sk, 64) * sizeof(T) + 4096)); }
    }
    if (dac_ready && T_dac > 0) {
      long upf = 1;
      for (int i = 0; i < cfg.dac_n_up; ++i) upf *= cfg.dac_up_factors[i];
      const int C = cfg.dac_latent_dim;
      long maxel = (long)T_dac * upf * std::max(C * 4, cfg.dac_decoder_dim);
      long rows = (long)T_dac * upf;
      for (auto& b : dblocks) { rows *= b.r; maxel = std::max(maxel, rows * b.co); }
      const long PADF = 64L * std::max(cfg.dac_decoder_dim, 4 * C);
      const size_t bytes = (size_t)(maxel + PADF + 128L * 4 * C) * sizeof(float);
      CK(b_dacA.reserve(bytes)); CK(b_dacB.reserve(bytes)); CK(b_dacC.reserve(bytes));
    }
    return ECHO_OK;
  }

  int debug_get_kv(int which, int layer, float* k, float* v, int* Bo, int* To) override {
    const int D = cfg.model_size, L = cfg.num_layers;
    const long ld = (long)L * 2 * D;
    DevBuf* kv = which == 0 ? &b_kv_text : which == 1 ? &b_kv_spk : &b_kv_lat;
    const int Tn = which == 0 ? text_T : which == 1 ? spk_T : lat_T;
    const int B = which == 2 ? latB : which == 1 ? spkB : kvB;
    if (Bo) *Bo = B;
    if (To) *To = Tn;
    if (!k || !v || Tn == 0) return ECHO_OK;
    if (layer < 0 || layer >= L) return fail("bad layer");
    CK(launch_convert_to_f32<T>(kv->as<T>() + (long)layer * 2 * D, ld, k, D, B * Tn, D, nullptr));
    CK(launch_convert_to_f32<T>(kv->as<T>() + (long)layer * 2 * D + D, ld, v, D, B * Tn, D, nullptr));
    CK(hipDeviceSynchronize());
    return ECHO_OK;
  }

  // ------------------------------------------------------------------ EchoDiT forward (model.py:563-604)
  int reserve_dit_ws(int M, int rows, int S) {
    const int D = cfg.model_size, F = cfg.intermediate_size;
    const int Sp = (int)rup(S, 64);
    CK(b_xin.reserve((size_t)(M + 128) * lat_pad * sizeof(T)));
    CK(b_x.reserve((size_t)(M + 128) * D * sizeof(T)));
    CK(b_xn.reserve((size_t)(M + 128) * D * sizeof(T)));
    CK(b_qkvg.reserve((size_t)(M + 256) * 4 * D * sizeof(T)));
    CK(b_vt_self.reserve((size_t)rows * D * Sp * sizeof(T) + 4096));
    CK(b_attn.reserve((size_t)(M + 128) * D * sizeof(T)));
    CK(b_h.reserve((size_t)(M + 128) * F * sizeof(T)));
    CK(b_vout.reserve((size_t)(M + 128) * 128 * sizeof(T)));
    return ECHO_OK;
  }
  // host_nk rows must already describe this forward (set_row_keys)
  void set_row_keys(int rows, int B, int S, int start_pos, bool use_latent, const int32_t* ton, const int32_t* son) {
    for (int r = 0; r < MAXROWS; ++r) {
      const int b = B > 0 ? r % B : 0;
      const bool in = r < rows;
      host_nk[0 * MAXROWS + r] = in ? S : 0;
      int nl = 0;
      if (in && use_latent && lat_T > 0) nl = std::min(lat_T, (start_pos + cfg.speaker_patch_size - 1) / cfg.speaker_patch_size);
      host_nk[1 * MAXROWS + r] = nl;
      host_nk[2 * MAXROWS + r] = (in && text_T > 0 && (!ton || ton[r])) ? text_nk[b] : 0;
      host_nk[3 * MAXROWS + r] = (in && spk_T > 0 && spkB > 0 && (!son || son[r])) ? spk_nk[b % spkB] : 0;
    }
  }

  // Rows [row0, row0 + nrows) of one forward share the timestep whose modulation table is `mod` ([2L][3][D]).  The sampler's forwards are
  // one segment; EchoDiT.forward with per-row timesteps (model.py:563-604 takes any t (R,)) is one segment per run of equal t.  Only the
  // launches that read the modulation (the two AdaLN norms and the gated-residual tails of wo / w2) are issued per segment, on row
  // sub-ranges of the same buffers; the t-independent ones (QKVG, attention, w1|w3) stay one launch over all rows.
  struct ModSeg { int row0, nrows; const T* mod; };

  // xin (rows*S, lat_pad) -> vout (rows*S, 128); modrow = this step's [2L][3][D] modulation table (or `segs`: per-row-range tables)
  int forward_rows(int rows, int B, int S, int start_pos, bool use_latent, const T* modrow, hipStream_t st,
                   const std::vector<ModSeg>* segs_in = nullptr) {
    const std::vector<ModSeg> one = {ModSeg{0, rows, modrow}};
    const std::vector<ModSeg>& segs = segs_in ? *segs_in : one;
    const int D = cfg.model_size, L = cfg.num_layers, H = cfg.num_heads, F = cfg.intermediate_size, M = rows * S;
    const int Sp = (int)rup(S, 64);
    T *xin = b_xin.as<T>(), *x = b_x.as<T>(), *xn = b_xn.as<T>(), *qkvg = b_qkvg.as<T>(), *vts = b_vt_self.as<T>(),
      *ao = b_attn.as<T>(), *hh = b_h.as<T>(), *vout = b_vout.as<T>();
    if (start_pos + S > rope_npos) return fail("rope table too short");
    {
      GemmArgs g = G(xin, lat_pad, in_w, lat_pad, x, D, M, D, lat_pad);
      g.bias = in_b;
      CKI(run(g, st));
    }
    const long kvld = (long)L * 2 * D;
    int max_lat = 0, max_text = 0, max_spk = 0;
    for (int r = 0; r < rows; ++r) {
      max_lat = std::max(max_lat, host_nk[1 * MAXROWS + r]);
      max_text = std::max(max_text, host_nk[2 * MAXROWS + r]);
      max_spk = std::max(max_spk, host_nk[3 * MAXROWS + r]);
    }
    static const bool gate_act_on = getenv("ECHO_GATE_ACT") ? atoi(getenv("ECHO_GATE_ACT")) != 0 : true;     // 0: sigmoid in the attention epilogue (A/B aid)
    const int gate_act = (Num<T>::is_bf16 && D % 256 == 0 && gate_act_on) ? 1 : 0;
    for (int l = 0; l < L; ++l) {
      const long mao = (long)(2 * l) * 3 * D, mmo = (long)(2 * l + 1) * 3 * D;     // offsets of this layer's attention / mlp modulation in a table
      // fp8 engine: the AdaLN output goes straight to e4m3 rows (b_q8 / b_qs) when the following GEMM takes fp8 operands
      const bool nq1 = fp8 && D % 256 == 0 && q_wqkvg[l] != nullptr && D <= 4096;
      const bool nq2 = fp8 && q_w13[l] != nullptr && D <= 4096;
      if (nq1) CKI(fp8_reserve(M, std::max(D, F)));
      for (const ModSeg& sgm : segs) {
        const long off = (long)sgm.row0 * S;
        const int Ms = sgm.nrows * S;
        const T* ma = sgm.mod + mao;
        if (nq1) CK(launch_norm_adaln_fp8(x + off * D, D, b_q8.as<uint8_t>() + off * D, D, b_qs.as<float>() + off, Ms, D, cfg.norm_eps, ma + D, ma, st));
        else CK(launch_norm<T>(NORM_ADALN, x + off * D, D, xn + off * D, D, Ms, D, cfg.norm_eps, ma + D, ma, st));
      }
      if (D % 256 == 0) {
        // one launch: projection + q/k head RMSNorm + half-head RoPE + transposed V (gemm.hip fused QKV epilogue)
        GemmArgs g = G(xn, D, wqkvg[l], D, qkvg, 4 * D, M, 4 * D, D);
        g.qkv_mode = 1; g.qkv_D = D; g.qkv_S = S; g.rope_heads = H / 2; g.pos0 = start_pos; g.qk_eps = cfg.norm_eps;
        g.qk_w = qkn + (long)l * 2 * D; g.rope = rope; g.vt = vts; g.vt_ld = Sp; g.vt_row_stride = (long)D * Sp;
        g.qkv_gate_act = gate_act;       // bf16: the gate leaves the GEMM as bf16(sigmoid(gate)), the attention epilogue only multiplies
        if (fp8) CKI(to_fp8(g, q_wqkvg[l], s_wqkvg[l], st, nq1));
        CKI(run(g, st));
      } else {
        CKI(run(G(xn, D, wqkvg[l], D, qkvg, 4 * D, M, 4 * D, D), st));
        CK(launch_headnorm_rope_nt<T>(qkvg, 4 * D, D, 2, M, S, H, qkn + (long)l * 2 * D, D, cfg.norm_eps, 1, H / 2, rope, start_pos, 1, st));
        CK(launch_transpose_heads<T>(qkvg + 2 * D, 4 * D, vts, Sp, (long)D * Sp, rows, S, H, 128, st));
      }
      SegDesc sg[4];
      sg[0].which = 0; sg[0].maxk = S; sg[0].K = qkvg + D; sg[0].k_ld = 4 * D; sg[0].k_row_stride = (long)S * 4 * D;
      sg[0].Vt = vts; sg[0].vt_ld = Sp; sg[0].vt_row_stride = (long)D * Sp;
      sg[1].which = 1; sg[1].maxk = max_lat; sg[1].kv_mod = latB;
      if (max_lat > 0) {
        sg[1].K = b_kv_lat.as<T>() + (long)l * 2 * D; sg[1].k_ld = kvld; sg[1].k_row_stride = (long)lat_T * kvld;
        sg[1].Vt = b_vt_lat.as<T>() + (long)l * latB * D * lat_vld; sg[1].vt_ld = lat_vld; sg[1].vt_row_stride = (long)D * lat_vld;
      }
      sg[2].which = 2; sg[2].maxk = max_text; sg[2].kv_mod = kvB;
      if (max_text > 0) {
        sg[2].K = b_kv_text.as<T>() + (long)l * 2 * D; sg[2].k_ld = kvld; sg[2].k_row_stride = (long)text_T * kvld;
        sg[2].Vt = b_vt_text.as<T>() + (long)l * kvB * D * text_vld; sg[2].vt_ld = text_vld; sg[2].vt_row_stride = (long)D * text_vld;
        if (text_has_bias) { sg[2].bias = b_bias_text.as<float>(); sg[2].bias_ld = text_bias_ld; }
      }
      sg[3].which = 3; sg[3].maxk = max_spk; sg[3].kv_mod = spkB;
      if (max_spk > 0) {
        sg[3].K = b_kv_spk.as<T>() + (long)l * 2 * D; sg[3].k_ld = kvld; sg[3].k_row_stride = (long)spk_T * kvld;
        sg[3].Vt = b_vt_spk.as<T>() + (long)l * spkB * D * spk_vld; sg[3].vt_ld = spk_vld; sg[3].vt_row_stride = (long)D * spk_vld;
        if (spk_has_bias) { sg[3].bias = b_bias_spk.as<float>(); sg[3].bias_ld = spk_bias_ld; }
      }
      // static activation scales (fp8 engine, calibrated): the attention epilogue writes wo's e4m3 operand, the SwiGLU tail w2's
      const bool st8 = fp8 && (int)fp8_static.size() == 2 * L && q_wo[l] != nullptr && q_w2[l] != nullptr && q_w13[l] != nullptr && !calibrating;
      if (st8) {
        CKI(fp8_reserve(M, std::max(D, F)));
        CK(b_h8.reserve((size_t)(M + 256) * F));
        CKI(attention(qkvg, 4 * D, qkvg + 3 * D, 4 * D, ao, D, rows, S, H, sg, 4, false, st, b_q8.as<uint8_t>(), D, 1.0f / fp8_static[2 * l], gate_act));
      } else {
        CKI(attention(qkvg, 4 * D, qkvg + 3 * D, 4 * D, ao, D, rows, S, H, sg, 4, false, st, nullptr, 0, 0.f, gate_act));
      }
      for (const ModSeg& sgm : segs) {       // x += tanh(gate) * wo(attn): the gate is this segment's
        const long off = (long)sgm.row0 * S;
        const int Ms = sgm.nrows * S;
        GemmArgs g = G(ao + off * D, D, wo[l], D, x + off * D, D, Ms, D, D);
        g.colscale = sgm.mod + mao + 2 * D; g.res = x + off * D; g.ldres = D;
        if (st8) {
          g.A = b_q8.as<uint8_t>() + off * D; g.lda = D; g.W = q_wo[l]; g.ldw = D; g.fp8 = 1; g.a_scale = nullptr; g.a_scale_const = fp8_static[2 * l]; g.w_scale = s_wo[l];
        } else if (fp8) CKI(to_fp8(g, q_wo[l], s_wo[l], st, false, 2 * l));
        CKI(run(g, st));
      }
      if (nq2) CKI(fp8_reserve(M, std::max(D, F)));
      for (const ModSeg& sgm : segs) {
        const long off = (long)sgm.row0 * S;
        const int Ms = sgm.nrows * S;
        const T* mm = sgm.mod + mmo;
        if (nq2) CK(launch_norm_adaln_fp8(x + off * D, D, b_q8.as<uint8_t>() + off * D, D, b_qs.as<float>() + off, Ms, D, cfg.norm_eps, mm + D, mm, st));
        else CK(launch_norm<T>(NORM_ADALN, x + off * D, D, xn + off * D, D, Ms, D, cfg.norm_eps, mm + D, mm, st));
      }
      {
        GemmArgs g = G(xn, D, w13[l], D, hh, F, M, 2 * F, D);
        g.swiglu = 1;
        if (fp8) CKI(to_fp8(g, q_w13[l], s_w13[l], st, nq2));
        if (st8 && g.fp8) { g.c8 = b_h8.p; g.c8_ld = F; g.c8_inv = 1.0f / fp8_static[2 * l + 1]; }
        CKI(run(g, st));
      }
      for (const ModSeg& sgm : segs) {       // x += tanh(gate) * w2(h)
        const long off = (long)sgm.row0 * S;
        const int Ms = sgm.nrows * S;
        GemmArgs g = G(hh + off * F, F, w2[l], F, x + off * D, D, Ms, D, F);
        g.colscale = sgm.mod + mmo + 2 * D; g.res = x + off * D; g.ldres = D;
        if (st8) {
          g.A = b_h8.as<uint8_t>() + off * F; g.lda = F; g.W = q_w2[l]; g.ldw = F; g.fp8 = 1; g.a_scale = nullptr; g.a_scale_const = fp8_static[2 * l + 1]; g.w_scale = s_w2[l];
        } else if (fp8) CKI(to_fp8(g, q_w2[l], s_w2[l], st, false, 2 * l + 1));
        CKI(run(g, st));
      }
    }
    CK(launch_norm<T>(NORM_RMS_W, x, D, xn, D, M, D, cfg.norm_eps, out_norm, nullptr, st));
    {
      GemmArgs g = G(xn, D, out_w, D, vout, 128, M, cfg.latent_size, D);
      g.bias = out_b;
      CKI(run(g, st));
    }
    return ECHO_OK;
  }

  // modulation tables for N timesteps (model.py:583, 70-74, 79-81): mod[n][2L][3][D]
  int compute_mod(const T* temb, int N, hipStream_t st) {
    const int D = cfg.model_size, L = cfg.num_layers, E = cfg.timestep_embed_size, R = cfg.adaln_rank;
    const int A = 2 * L * 3;
    CK(b_mod.reserve((size_t)(N + 128) * A * D * sizeof(T)));
    CK(b_c1.reserve((size_t)(N + 128) * D * sizeof(T)));
    CK(b_c2.reserve((size_t)(N + 128) * D * sizeof(T)));
    CK(b_cond.reserve((size_t)(N + 128) * 3 * D * sizeof(T)));
    CK(b_sc.reserve((size_t)(N + 128) * 3 * D * sizeof(T)));
    CK(b_dn.reserve((size_t)A * (N + 128) * R * sizeof(T)));
    T *c1 = b_c1.as<T>(), *c2 = b_c2.as<T>(), *cond = b_cond.as<T>(), *sc = b_sc.as<T>(), *dn = b_dn.as<T>(), *mod = b_mod.as<T>();
    { GemmArgs g = G(temb, E, cond0, E, c1, D, N, D, E); g.act = 1; CKI(run(g, st)); }
    { GemmArgs g = G(c1, D, cond2, D, c2, D, N, D, D); g.act = 1; CKI(run(g, st)); }
    CKI(run(G(c2, D, cond4, D, cond, 3 * D, N, 3 * D, D), st));
    CK(launch_silu<T>(cond, sc, (long)N * 3 * D, st));
    {
      GemmArgs g = G(sc, 3 * D, mod_down, D, dn, R, N, R, D);
      g.nbatch = A; g.nbi = 3;
      g.a_bo = 0; g.a_bi = D;
      g.w_bo = 3L * rank_pad * D; g.w_bi = (long)rank_pad * D;
      g.c_bo = 3L * N * R; g.c_bi = (long)N * R;
      CKI(run(g, st));
    }
    {
      GemmArgs g = G(dn, R, mod_up, R, mod, (long)A * D, N, D, R);
      g.nbatch = A; g.nbi = 3;
      g.a_bo = 3L * N * R; g.a_bi = (long)N * R;
      g.w_bo = 3L * D * R; g.w_bi = (long)D * R;
      g.c_bo = 3L * D; g.c_bi = D;
      g.bias = mod_up_b; g.bias_bo = 3L * D; g.bias_bi = D;
      g.res = cond; g.ldres = 3 * D; g.res_bo = 0; g.res_bi = D;
      CKI(run(g, st));
    }
    CK(launch_mod_finalize<T>(mod, (long)N * 2 * L, D, st));
    return ECHO_OK;
  }

  // temb: (n_t, E) timestep embeddings; row_t (host, rows) = which of them each row uses (nullptr: all rows use embedding 0)
  int dit_forward(const void* x, const void* temb, int n_t, const int32_t* row_t, int rows, int B, int S, int start_pos, int use_latent,
                  const int32_t* ton, const int32_t* son, float* v, hipStream_t st) override {
    if (!dit_ready) return fail("echo_finalize_dit was not called");
    if (rows < 1 || rows > MAXROWS || B < 1 || rows % B || B != kvB) return fail("bad rows/B");
    if (spk_T > 0 && (spkB < 1 || kvB % spkB)) return fail("the speaker cache's batch size must divide the text cache's");
    if (n_t < 1 || n_t > rows || (n_t > 1 && !row_t)) return fail("bad timestep table (1 <= n_t <= rows, row_t needed when n_t > 1)");
    const int M = rows * S;
    CKI(reserve_dit_ws(M, rows, S));
    CKI(compute_mod((const T*)temb, n_t, st));
    const long modstride = (long)2 * cfg.num_layers * 3 * cfg.model_size;
    std::vector<ModSeg> segs;            // runs of consecutive rows with the same timestep
    for (int r = 0; r < rows; ++r) {
      const int ti = row_t ? row_t[r] : 0;
      if (ti < 0 || ti >= n_t) return fail("row_t entry out of range");
      const T* mod = b_mod.as<T>() + (long)ti * modstride;
      if (!segs.empty() && segs.back().mod == mod) ++segs.back().nrows;
      else segs.push_back(ModSeg{r, 1, mod});
    }
    set_row_keys(rows, B, S, start_pos, use_latent != 0, ton, son);
    CKI(push_nkeys(st));
    // x (M, latent) -> xin (M, lat_pad) zero padded
    CK(hipMemcpy2DAsync(b_xin.p, lat_pad * sizeof(T), x, cfg.latent_size * sizeof(T), cfg.latent_size * sizeof(T), M,
                        hipMemcpyDeviceToDevice, st));
    CKI(forward_rows(rows, B, S, start_pos, use_latent != 0, b_mod.as<T>(), st, &segs));
    CK(launch_convert_to_f32<T>(b_vout.as<T>(), 128, v, cfg.latent_size, M, cfg.latent_size, st));
    return ECHO_OK;
  }

  // ------------------------------------------------------------------ sampler (inference.py:427-517)
  int sample_euler(const echo_sampler_params* p, const float* x0, float* out, hipStream_t st) override {
    if (!dit_ready) return fail("echo_finalize_dit was not called");
    const int B = p->B, S = p->S, N = p->num_steps, Lz = cfg.latent_size;
    if (B != kvB || 3 * B > MAXROWS || S < 1 || N < 1) return fail("bad sampler params");
    if (spk_T > 0 && (spkB < 1 || kvB % spkB)) return fail("the speaker cache's batch size must divide the text cache's");
    const int rows3 = 3 * B;
    CKI(reserve_dit_ws(rows3 * S, rows3, S));
    CK(b_xstate.reserve((size_t)B * S * Lz * sizeof(float)));
    float* xs = b_xstate.as<float>();
    if (profiling) {
      for (auto& e : ev) if (!e) CK(hipEventCreate(&e));
      gemm_events_used = 0; attn_events_used = 0;
      CK(hipEventRecord(ev[0], st));
    }
    CK(hipMemcpyAsync(xs, x0, (size_t)B * S * Lz * sizeof(float), hipMemcpyDeviceToDevice, st));
    CKI(compute_mod((const T*)p->temb, N, st));
    if (profiling) CK(hipEventRecord(ev[1], st));
    // CFG row layout: rows [0,B) cond, [B,2B) text-uncond, [2B,3B) speaker-uncond (inference.py:474-475)
    std::vector<int32_t> ton(rows3, 1), son(rows3, 1);
    for (int b = 0; b < B; ++b) { ton[B + b] = 0; son[2 * B + b] = 0; }
    set_row_keys(rows3, B, S, p->start_pos, p->use_latent != 0, ton.data(), son.data());
    CKI(push_nkeys(st));
    const long modstride = (long)2 * cfg.num_layers * 3 * cfg.model_size;
    EulerArgs e;
    memset(&e, 0, sizeof(e));
    e.x = xs; e.v = b_vout.p; e.ldv = 128; e.xin = b_xin.p; e.ld_xin = lat_pad;
    e.B = B; e.S = S; e.L = Lz; e.R = 1;
    e.R_next = p->steps[0].has_cfg ? 3 : 1;
    e.init = 1; e.init_scale = p->has_truncation ? p->init_scale : 1.0f;   // x_t = x_t * truncation_factor (inference.py:478-479) only when one was given; 0.0 is then honoured
    CK(launch_euler<T>(e, st));
    for (int i = 0; i < N; ++i) {
      const echo_step& sp = p->steps[i];
      const int rows = sp.has_cfg ? rows3 : B;
      CKI(forward_rows(rows, B, S, p->start_pos, p->use_latent != 0, b_mod.as<T>() + (long)i * modstride, st));
      if (sp.kv_unscale_after) CKI(scale_speaker_kv(1.0f / p->kv_scale, p->kv_max_layers, st));
      e.init = 0;
      e.R = sp.has_cfg ? 3 : 1;
      e.R_next = (i + 1 < N && p->steps[i + 1].has_cfg) ? 3 : 1;
      e.s_text = p->cfg_scale_text; e.s_spk = p->cfg_scale_speaker;
      e.rescale = sp.rescale; e.r_inv1mt = sp.r_inv1mt; e.r_ratio = sp.r_ratio; e.r_1mt = sp.r_1mt;
      e.dt = sp.dt;
      CK(launch_euler<T>(e, st));
    }
    CK(hipMemcpyAsync(out, xs, (size_t)B * S * Lz * sizeof(float), hipMemcpyDeviceToDevice, st));
    if (profiling) {
      CK(hipEventRecord(ev[2], st));
      CK(hipEventSynchronize(ev[2]));
      CK(hipEventElapsedTime(&prof.ms_mod, ev[0], ev[1]));
      CK(hipEventElapsedTime(&prof.ms_steps, ev[1], ev[2]));
      prof.ms_total = prof.ms_mod + prof.ms_steps;
      collect_gemm_times();
    }
    return ECHO_OK;
  }
  void collect_gemm_times() {
    prof.ms_gemm_sum = 0.f; prof.n_gemm = 0; prof.ms_pp_sum = 0.f; prof.n_pp = 0; prof.flops_pp = 0.0;
    prof.ms_attn_sum = 0.f; prof.n_attn = 0;
    for (size_t i = 0; i < attn_events_used; ++i) {
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, attn_events[i].first, attn_events[i].second) == hipSuccess) { prof.ms_attn_sum += ms; ++prof.n_attn; }
    }
    for (size_t i = 0; i < gemm_events_used; ++i) {
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, gemm_events[i].first, gemm_events[i].second) == hipSuccess) {
        prof.ms_gemm_sum += ms; ++prof.n_gemm;
        if (i < gemm_event_flops.size() && gemm_event_flops[i] > 0.0) { prof.ms_pp_sum += ms; ++prof.n_pp; prof.flops_pp += gemm_event_flops[i]; }
      }
    }
    if (getenv("ECHO_PROFILE_SHAPES")) {   // debugging aid: time per GEMM shape of the profiled call
      std::map<std::vector<long>, std::pair<double, int>> by;
      for (size_t i = 0; i < gemm_events_used && i < gemm_event_shape.size(); ++i) {
        float ms = 0.f;
        if (gemm_event_shape[i].empty() || hipEventElapsedTime(&ms, gemm_events[i].first, gemm_events[i].second) != hipSuccess) continue;
        auto& a = by[gemm_event_shape[i]];
        a.first += ms; a.second += 1;
      }
      for (auto& kv : by) {
        const auto& k = kv.first;
        const double us = 1e3 * kv.second.first / kv.second.second;
        fprintf(stderr, "[echo] shape M=%ld N=%ld K=%ld taps=%ld swiglu=%ld qkv=%ld fp8=%ld cfg=%ld ksplit=%ld nb=%ld: %d launches, %.1f us avg, %.1f ms total, %.0f TF\n",
                k[0], k[1], k[2], k[3], k[4], k[5], k[6], k[7], k[8], k[9], kv.second.second, us, kv.second.first,
                2.0 * k[0] * k[1] * k[2] * k[3] * k[9] / (us * 1e-6) / 1e12);
      }
    }
  }

  // ------------------------------------------------------------------ Fish S1-DAC decode (fp32 weights/activations)
  struct DacLayer { float *wqkv, *wo, *w13, *w2, *an, *fn, *ga, *gf; };
  struct DacUp { float *w, *b, *dw, *db, *lnw, *lnb, *p1w, *p1b, *p2w, *p2b, *gamma; int f; };
  struct DacRU { float *a0, *w7, *b7, *a1, *w1, *b1; };
  struct DacBlock { float *alpha, *wt, *bt; DacRU ru[3]; int ci, co, r; };
  std::vector<DacLayer> dpost; float* dpost_norm = nullptr;
  std::vector<DacUp> dups;
  float *dconv0_w = nullptr, *dconv0_b = nullptr, *dfinal_alpha = nullptr, *dout_w = nullptr, *pca_w = nullptr, *pca_b = nullptr;
  float dout_b = 0.f;
  std::vector<DacBlock> dblocks;
  int pca_kpad = 0;

  int fpack(const std::string& name, float** dst, long rows_pad, int rows, int cols, long cols_pad, hipStream_t st, int swh = -1, float** existing = nullptr) {
    const RawTensor* r = find(name);
    if (!r) return fail("missing tensor: " + name);
    if (r->numel != (long)rows * cols) return fail("shape mismatch for " + name + " (" + std::to_string(r->numel) + " vs " + std::to_string((long)rows * cols) + ")");
    if (!existing) { CK(alloc_zero((void**)dst, (size_t)rows_pad * cols_pad * sizeof(float))); } else { *dst = *existing; }
    CK(launch_pack_rows(r->d, r->dtype, cols, *dst, ECHO_F32, cols_pad, rows, cols, 0, swh, st));
    return ECHO_OK;
  }
  int fvec(const std::string& name, float** dst, int n, hipStream_t st) { return fpack(name, dst, 1, 1, n, rup(n, 4), st); }

  int finalize_dac(hipStream_t st) override {
    const int C = cfg.dac_latent_dim, nh = cfg.dac_post_heads, hd = cfg.dac_post_head_dim, ff = cfg.dac_post_ffn;
    if (C % 32 || C > 1024 || (nh * hd) % 32 || ff % 64 || hd != 64 || C != nh * hd) return fail("unsupported DAC sizes");
    const std::string pm = "quantizer.post_module";
    for (int i = 0; i < cfg.dac_post_layers; ++i) {
      const std::string lp = pm + ".layers." + std::to_string(i);
      DacLayer L{};
      CKI(fpack(lp + ".attention.wqkv.weight", &L.wqkv, rup(3 * nh * hd, 128), 3 * nh * hd, C, C, st));
      CKI(fpack(lp + ".attention.wo.weight", &L.wo, rup(C, 128), C, nh * hd, nh * hd, st));
      CKI(fpack(lp + ".feed_forward.w1.weight", &L.w13, rup(2 * ff, 128), ff, C, C, st, 0));
      CKI(fpack(lp + ".feed_forward.w3.weight", &L.w13, rup(2 * ff, 128), ff, C, C, st, 1, &L.w13));
      CKI(fpack(lp + ".feed_forward.w2.weight", &L.w2, rup(C, 128), C, ff, ff, st));
      CKI(fvec(lp + ".attention_norm.weight", &L.an, C, st));
      CKI(fvec(lp + ".ffn_norm.weight", &L.fn, C, st));
      CKI(fvec(lp + ".attention_layer_scale.gamma", &L.ga, C, st));
      CKI(fvec(lp + ".ffn_layer_scale.gamma", &L.gf, C, st));
      dpost.push_back(L);
    }
    CKI(fvec(pm + ".norm.weight", &dpost_norm, C, st));
    for (int i = 0; i < cfg.dac_n_up; ++i) {
      const std::string up = "quantizer.upsample." + std::to_string(i);
      DacUp U{};
      U.f = cfg.dac_up_factors[cfg.dac_n_up - 1 - i];
      CKI(fpack(up + ".0.conv.weight", &U.w, rup(U.f * C, 128), U.f * C, C, C, st));           // (f*Co, Ci), GEMM form
      CKI(fvec(up + ".0.conv.bias", &U.b, C, st));
      CKI(fpack(up + ".1.dwconv.conv.weight", &U.dw, C, C, 7, 7, st));
      CKI(fvec(up + ".1.dwconv.conv.bias", &U.db, C, st));
      CKI(fvec(up + ".1.norm.weight", &U.lnw, C, st));
      CKI(fvec(up + ".1.norm.bias", &U.lnb, C, st));
      CKI(fpack(up + ".1.pwconv1.weight", &U.p1w, rup(4 * C, 128), 4 * C, C, C, st));
      CKI(fvec(up + ".1.pwconv1.bias", &U.p1b, 4 * C, st));
      CKI(fpack(up + ".1.pwconv2.weight", &U.p2w, rup(C, 128), C, 4 * C, 4 * C, st));
      CKI(fvec(up + ".1.pwconv2.bias", &U.p2b, C, st));
      CKI(fvec(up + ".1.gamma", &U.gamma, C, st));
      dups.push_back(U);
    }
    const std::string dm = "decoder.model";
    int ch = cfg.dac_decoder_dim;
    if (ch % 32) return fail("decoder_dim must be a multiple of 32");
    CKI(fpack(dm + ".0.conv.weight", &dconv0_w, rup(ch, 128), ch, 7 * C, 7 * C, st));          // (Co, 7*Ci)
    CKI(fvec(dm + ".0.conv.bias", &dconv0_b, ch, st));
    const int nr = cfg.dac_n_rates;
    for (int i = 0; i < nr; ++i) {
      DacBlock Bk{};
      Bk.ci = ch >> i; Bk.co = ch >> (i + 1); Bk.r = cfg.dac_rates[i];
      if (Bk.co % 32) return fail("decoder channel counts must be multiples of 32");
      const std::string bp = dm + "." + std::to_string(i + 1) + ".block";
      CKI(fvec(bp + ".0.alpha", &Bk.alpha, Bk.ci, st));
      CKI(fpack(bp + ".1.conv.weight", &Bk.wt, rup(Bk.r * Bk.co, 128), Bk.r * Bk.co, 2 * Bk.ci, 2 * Bk.ci, st));  // (r*Co, 2*Ci)
      CKI(fvec(bp + ".1.conv.bias", &Bk.bt, Bk.co, st));
      for (int j = 0; j < 3; ++j) {
        const std::string rp = bp + "." + std::to_string(2 + j) + ".block";
        DacRU& ru = Bk.ru[j];
        CKI(fvec(rp + ".0.alpha", &ru.a0, Bk.co, st));
        CKI(fpack(rp + ".1.conv.weight", &ru.w7, rup(Bk.co, 128), Bk.co, 7 * Bk.co, 7 * Bk.co, st));
        CKI(fvec(rp + ".1.conv.bias", &ru.b7, Bk.co, st));
        CKI(fvec(rp + ".2.alpha", &ru.a1, Bk.co, st));
        CKI(fpack(rp + ".3.conv.weight", &ru.w1, rup(Bk.co, 128), Bk.co, Bk.co, Bk.co, st));
        CKI(fvec(rp + ".3.conv.bias", &ru.b1, Bk.co, st));
      }
      dblocks.push_back(Bk);
    }
    const int cl = ch >> nr;
    CKI(fvec(dm + "." + std::to_string(nr + 1) + ".alpha", &dfinal_alpha, cl, st));
    CKI(fpack(dm + "." + std::to_string(nr + 2) + ".conv.weight", &dout_w, 7, 7, cl, cl, st));    // (7, C)
    {
      const RawTensor* r = find(dm + "." + std::to_string(nr + 2) + ".conv.bias");
      if (!r || r->numel != 1) return fail("missing decoder output bias");
      float* tmp; CK(alloc_zero((void**)&tmp, 16));
      CK(launch_convert_any(r->d, r->dtype, tmp, ECHO_F32, 1, st));
      CK(hipStreamSynchronize(st));
      CK(hipMemcpy(&dout_b, tmp, sizeof(float), hipMemcpyDeviceToHost));
    }
    pca_kpad = (int)rup(cfg.latent_size, 32);
    CK(alloc_zero((void**)&pca_w, (size_t)rup(C, 128) * pca_kpad * sizeof(float)));
    CK(alloc_zero((void**)&pca_b, (size_t)rup(C, 4) * sizeof(float)));
    // the decoder's conv weights in the SPLIT3 kernels' pre-split format (gemm.hip, GemmArgs.w_presplit): they are GEMM W operands only
    if (dac_split3) {
      CK(launch_presplit_w(dconv0_w, rup(cfg.dac_decoder_dim, 128), 7L * C, st));
      for (auto& Bk : dblocks) {
        CK(launch_presplit_w(Bk.wt, rup(Bk.r * Bk.co, 128), 2L * Bk.ci, st));
        for (auto& ru : Bk.ru) {
          CK(launch_presplit_w(ru.w7, rup(Bk.co, 128), 7L * Bk.co, st));
          CK(launch_presplit_w(ru.w1, rup(Bk.co, 128), Bk.co, st));
        }
      }
    }
    CK(hipStreamSynchronize(st));
    drop_raw({"quantizer.", "decoder."});
    dac_ready = true;
    return ECHO_OK;
  }

  bool dac_split3 = getenv("ECHO_DAC_EXACT_FP32") ? atoi(getenv("ECHO_DAC_EXACT_FP32")) == 0 : true;
  // Snake of the conv tails on v_sin_f32 (common.h sin_fast) unless ECHO_DAC_FAST_SIN=0: full-size decode of 640 frames, waveform against the
  // sinf() build: 4.8e-8 RMS / 2.5e-7 max on a 1.05e-2 RMS signal (tools/dac_sin_check.py) - the size of the split-3 path's own distance to the
  // CPU restatement, three orders inside the 1e-4 waveform tolerance; sinf() was ~38 VALU instructions per output element of every conv
  bool dac_fast_sin = getenv("ECHO_DAC_FAST_SIN") ? atoi(getenv("ECHO_DAC_FAST_SIN")) != 0 : true;
  DevBuf b_sink;      // GemmArgs.sink: where the branch-free conv tails send the stores of their idle threads
  int frun(const GemmArgs& g_in, hipStream_t st) {
    GemmArgs g = g_in;
    g.split3 = dac_split3 ? 1 : 0;
    if (!b_sink.p) CK(b_sink.reserve(262144));
    g.sink = b_sink.p;
    g.snake_fast = dac_fast_sin ? 1 : 0;
    CKI(plan_gemm<float>(g, st));
    return run_planned<float>(g, st, false);
  }
  static GemmArgs FG(const float* A, long lda, const float* W, long ldw, float* C, long ldc, long M, int N, int K) {
    GemmArgs g;
    gemm_args_init(&g);
    g.A = A; g.lda = lda; g.W = W; g.ldw = ldw; g.C = C; g.ldc = ldc; g.M = (int)M; g.N = N; g.K = K;
    g.Npad = (int)rup(N, 128);
    return g;
  }

  // PCA inverse operands (inference.py:228): w = componentsᵀ as (C, latent) row-major, mean (C)
  int set_pca(const float* w, const float* mean, int on_device, hipStream_t st) override {
    if (!dac_ready) return fail("echo_finalize_dac was not called");
    const int C = cfg.dac_latent_dim, Lz = cfg.latent_size;
    const hipMemcpyKind kind = on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    CK(hipMemcpy2DAsync(pca_w, pca_kpad * sizeof(float), w, Lz * sizeof(float), Lz * sizeof(float), C, kind, st));
    CK(hipMemcpyAsync(pca_b, mean, C * sizeof(float), kind, st));
    CK(hipStreamSynchronize(st));
    pca_set = true;
    return ECHO_OK;
  }
  bool pca_set = false;
  int dac_decode(const float* lat, int Tn, float latent_scale, float* wav, hipStream_t st) override {
    if (!pca_set) return fail("echo_set_pca was not called");
    return dac_run(lat, nullptr, Tn, latent_scale, wav, st);
  }
  int dac_decode_zq(const float* z, int Tn, float* wav, hipStream_t st) override { return dac_run(nullptr, z, Tn, 1.0f, wav, st); }
  // streaming decode: the post_module transformer sees all Tn frames (its stacked causal windows reach back
  // 8 x 127 frames), the convolutional stack (95 % of the FLOPs) only runs on frames f0.. and wav receives (Tn - f0) * hop * up
  // samples; every conv is causal (autoencoder.py:264-331), so the samples further than the stack's receptive field (< 10
  // frames) from frame f0 equal the whole-utterance decode.  The caller drops the context frames.
  int dac_decode_tail(const float* lat, int Tn, int f0, float latent_scale, float* wav, hipStream_t st) override {
    if (!pca_set) return fail("echo_set_pca was not called");
    if (f0 < 0 || f0 >= Tn) return fail("bad first frame");
    return dac_run(lat, nullptr, Tn, latent_scale, wav, st, f0);
  }

  // WindowLimitedTransformer layers on channels-last x (Tn, C), in place; scratch: xn_buf, ao_buf (>= Tp*C + Tp*ff floats
  // and Tn*C floats).  autoencoder.py:786-802, 590-626, 663-717.  The final RMSNorm is applied by the caller.
  // `B` items of `Tn` frames each, stacked as B * Tn rows: every row-wise GEMM / norm sees all rows at once (M = 15360 instead of 24 x 640
  // for a bench call), the attention is per (item, head) through the two-level batch strides of the GEMM descriptor
  int dac_transformer(std::vector<DacLayer>& layers, float* x, int Tn, int C, int nh, int hd, int ff, int window, float* xn_buf,
                      float* ao_buf, hipStream_t st, int B = 1) {
    if (!ae_rope || Tn > ae_rope_npos) return fail("ae rope table missing or too short");
    if (hd != 64 || C != nh * hd || ff % 64) return fail("unsupported transformer sizes");
    const int R = B * Tn;
    const int Tp = (int)rup(R, 128);
    const long ldq = 3L * nh * hd;
    const int Tk = (int)rup(Tn, 32);
    const int vld = (int)rup(Tn, 64);
    const long vt_item = (long)nh * hd * vld;
    CK(b_dq.reserve((size_t)(Tp + 128) * ldq * sizeof(float)));
    CK(b_dscore.reserve((size_t)B * nh * Tn * Tk * sizeof(float)));
    CK(b_dvt.reserve((size_t)(B * vt_item + (long)hd * vld + 128L * vld) * sizeof(float)));
    float *xn = xn_buf, *qkv = b_dq.as<float>(), *sc = b_dscore.as<float>(), *vt = b_dvt.as<float>(), *ao = ao_buf, *hh = xn_buf + (long)Tp * C;
    for (auto& L : layers) {
      CK(launch_norm<float>(NORM_AE_RMS, x, C, xn, C, R, C, cfg.dac_norm_eps, L.an, nullptr, st));
      CKI(frun(FG(xn, C, L.wqkv, C, qkv, ldq, R, 3 * nh * hd, C), st));
      CK(launch_ae_rope(qkv, ldq, R, Tn, nh, hd, ae_rope, st));
      CK(launch_ae_rope(qkv + nh * hd, ldq, R, Tn, nh, hd, ae_rope, st));
      CK(launch_transpose_heads<float>(qkv + 2 * nh * hd, ldq, vt, vld, vt_item, B, Tn, nh, hd, st));
      {
        GemmArgs g = FG(qkv, ldq, qkv + nh * hd, ldq, sc, Tk, Tn, Tk, hd);
        g.nbatch = B * nh; g.nbi = nh;
        g.a_bo = (long)Tn * ldq; g.a_bi = hd; g.w_bo = (long)Tn * ldq; g.w_bi = hd; g.c_bo = (long)nh * Tn * Tk; g.c_bi = (long)Tn * Tk;
        g.acc_scale = 1.0f / sqrtf((float)hd);
        CKI(frun(g, st));
      }
      CK(launch_softmax_f32(sc, Tk, Tn, B * nh, Tn, Tk, nullptr, 0, 1, 1, window, st));
      {
        GemmArgs g = FG(sc, Tk, vt, vld, ao, C, Tn, hd, Tk);
        g.nbatch = B * nh; g.nbi = nh;
        g.a_bo = (long)nh * Tn * Tk; g.a_bi = (long)Tn * Tk; g.w_bo = vt_item; g.w_bi = (long)hd * vld; g.c_bo = (long)Tn * C; g.c_bi = hd;
        CKI(frun(g, st));
      }
      { GemmArgs g = FG(ao, C, L.wo, nh * hd, x, C, R, C, nh * hd); g.colscale = L.ga; g.res = x; g.ldres = C; CKI(frun(g, st)); }
      CK(launch_norm<float>(NORM_AE_RMS, x, C, xn, C, R, C, cfg.dac_norm_eps, L.fn, nullptr, st));
      { GemmArgs g = FG(xn, C, L.w13, C, hh, ff, R, 2 * ff, C); g.swiglu = 1; CKI(frun(g, st)); }
      { GemmArgs g = FG(hh, ff, L.w2, ff, x, C, R, C, ff); g.colscale = L.gf; g.res = x; g.ldres = C; CKI(frun(g, st)); }
    }
    return ECHO_OK;
  }

  // ------------------------------------------------------------------ Fish S1-DAC encode (speaker reference -> latents)
  // inference.py:218-224 ae_encode -> DAC.encode_zq (autoencoder.py:1080-1126): Encoder (autoencoder.py:903-929), quantizer
  // downsample + pre_module + semantic VQ + residual VQs (autoencoder.py:451-464, 184-220, 130-158), from_codes, PCA.
  // Same building blocks as the decoder: channels-last fp32 rows, every conv a taps-GEMM (a stride-s conv with k = 2s reads
  // the input as rows of s*Ci and becomes a 2-tap GEMM), Snake fused into the producing epilogue.
  struct EncRU { float *a0, *w7, *b7, *a1, *w1, *b1; };
  struct EncBlock { EncRU ru[3]; float *alpha, *wc, *bc; int ci, co, r; std::vector<DacLayer> tl; float* tnorm = nullptr; };
  struct VQ { float *in_w, *in_b, *cbn, *cb, *out_w, *out_nb; int size; };
  std::vector<EncBlock> eblocks;
  std::vector<DacUp> ddowns;
  std::vector<DacLayer> dpre; float* dpre_norm = nullptr;
  std::vector<VQ> vqs;
  float *econv0_w = nullptr, *econv0_b = nullptr, *efinal_alpha = nullptr, *econvL_w = nullptr, *econvL_b = nullptr;
  float *fc_w = nullptr, *fc_b = nullptr, *pcae_w = nullptr, *pcae_b = nullptr, *pcae_scale = nullptr;
  int fc_kpad = 0;
  bool enc_ready = false, pcae_set = false;
  DevBuf b_vq_e, b_vq_zst, b_vq_g, b_vq_idx;

  int load_dac_layers(const std::string& pre, int layers, int C, int nh, int hd, int ff, std::vector<DacLayer>& out, float** fnorm,
                      hipStream_t st) {
    for (int i = 0; i < layers; ++i) {
      const std::string lp = pre + ".layers." + std::to_string(i);
      DacLayer L{};
      CKI(fpack(lp + ".attention.wqkv.weight", &L.wqkv, rup(3 * nh * hd, 128), 3 * nh * hd, C, C, st));
      CKI(fpack(lp + ".attention.wo.weight", &L.wo, rup(C, 128), C, nh * hd, nh * hd, st));
      CKI(fpack(lp + ".feed_forward.w1.weight", &L.w13, rup(2 * ff, 128), ff, C, C, st, 0));
      CKI(fpack(lp + ".feed_forward.w3.weight", &L.w13, rup(2 * ff, 128), ff, C, C, st, 1, &L.w13));
      CKI(fpack(lp + ".feed_forward.w2.weight", &L.w2, rup(C, 128), C, ff, ff, st));
      CKI(fvec(lp + ".attention_norm.weight", &L.an, C, st));
      CKI(fvec(lp + ".ffn_norm.weight", &L.fn, C, st));
      CKI(fvec(lp + ".attention_layer_scale.gamma", &L.ga, C, st));
      CKI(fvec(lp + ".ffn_layer_scale.gamma", &L.gf, C, st));
      out.push_back(L);
    }
    CKI(fvec(pre + ".norm.weight", fnorm, C, st));
    return ECHO_OK;
  }

  int finalize_dac_encoder(hipStream_t st) override {
    if (!dac_ready) return fail("echo_finalize_dac must be called first");
    const int C = cfg.dac_latent_dim, nh = cfg.dac_post_heads, hd = cfg.dac_post_head_dim, ff = cfg.dac_post_ffn;
    const int C0 = cfg.dac_enc_dim, nr = cfg.dac_enc_n_rates, cd = cfg.dac_codebook_dim;
    if (C0 <= 0 || C0 % 32 || nr < 1 || nr > 8 || cd != 8) return fail("unsupported DAC encoder sizes (channels % 32, codebook_dim 8)");
    CKI(fpack("enc.conv0.w", &econv0_w, C0, C0, 7, 7, st));      // (C0, 7)
    CKI(fvec("enc.conv0.b", &econv0_b, C0, st));
    int ch = C0;
    for (int i = 0; i < nr; ++i) {
      EncBlock B{};
      B.ci = ch; B.co = 2 * ch; B.r = cfg.dac_enc_rates[i];
      const std::string bp = "enc.b" + std::to_string(i);
      for (int j = 0; j < 3; ++j) {
        const std::string rp = bp + ".ru" + std::to_string(j);
        EncRU& ru = B.ru[j];
        CKI(fvec(rp + ".a0", &ru.a0, B.ci, st));
        CKI(fpack(rp + ".w7", &ru.w7, rup(B.ci, 128), B.ci, 7 * B.ci, 7 * B.ci, st));
        CKI(fvec(rp + ".b7", &ru.b7, B.ci, st));
        CKI(fvec(rp + ".a1", &ru.a1, B.ci, st));
        CKI(fpack(rp + ".w1", &ru.w1, rup(B.ci, 128), B.ci, B.ci, B.ci, st));
        CKI(fvec(rp + ".b1", &ru.b1, B.ci, st));
      }
      CKI(fvec(bp + ".alpha", &B.alpha, B.ci, st));
      CKI(fpack(bp + ".wc", &B.wc, rup(B.co, 128), B.co, 2 * B.r * B.ci, 2L * B.r * B.ci, st));
      CKI(fvec(bp + ".bc", &B.bc, B.co, st));
      if (cfg.dac_enc_tlayers[i] > 0) {
        if (B.co % 64) return fail("encoder transformer width must be a multiple of 64");
        CKI(load_dac_layers(bp + ".t", cfg.dac_enc_tlayers[i], B.co, B.co / 64, 64, 3 * B.co, B.tl, &B.tnorm, st));
      }
      eblocks.push_back(B);
      ch *= 2;
    }
    CKI(fvec("enc.final.alpha", &efinal_alpha, ch, st));
    CKI(fpack("enc.final.w", &econvL_w, rup(C, 128), C, 3 * ch, 3L * ch, st));
    CKI(fvec("enc.final.b", &econvL_b, C, st));
    for (int i = 0; i < cfg.dac_n_up; ++i) {
      const std::string dn = "enc.down" + std::to_string(i);
      DacUp U{};
      U.f = cfg.dac_up_factors[i];
      CKI(fpack(dn + ".w", &U.w, rup(C, 128), C, U.f * C, (long)U.f * C, st));
      CKI(fvec(dn + ".b", &U.b, C, st));
      CKI(fpack(dn + ".dw", &U.dw, C, C, 7, 7, st));
      CKI(fvec(dn + ".db", &U.db, C, st));
      CKI(fvec(dn + ".lnw", &U.lnw, C, st));
      CKI(fvec(dn + ".lnb", &U.lnb, C, st));
      CKI(fpack(dn + ".p1w", &U.p1w, rup(4 * C, 128), 4 * C, C, C, st));
      CKI(fvec(dn + ".p1b", &U.p1b, 4 * C, st));
      CKI(fpack(dn + ".p2w", &U.p2w, rup(C, 128), C, 4 * C, 4 * C, st));
      CKI(fvec(dn + ".p2b", &U.p2b, C, st));
      CKI(fvec(dn + ".gamma", &U.gamma, C, st));
      ddowns.push_back(U);
    }
    CKI(load_dac_layers("enc.pre", cfg.dac_post_layers, C, nh, hd, ff, dpre, &dpre_norm, st));
    for (int q = 0; q <= cfg.dac_n_codebooks; ++q) {
      const std::string vp = "enc.vq" + std::to_string(q);
      VQ v{};
      v.size = q == 0 ? cfg.dac_semantic_size : cfg.dac_codebook_size;
      CKI(fpack(vp + ".in_w", &v.in_w, 128, cd, C, C, st));
      CKI(fvec(vp + ".in_b", &v.in_b, cd, st));
      CKI(fpack(vp + ".cbn", &v.cbn, v.size, v.size, cd, cd, st));
      CKI(fpack(vp + ".cb", &v.cb, v.size, v.size, cd, cd, st));
      CKI(fpack(vp + ".out_w", &v.out_w, rup(C, 128), C, cd, 32, st));      // K padded 8 -> 32
      CKI(fvec(vp + ".out_nb", &v.out_nb, C, st));
      vqs.push_back(v);
    }
    fc_kpad = (int)rup((cfg.dac_n_codebooks + 1) * cd, 32);
    CKI(fpack("enc.fc.w", &fc_w, rup(C, 128), C, (cfg.dac_n_codebooks + 1) * cd, fc_kpad, st));
    CKI(fvec("enc.fc.b", &fc_b, C, st));
    CK(alloc_zero((void**)&pcae_w, (size_t)128 * C * sizeof(float)));
    CK(alloc_zero((void**)&pcae_b, (size_t)128 * sizeof(float)));
    CK(alloc_zero((void**)&pcae_scale, (size_t)128 * sizeof(float)));
    CK(hipStreamSynchronize(st));
    drop_raw({"enc."});
    enc_ready = true;
    return ECHO_OK;
  }

  int set_pca_encode(const float* w, const float* bias, float scale, int on_device, hipStream_t st) override {
    if (!enc_ready) return fail("echo_finalize_dac_encoder was not called");
    const int C = cfg.dac_latent_dim, Lz = cfg.latent_size;
    if (Lz > 128 || (Lz & 3)) return fail("unsupported latent_size");
    const hipMemcpyKind kind = on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    CK(hipMemcpyAsync(pcae_w, w, (size_t)Lz * C * sizeof(float), kind, st));
    CK(hipMemcpyAsync(pcae_b, bias, (size_t)Lz * sizeof(float), kind, st));
    std::vector<float> sc(128, scale);
    CK(hipMemcpyAsync(pcae_scale, sc.data(), 128 * sizeof(float), hipMemcpyHostToDevice, st));
    CK(hipStreamSynchronize(st));
    pcae_set = true;
    return ECHO_OK;
  }

  int dac_encode(const float* audio, long n, float* lat_out, int32_t* codes_out, float* zq_out, hipStream_t st) override {
    if (!enc_ready) return fail("echo_finalize_dac_encoder was not called");
    if (!pcae_set && lat_out) return fail("echo_set_pca_encode was not called");
    const int C = cfg.dac_latent_dim, nh = cfg.dac_post_heads, hd = cfg.dac_post_head_dim, ff = cfg.dac_post_ffn;
    const int C0 = cfg.dac_enc_dim, cd = cfg.dac_codebook_dim;
    long frame = 1;
    for (auto& b : eblocks) frame *= b.r;
    for (auto& d : ddowns) frame *= d.f;
    if (n <= 0 || n % frame) return fail("audio length must be a positive multiple of the frame length");
    if (n * C0 > (1L << 31) - 4096) return fail("audio chunk too long");
    const long PADF = 64L * std::max(std::max(cfg.dac_decoder_dim, 4 * C), 16 * C0);
    long maxel = std::max(n * (long)C0, (long)(n / frame) * 4L * 4 * C);
    {
      long rows = n; int ch = C0;
      for (auto& b : eblocks) { rows /= b.r; ch *= 2; maxel = std::max(maxel, rows * (long)ch * 4); }   // transformer scratch: x4
    }
    const size_t bytes = (size_t)(maxel + PADF + 128L * 4 * C) * sizeof(float);
    CK(b_dacA.reserve(bytes)); CK(b_dacB.reserve(bytes)); CK(b_dacC.reserve(bytes));
    float *Y = b_dacA.as<float>() + PADF, *S = b_dacB.as<float>() + PADF, *U = b_dacC.as<float>() + PADF;
    // ---- Encoder head: conv k7 1 -> C0, + Snake of the first ResidualUnit
    CK(launch_conv_in_snake(audio, n, C0, 7, econv0_w, econv0_b, eblocks[0].ru[0].a0, Y, S, C0, st));
    long rows = n;
    const int dil[3] = {1, 3, 9};
    for (size_t bi = 0; bi < eblocks.size(); ++bi) {
      EncBlock& Bk = eblocks[bi];
      const int ci = Bk.ci, co = Bk.co, r = Bk.r;
      for (int j = 0; j < 3; ++j) {
        EncRU& ru = Bk.ru[j];
        {   // Snake -> conv k7 (dilated) -> Snake, only the Snake'd result is kept
          GemmArgs g = FG(S, ci, ru.w7, 7L * ci, U, ci, rows, ci, ci);
          g.taps = 7; g.tap_base = -6 * dil[j]; g.tap_shift = dil[j]; g.bias = ru.b7;
          g.store_main = 0; g.C2 = U; g.snake_alpha = ru.a1;
          CKI(frun(g, st));
        }
        {   // conv k1 + residual; second output = Snake with the next consumer's alpha
          GemmArgs g = FG(U, ci, ru.w1, ci, Y, ci, rows, ci, ci);
          g.bias = ru.b1; g.res = Y; g.ldres = ci;
          g.store_main = j < 2 ? 1 : 0; g.C2 = S; g.snake_alpha = j < 2 ? Bk.ru[j + 1].a0 : Bk.alpha;
          CKI(frun(g, st));
        }
      }
      {   // conv k = 2r, stride r, ci -> co: rows of r*ci, taps {row j-1, row j}
        const long rows2 = rows / r;
        GemmArgs g = FG(S, (long)r * ci, Bk.wc, 2L * r * ci, Y, co, rows2, co, r * ci);
        g.taps = 2; g.tap_base = -1; g.tap_shift = 1; g.bias = Bk.bc;
        if (bi + 1 < eblocks.size() && Bk.tl.empty()) { g.C2 = U; g.snake_alpha = eblocks[bi + 1].ru[0].a0; }
        CKI(frun(g, st));
        rows = rows2;
        if (g.C2) std::swap(S, U);
      }
      if (!Bk.tl.empty()) {
        CKI(dac_transformer(Bk.tl, Y, (int)rows, co, co / 64, 64, 3 * co, cfg.dac_enc_window, S, U, st));
        CK(launch_norm<float>(NORM_AE_RMS, Y, co, S, co, (int)rows, co, cfg.dac_norm_eps, Bk.tnorm, nullptr, st));
        std::swap(Y, S);
        if (bi + 1 < eblocks.size()) CK(launch_snake_f32(Y, co, S, co, rows, co, eblocks[bi + 1].ru[0].a0, st));
      }
    }
    const int chL = eblocks.back().co;
    CK(launch_snake_f32(Y, chL, S, chL, rows, chL, efinal_alpha, st));
    {   // conv k3 -> latent_dim
      GemmArgs g = FG(S, chL, econvL_w, 3L * chL, Y, C, rows, C, chL);
      g.taps = 3; g.tap_base = -2; g.tap_shift = 1; g.bias = econvL_b;
      CKI(frun(g, st));
    }
    // ---- quantizer.downsample: [conv k = f stride f ; ConvNeXt] (autoencoder.py:418-426, 360-373)
    for (auto& D : ddowns) {
      const long rows2 = rows / D.f;
      { GemmArgs g = FG(Y, (long)D.f * C, D.w, (long)D.f * C, S, C, rows2, C, D.f * C); g.bias = D.b; CKI(frun(g, st)); }
      rows = rows2;
      CK(launch_dwconv_ln(S, C, U, C, (int)rows, (int)rows, C, D.dw, D.db, D.lnw, D.lnb, 1e-6f, st));
      { GemmArgs g = FG(U, C, D.p1w, C, Y, 4L * C, rows, 4 * C, C); g.bias = D.p1b; g.act = 2; CKI(frun(g, st)); }
      { GemmArgs g = FG(Y, 4L * C, D.p2w, 4L * C, S, C, rows, C, 4 * C); g.bias = D.p2b; g.colscale = D.gamma; g.res = S; g.ldres = C; CKI(frun(g, st)); }
      std::swap(Y, S);
    }
    // ---- pre_module
    const int Tn = (int)rows;
    CKI(dac_transformer(dpre, Y, Tn, C, nh, hd, ff, cfg.dac_post_window, S, U, st));
    CK(launch_norm<float>(NORM_AE_RMS, Y, C, S, C, Tn, C, cfg.dac_norm_eps, dpre_norm, nullptr, st));
    std::swap(Y, S);
    // ---- semantic VQ + residual VQs: residual <- residual - out_proj(straight-through code) after every quantizer
    const int nq = (int)vqs.size();
    CK(b_vq_e.reserve((size_t)(Tn + 256) * cd * sizeof(float)));
    CK(b_vq_zst.reserve((size_t)(Tn + 256) * 32 * sizeof(float)));
    CK(b_vq_g.reserve((size_t)(Tn + 256) * fc_kpad * sizeof(float)));
    CK(b_vq_idx.reserve((size_t)nq * Tn * sizeof(int)));
    float *e8 = b_vq_e.as<float>(), *zst = b_vq_zst.as<float>(), *gath = b_vq_g.as<float>();
    int* idx = b_vq_idx.as<int>();
    for (int q = 0; q < nq; ++q) {
      VQ& v = vqs[q];
      { GemmArgs g = FG(Y, C, v.in_w, C, e8, cd, Tn, cd, C); g.bias = v.in_b; CKI(frun(g, st)); }
      CK(launch_vq_argmax(e8, cd, Tn, v.cbn, v.cb, v.size, idx + (long)q * Tn, zst, 32, gath + q * cd, fc_kpad, st));
      { GemmArgs g = FG(zst, 32, v.out_w, 32, Y, C, Tn, C, 32); g.acc_scale = -1.0f; g.bias = v.out_nb; g.res = Y; g.ldres = C; CKI(frun(g, st)); }
    }
    // ---- encode_zq: sum over quantizers of out_proj(codebook[code]) as one GEMM over the gathered code vectors
    { GemmArgs g = FG(gath, fc_kpad, fc_w, fc_kpad, S, C, Tn, C, fc_kpad); g.bias = fc_b; CKI(frun(g, st)); }
    if (zq_out) CK(hipMemcpyAsync(zq_out, S, (size_t)Tn * C * sizeof(float), hipMemcpyDeviceToDevice, st));
    if (codes_out) CK(hipMemcpyAsync(codes_out, idx, (size_t)nq * Tn * sizeof(int), hipMemcpyDeviceToDevice, st));
    if (lat_out) {   // ((z_q - mean) @ P^T) * scale, mean folded into the bias (inference.py:221-223)
      GemmArgs g = FG(S, C, pcae_w, C, lat_out, cfg.latent_size, Tn, cfg.latent_size, C);
      g.bias = pcae_b; g.colscale = pcae_scale;
      CKI(frun(g, st));
    }
    return ECHO_OK;
  }

  // lat: (T, latent) latents (PCA applied here)  OR  zq: (T, C) channels-last quantizer output
  // ---- the front of the decoder (PCA inverse, post_module transformer, its final norm) for B items at once: B * Tn rows through every
  // row-wise GEMM (a 24-item call: M = 15360 instead of 24 launches at M = 640 with split-K reductions), attention per (item, head).
  // quantizer.upsample rides along; item b's upsampled rows end up at dac_front_out + b * dac_front_item_rows * C and
  // dac_run(..., front = that pointer) runs the decoder convolutions on them.
  DevBuf b_dfx, b_dfs, b_dfu, b_dfn;
  float dac_front_ms = 0.f;         // profiling: the last batched front's time (dac_decode_batch divides it over its items)
  float* dac_front_out = nullptr;   // item b's upsampled rows: dac_front_out + b * dac_front_item_rows * C
  long dac_front_item_rows = 0;
  int dac_front_batch(const float* lat, int B, int Tn, float latent_scale, hipStream_t st) {
    if (!dac_ready) return fail("echo_finalize_dac was not called");
    if (!pca_set) return fail("echo_set_pca was not called");
    const int C = cfg.dac_latent_dim, nh = cfg.dac_post_heads, hd = cfg.dac_post_head_dim, ff = cfg.dac_post_ffn;
    const long R = (long)B * Tn, Tp = rup(R, 128);
    long upf = 1;
    for (int i = 0; i < cfg.dac_n_up; ++i) upf *= cfg.dac_up_factors[i];
    const long Rup = rup(R * upf, 128) + 128;       // rows after quantizer.upsample
    CK(b_dfx.reserve((size_t)Rup * C * sizeof(float)));
    CK(b_dfs.reserve((size_t)std::max((Tp + 128) * (long)(C + ff), Rup * 4L * C) * sizeof(float)));
    CK(b_dfu.reserve((size_t)(Tp + 128) * C * sizeof(float)));
    CK(b_dfn.reserve((size_t)Rup * C * sizeof(float)));
    CK(b_dmisc.reserve((size_t)Tp * pca_kpad * sizeof(float)));
    if (profiling) {
      for (auto& e : ev) if (!e) CK(hipEventCreate(&e));
      CK(hipEventRecord(ev[4], st));
    }
    float* x = b_dfx.as<float>();
    CK(launch_pca_prep(lat, b_dmisc.as<float>(), pca_kpad, R, cfg.latent_size, pca_kpad, latent_scale, st));
    {
      GemmArgs g = FG(b_dmisc.as<float>(), pca_kpad, pca_w, pca_kpad, x, C, R, C, pca_kpad);
      g.bias = pca_b;
      CKI(frun(g, st));
    }
    CKI(dac_transformer(dpost, x, Tn, C, nh, hd, ff, cfg.dac_post_window, b_dfs.as<float>(), b_dfu.as<float>(), st, B));
    CK(launch_norm<float>(NORM_AE_RMS, x, C, b_dfn.as<float>(), C, (int)R, C, cfg.dac_norm_eps, dpost_norm, nullptr, st));
    // quantizer.upsample on the stacked rows as well (autoencoder.py:427-435, 360-373): the ConvTranspose is a 1-tap GEMM, the ConvNeXt's
    // causal depthwise conv restarts its padding per item (dwconv_ln_kernel: position = row % rows_per_item), the rest is row-wise
    {
      float *cur = b_dfn.as<float>(), *other = x, *third = b_dfs.as<float>();
      long rows = R, per_item = Tn;
      for (auto& U : dups) {
        { GemmArgs g = FG(cur, C, U.w, C, other, (long)U.f * C, rows, U.f * C, C); g.bias = U.b; g.vec_mod = C; CKI(frun(g, st)); }
        rows *= U.f; per_item *= U.f;
        CK(launch_dwconv_ln(other, C, cur, C, (int)rows, (int)per_item, C, U.dw, U.db, U.lnw, U.lnb, 1e-6f, st));
        { GemmArgs g = FG(cur, C, U.p1w, C, third, 4L * C, rows, 4 * C, C); g.bias = U.p1b; g.act = 2; CKI(frun(g, st)); }
        { GemmArgs g = FG(third, 4L * C, U.p2w, 4L * C, other, C, rows, C, 4 * C); g.bias = U.p2b; g.colscale = U.gamma; g.res = other; g.ldres = C; CKI(frun(g, st)); }
        std::swap(cur, other);
      }
      dac_front_out = cur; dac_front_item_rows = per_item;
    }
    if (profiling) {
      CK(hipEventRecord(ev[5], st));
      CK(hipEventSynchronize(ev[5]));
      CK(hipEventElapsedTime(&dac_front_ms, ev[4], ev[5]));
    }
    return ECHO_OK;
  }
  int dac_decode_batch(const float* lat, int B, int Tn, float latent_scale, float* wav, long wav_stride, hipStream_t st) override {
    if (B < 1 || Tn < 1) return fail("dac_decode_batch: empty batch");
    if (B == 1) return dac_decode(lat, Tn, latent_scale, wav, st);
    CKI(dac_front_batch(lat, B, Tn, latent_scale, st));
    float ms = 0.f, ms_gemm = 0.f;
    for (int b = 0; b < B; ++b) {
      CKI(dac_run(nullptr, nullptr, Tn, 1.0f, wav + (long)b * wav_stride, st, 0, dac_front_out + (long)b * dac_front_item_rows * cfg.dac_latent_dim));
      ms += prof.ms_total; ms_gemm += prof.ms_gemm_sum;
    }
    if (profiling) {     // per-item averages incl. the shared front (the front's GEMM launches are not in ms_gemm_sum)
      prof.ms_total = (ms + dac_front_ms) / B; prof.ms_steps = prof.ms_total; prof.ms_gemm_sum = ms_gemm / B;
    }
    return ECHO_OK;
  }

  int dac_run(const float* lat, const float* zq, int Tn, float latent_scale, float* wav, hipStream_t st, int f0 = 0, const float* front = nullptr) {
    if (!dac_ready) return fail("echo_finalize_dac was not called");
    if (!ae_rope || Tn > ae_rope_npos) return fail("ae rope table missing or too short");
    const int C = cfg.dac_latent_dim, nh = cfg.dac_post_heads, hd = cfg.dac_post_head_dim, ff = cfg.dac_post_ffn;
    long hop = 1;
    for (int i = 0; i < cfg.dac_n_rates; ++i) hop *= cfg.dac_rates[i];
    long upf = 1;
    for (int i = 0; i < cfg.dac_n_up; ++i) upf *= cfg.dac_up_factors[i];
    // largest activation: rows * channels after each decoder block
    long maxel = (long)Tn * upf * std::max(C * 4, cfg.dac_decoder_dim);
    {
      long rows = (long)Tn * upf;
      for (auto& b : dblocks) { rows *= b.r; maxel = std::max(maxel, rows * b.co); }
    }
    const long PADF = 64L * std::max(cfg.dac_decoder_dim, 4 * C);   // zero rows in front of every buffer (causal left padding)
    const size_t bytes = (size_t)(maxel + PADF + 128L * 4 * C) * sizeof(float);
    CK(b_dacA.reserve(bytes)); CK(b_dacB.reserve(bytes)); CK(b_dacC.reserve(bytes));
    float *bufY = b_dacA.as<float>() + PADF, *bufS = b_dacB.as<float>() + PADF, *bufU = b_dacC.as<float>() + PADF;
    if (profiling) {
      for (auto& e : ev) if (!e) CK(hipEventCreate(&e));
      gemm_events_used = 0;
      CK(hipEventRecord(ev[0], st));
    }
    // ---- PCA inverse (inference.py:228): z = (latent / scale) @ components + mean
    const int Tp = (int)rup(Tn, 128);
    CK(b_dmisc.reserve((size_t)Tp * pca_kpad * sizeof(float)));
    float* x = bufY;
    if (front) {
      // the front (incl. quantizer.upsample) already ran for the whole batch (dac_front_batch): this item's rows go to its own buffer,
      // whose zero rows in front are the causal padding of the first decoder conv
      CK(hipMemcpyAsync(bufS, front, (size_t)Tn * upf * C * sizeof(float), hipMemcpyDeviceToDevice, st));
    } else {
    if (lat) {
      CK(launch_pca_prep(lat, b_dmisc.as<float>(), pca_kpad, Tn, cfg.latent_size, pca_kpad, latent_scale, st));
      GemmArgs g = FG(b_dmisc.as<float>(), pca_kpad, pca_w, pca_kpad, x, C, Tn, C, pca_kpad);
      g.bias = pca_b;
      CKI(frun(g, st));
    } else {
      CK(hipMemcpyAsync(x, zq, (size_t)Tn * C * sizeof(float), hipMemcpyDeviceToDevice, st));
    }
    // ---- post_module: window-limited causal transformer (autoencoder.py:786-802)
    CKI(dac_transformer(dpost, x, Tn, C, nh, hd, ff, cfg.dac_post_window, bufS, bufU, st));
    CK(launch_norm<float>(NORM_AE_RMS, x, C, bufS, C, Tn, C, cfg.dac_norm_eps, dpost_norm, nullptr, st));
    }
    // ---- quantizer.upsample: [ConvT k=f s=f ; ConvNeXt] (autoencoder.py:427-435, 360-373)
    // f0 > 0: only frames f0.. go through the convolutions; the first conv's causal taps then read the real frames in front
    // of f0 (still in bufS), every later one the zero rows in front of its buffer
    float* cur = bufS + (long)f0 * C;      // (rows, C)
    float* other = bufY;
    float* third = bufU;
    long rows = front ? (long)Tn * upf : Tn - f0;
    if (!front)
    for (auto& U : dups) {
      { GemmArgs g = FG(cur, C, U.w, C, other, (long)U.f * C, rows, U.f * C, C); g.bias = U.b; g.vec_mod = C; CKI(frun(g, st)); }
      rows *= U.f;
      CK(launch_dwconv_ln(other, C, cur, C, (int)rows, (int)rows, C, U.dw, U.db, U.lnw, U.lnb, 1e-6f, st));
      { GemmArgs g = FG(cur, C, U.p1w, C, third, 4L * C, rows, 4 * C, C); g.bias = U.p1b; g.act = 2; CKI(frun(g, st)); }
      { GemmArgs g = FG(third, 4L * C, U.p2w, 4L * C, other, C, rows, C, 4 * C); g.bias = U.p2b; g.colscale = U.gamma; g.res = other; g.ldres = C; CKI(frun(g, st)); }
      std::swap(cur, other);
    }
    // ---- decoder (autoencoder.py:971-998): every conv is a taps-GEMM on channels-last rows
    float *Y = other, *S = third, *Uu = nullptr;
    {
      // conv k7 C -> ch, epilogue writes only snake_{block1}(y)
      const int ch = cfg.dac_decoder_dim;
      GemmArgs g = FG(cur, C, dconv0_w, 7L * C, Y, ch, rows, ch, C);
      g.w_presplit = dac_split3;
      g.taps = 7; g.tap_base = -6; g.tap_shift = 1; g.bias = dconv0_b;
      g.store_main = 0; g.C2 = S; g.snake_alpha = dblocks[0].alpha;
      CKI(frun(g, st));
      Uu = cur;
    }
    for (size_t bi = 0; bi < dblocks.size(); ++bi) {
      DacBlock& Bk = dblocks[bi];
      {
        // ConvTranspose k=2r s=r as a 2-tap GEMM with N = r*Co; rows of C are r consecutive output steps
        GemmArgs g = FG(S, Bk.ci, Bk.wt, 2L * Bk.ci, Y, (long)Bk.r * Bk.co, rows, Bk.r * Bk.co, Bk.ci);
        g.w_presplit = dac_split3;
        g.taps = 2; g.tap_base = -1; g.tap_shift = 1; g.bias = Bk.bt; g.vec_mod = Bk.co;
        g.C2 = Uu; g.snake_alpha = Bk.ru[0].a0;
        CKI(frun(g, st));
        rows *= Bk.r;
        std::swap(S, Uu);   // S now holds snake_{ru0.a0}(y)
      }
      const int dil[3] = {1, 3, 9};
      for (int j = 0; j < 3; ++j) {
        DacRU& ru = Bk.ru[j];
        {
          GemmArgs g = FG(S, Bk.co, ru.w7, 7L * Bk.co, Uu, Bk.co, rows, Bk.co, Bk.co);
          g.w_presplit = dac_split3;
          g.taps = 7; g.tap_base = -6 * dil[j]; g.tap_shift = dil[j]; g.bias = ru.b7;
          g.store_main = 0; g.C2 = Uu; g.snake_alpha = ru.a1;
          CKI(frun(g, st));
        }
        {
          const float* next_alpha = j < 2 ? Bk.ru[j + 1].a0 : (bi + 1 < dblocks.size() ? dblocks[bi + 1].alpha : dfinal_alpha);
          GemmArgs g = FG(Uu, Bk.co, ru.w1, Bk.co, Y, Bk.co, rows, Bk.co, Bk.co);
          g.w_presplit = dac_split3;
          g.bias = ru.b1; g.res = Y; g.ldres = Bk.co;
          g.store_main = j < 2 ? 1 : 0; g.C2 = S; g.snake_alpha = next_alpha;
          CKI(frun(g, st));
        }
      }
    }
    const int cl = cfg.dac_decoder_dim >> cfg.dac_n_rates;
    CK(launch_conv_out_tanh(S, cl, wav, rows, (int)rows, cl, 7, dout_w, dout_b, st));
    if (profiling) {
      CK(hipEventRecord(ev[2], st));
      CK(hipEventSynchronize(ev[2]));
      CK(hipEventElapsedTime(&prof.ms_total, ev[0], ev[2]));
      prof.ms_mod = 0.f; prof.ms_steps = prof.ms_total;
      collect_gemm_times();
    }
    return ECHO_OK;
  }
};

__global__ void add_seg_bias_kernel(float* __restrict__ dst, long ld, int rows, int off, int w, const float* __restrict__ bias, long bias_ld,
                                    int kv_mod) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)rows * w) return;
  const int r = (int)(i / w), k = (int)(i - (long)r * w);
  const int kr = kv_mod ? r % kv_mod : r;
  dst[(long)r * ld + off + k] += bias[(long)kr * bias_ld + k];
}
template <typename T>
int Engine<T>::add_seg_bias(float* biasrows, long ld, int rows, int off, int w, const float* bias, long bias_ld, int kv_mod, hipStream_t st) {
  hipLaunchKernelGGL(add_seg_bias_kernel, dim3((unsigned)(((long)rows * w + 255) / 256)), dim3(256), 0, st, biasrows, ld, rows, off, w,
                     bias, bias_ld, kv_mod);
  CK(hipGetLastError());
  return ECHO_OK;
}

}  // namespace

// =============================================================================== C ABI
struct echo_ctx {
  std::unique_ptr<EngineBase> eng;
};

extern "C" {

int echo_abi_version(void) { return ECHO_ABI_VERSION; }

const char* echo_last_error(echo_ctx* ctx) { return ctx ? ctx->eng->err.c_str() : g_create_error.c_str(); }

int echo_ctx_create(const echo_config* cfg, int device, echo_ctx** out) {
  if (!cfg || !out) { g_create_error = "null argument"; return ECHO_ERR; }
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0) { g_create_error = "no HIP device available (libechohip has no CPU fallback)"; return ECHO_ERR; }
  if (device < 0 || device >= ndev) { g_create_error = "bad device index"; return ECHO_ERR; }
  e = hipSetDevice(device);
  if (e != hipSuccess) { g_create_error = std::string("hipSetDevice: ") + hipGetErrorString(e); return ECHO_ERR; }
  hipDeviceProp_t prop;
  e = hipGetDeviceProperties(&prop, device);
  if (e != hipSuccess) { g_create_error = std::string("hipGetDeviceProperties: ") + hipGetErrorString(e); return ECHO_ERR; }
  if (std::string(prop.gcnArchName).find("gfx950") == std::string::npos) {
    g_create_error = std::string("libechohip is built for gfx950 only, found ") + prop.gcnArchName;
    return ECHO_ERR;
  }
  echo_ctx* c = new echo_ctx();
  if (cfg->precision == ECHO_BF16) c->eng.reset(new Engine<bf16_t>());
  else if (cfg->precision == ECHO_F32) c->eng.reset(new Engine<float>());
  else { delete c; g_create_error = "bad precision"; return ECHO_ERR; }
  if (cfg->dit_fp8 && cfg->precision != ECHO_BF16) { delete c; g_create_error = "dit_fp8 needs precision ECHO_BF16"; return ECHO_ERR; }
  c->eng->cfg = *cfg;
  c->eng->device = device;
  *out = c;
  return ECHO_OK;
}

void echo_ctx_destroy(echo_ctx* ctx) { delete ctx; }

int echo_load_tensor(echo_ctx* ctx, const char* name, const void* data, int dtype, int ndim, const int64_t* shape, int on_device) {
  if (!ctx || !name || !data) return ECHO_ERR;
  EngineBase& E = *ctx->eng;
  if (dtype != ECHO_F32 && dtype != ECHO_BF16) return E.fail("bad dtype");
  RawTensor t;
  t.dtype = dtype;
  t.numel = 1;
  for (int i = 0; i < ndim; ++i) { t.shape.push_back(shape[i]); t.numel *= shape[i]; }
  const size_t bytes = (size_t)t.numel * (dtype == ECHO_F32 ? 4 : 2);
  hipError_t e = hipMalloc(&t.d, bytes ? bytes : 4);
  if (e != hipSuccess) return E.fail(e, "hipMalloc(raw tensor)");
  e = hipMemcpy(t.d, data, bytes, on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice);
  if (e != hipSuccess) { (void)hipFree(t.d); return E.fail(e, "hipMemcpy(raw tensor)"); }
  auto it = E.raw.find(name);
  if (it != E.raw.end()) { (void)hipFree(it->second.d); E.raw.erase(it); }
  E.raw[name] = t;
  return ECHO_OK;
}

int echo_finalize_dit(echo_ctx* ctx, void* stream) { return ctx ? ctx->eng->finalize_dit((hipStream_t)stream) : ECHO_ERR; }
int echo_finalize_dac(echo_ctx* ctx, void* stream) { return ctx ? ctx->eng->finalize_dac((hipStream_t)stream) : ECHO_ERR; }
int echo_set_rope_table(echo_ctx* ctx, const void* t, int npos) { return ctx ? ctx->eng->set_rope(t, npos) : ECHO_ERR; }
int echo_set_ae_rope_table(echo_ctx* ctx, const void* t, int npos) { return ctx ? ctx->eng->set_ae_rope(t, npos) : ECHO_ERR; }

int echo_encode_text(echo_ctx* ctx, const int32_t* ids, const float* key_bias, const int32_t* nkeys_host, int B, int Tt, void* stream) {
  return ctx ? ctx->eng->encode_text(ids, key_bias, nkeys_host, B, Tt, (hipStream_t)stream) : ECHO_ERR;
}
int echo_encode_speaker(echo_ctx* ctx, const void* latent, const float* key_bias, const int32_t* nkeys_host, int B, int Ts, void* stream) {
  return ctx ? ctx->eng->encode_speaker(latent, key_bias, nkeys_host, B, Ts, (hipStream_t)stream) : ECHO_ERR;
}
int echo_encode_latent_prefix(echo_ctx* ctx, const void* latent, int B, int n_latents, long row_stride, void* stream) {
  return ctx ? ctx->eng->encode_latent(latent, B, n_latents, row_stride, (hipStream_t)stream) : ECHO_ERR;
}
int echo_scale_speaker_kv(echo_ctx* ctx, float scale, int max_layers, void* stream) {
  return ctx ? ctx->eng->scale_speaker_kv(scale, max_layers, (hipStream_t)stream) : ECHO_ERR;
}
int echo_dit_forward(echo_ctx* ctx, const void* x, const void* temb, int rows, int B, int S, int start_pos, int use_latent,
                     const int32_t* ton, const int32_t* son, float* v_out, void* stream) {
  return ctx ? ctx->eng->dit_forward(x, temb, 1, nullptr, rows, B, S, start_pos, use_latent, ton, son, v_out, (hipStream_t)stream) : ECHO_ERR;
}
int echo_dit_forward_t(echo_ctx* ctx, const void* x, const void* temb, int n_t, const int32_t* row_t, int rows, int B, int S, int start_pos,
                       int use_latent, const int32_t* ton, const int32_t* son, float* v_out, void* stream) {
  return ctx ? ctx->eng->dit_forward(x, temb, n_t, row_t, rows, B, S, start_pos, use_latent, ton, son, v_out, (hipStream_t)stream) : ECHO_ERR;
}
int echo_sample_euler(echo_ctx* ctx, const echo_sampler_params* p, const float* x_init, float* latent_out, void* stream) {
  return ctx ? ctx->eng->sample_euler(p, x_init, latent_out, (hipStream_t)stream) : ECHO_ERR;
}
int echo_dac_decode(echo_ctx* ctx, const float* latent, int T, float latent_scale, float* wav_out, void* stream) {
  return ctx ? ctx->eng->dac_decode(latent, T, latent_scale, wav_out, (hipStream_t)stream) : ECHO_ERR;
}
int echo_dac_decode_batch(echo_ctx* ctx, const float* latent, int B, int T, float latent_scale, float* wav_out, int64_t wav_stride, void* stream) {
  return ctx ? ctx->eng->dac_decode_batch(latent, B, T, latent_scale, wav_out, (long)wav_stride, (hipStream_t)stream) : ECHO_ERR;
}
int echo_dac_decode_zq(echo_ctx* ctx, const float* z, int T, float* wav_out, void* stream) {
  return ctx ? ctx->eng->dac_decode_zq(z, T, wav_out, (hipStream_t)stream) : ECHO_ERR;
}
int echo_set_pca(echo_ctx* ctx, const float* w, const float* mean, int on_device, void* stream) {
  return ctx ? ctx->eng->set_pca(w, mean, on_device, (hipStream_t)stream) : ECHO_ERR;
}
int echo_finalize_dac_encoder(echo_ctx* ctx, void* stream) { return ctx ? ctx->eng->finalize_dac_encoder((hipStream_t)stream) : ECHO_ERR; }
int echo_dac_encode(echo_ctx* ctx, const float* audio, long n_samples, float* latent_out, int32_t* codes_out, float* zq_out, void* stream) {
  return ctx ? ctx->eng->dac_encode(audio, n_samples, latent_out, codes_out, zq_out, (hipStream_t)stream) : ECHO_ERR;
}
int echo_set_pca_encode(echo_ctx* ctx, const float* w, const float* bias, float scale, int on_device, void* stream) {
  return ctx ? ctx->eng->set_pca_encode(w, bias, scale, on_device, (hipStream_t)stream) : ECHO_ERR;
}
int echo_dac_hop(echo_ctx* ctx) {
  if (!ctx) return -1;
  long hop = 1;
  const echo_config& c = ctx->eng->cfg;
  for (int i = 0; i < c.dac_n_rates; ++i) hop *= c.dac_rates[i];
  for (int i = 0; i < c.dac_n_up; ++i) hop *= c.dac_up_factors[i];
  return (int)hop;
}
int echo_debug_corrupt_tile(echo_ctx* ctx, int nth) {
  if (!ctx || nth < 0) return ECHO_ERR;
  ctx->eng->corrupt_countdown = nth;
  return ECHO_OK;
}
int echo_debug_get_kv(echo_ctx* ctx, int which, int layer, float* k_out, float* v_out, int* B_out, int* T_out) {
  return ctx ? ctx->eng->debug_get_kv(which, layer, k_out, v_out, B_out, T_out) : ECHO_ERR;
}
// ---- per-voice cache, streaming decode, workspace sizing
struct echo_voice { VoiceSnap* v; };
int echo_voice_capture(echo_ctx* ctx, echo_voice** out, void* stream) {
  if (!ctx || !out) return ECHO_ERR;
  VoiceSnap* v = nullptr;
  const int r = ctx->eng->voice_capture(&v, (hipStream_t)stream);
  if (r != ECHO_OK) return r;
  *out = new echo_voice{v};
  return ECHO_OK;
}
int echo_voice_bind(echo_ctx* ctx, const echo_voice* voice, void* stream) {
  return ctx && voice ? ctx->eng->voice_bind(voice->v, (hipStream_t)stream) : ECHO_ERR;
}
int64_t echo_voice_bytes(const echo_voice* voice) {
  return voice && voice->v ? (int64_t)(voice->v->kv_bytes + voice->v->vt_bytes + voice->v->bias_bytes) : 0;
}
void echo_voice_destroy(echo_voice* voice) {
  if (!voice) return;
  if (voice->v) { (void)hipSetDevice(voice->v->device); delete voice->v; }
  delete voice;
}
int echo_dac_decode_tail(echo_ctx* ctx, const float* latent, int T, int first_frame, float latent_scale, float* wav_out, void* stream) {
  return ctx ? ctx->eng->dac_decode_tail(latent, T, first_frame, latent_scale, wav_out, (hipStream_t)stream) : ECHO_ERR;
}
int64_t echo_workspace_bytes(echo_ctx* ctx) { return ctx ? (int64_t)ctx->eng->workspace_bytes() : 0; }
int echo_reserve_workspace(echo_ctx* ctx, int B, int S, int Tt, int Ts, int T_dac) {
  return ctx ? ctx->eng->reserve_workspace(B, S, Tt, Ts, T_dac) : ECHO_ERR;
}

int echo_fp8_calibrate(echo_ctx* ctx, int on) { return ctx ? ctx->eng->fp8_calibrate(on) : ECHO_ERR; }
int echo_fp8_calibration(echo_ctx* ctx, float* out, int n) { return ctx ? ctx->eng->fp8_calibration(out, n) : ECHO_ERR; }
int echo_fp8_set_static_scales(echo_ctx* ctx, const float* scales, int n) { return ctx ? ctx->eng->fp8_set_static(scales, n) : ECHO_ERR; }

int echo_set_profiling(echo_ctx* ctx, int on) { if (!ctx) return ECHO_ERR; ctx->eng->profiling = on != 0; return ECHO_OK; }
int echo_get_profile(echo_ctx* ctx, echo_profile* out) { if (!ctx || !out) return ECHO_ERR; *out = ctx->eng->prof; return ECHO_OK; }

// ---- single-kernel entry points
static thread_local std::string g_op_error;
static int op_status(hipError_t e) {
  if (e == hipSuccess) return ECHO_OK;
  g_create_error = std::string("kernel launch failed: ") + hipGetErrorString(e);
  return ECHO_ERR;
}

int echo_op_quant_rows_fp8(const void* x, int64_t ldx, void* q, int64_t ldq, float* scale, int rows, int K, void* stream) {
  return op_status(launch_quant_rows_fp8(x, ldx, q, ldq, scale, rows, K, (hipStream_t)stream));
}

int echo_op_norm_adaln_fp8(const void* x, int64_t ldx, void* q, int64_t ldq, float* scale, int rows, int D, float eps,
                           const void* scale1p, const void* shift, void* stream) {
  if (D < 8 || (D & 7) || D > 4096 || rows < 1) return ECHO_ERR;
  return op_status(launch_norm_adaln_fp8(x, ldx, q, ldq, scale, rows, D, eps, scale1p, shift, (hipStream_t)stream));
}

int echo_op_gemm(int dtype, const echo_gemm_desc* d, void* stream) {
  GemmArgs g;
  gemm_args_init(&g);
  g.A = d->A; g.W = d->W; g.C = d->C; g.C2 = d->C2; g.M = d->M; g.N = d->N; g.K = d->K; g.Npad = d->Npad;
  g.lda = d->lda; g.ldw = d->ldw; g.ldc = d->ldc; g.taps = d->taps < 1 ? 1 : d->taps; g.tap_base = d->tap_base; g.tap_shift = d->tap_shift;
  g.nbatch = d->nbatch < 1 ? 1 : d->nbatch; g.nbi = d->nbi < 1 ? 1 : d->nbi;
  g.a_bo = d->a_bo; g.a_bi = d->a_bi; g.w_bo = d->w_bo; g.w_bi = d->w_bi; g.c_bo = d->c_bo; g.c_bi = d->c_bi;
  g.acc_scale = d->acc_scale == 0.0f ? 1.0f : d->acc_scale;
  g.bias = d->bias; g.bias_bo = d->bias_bo; g.bias_bi = d->bias_bi; g.vec_mod = d->vec_mod; g.div = d->div; g.act = d->act;
  g.colscale = d->colscale; g.res = d->res; g.ldres = d->ldres; g.res_bo = d->res_bo; g.res_bi = d->res_bi;
  g.snake_alpha = d->snake_alpha; g.store_main = d->store_main; g.swiglu = d->swiglu;
  g.cfg = d->cfg; g.ksplit = d->ksplit; g.ws = d->ws; g.ws_bytes = d->ws_bytes; g.split3 = d->split3; g.w_presplit = d->split3 ? d->w_presplit : 0;
  g.fp8 = d->fp8; g.a_scale = d->a_scale; g.w_scale = d->w_scale;
  g.a_scale_const = d->a_scale_const; g.c8 = d->c8; g.c8_ld = d->c8_ld; g.c8_inv = d->c8_inv; g.qkv_gate_act = d->qkv_gate_act;
  g.qkv_mode = d->qkv_mode; g.qkv_D = d->qkv_D; g.qkv_S = d->qkv_S; g.rope_heads = d->rope_heads; g.pos0 = d->pos0; g.qk_eps = d->qk_eps;
  g.qk_w = d->qk_w; g.rope = d->rope; g.vt = d->vt; g.vt_ld = d->vt_ld; g.vt_row_stride = d->vt_row_stride;
  return op_status(dtype == ECHO_BF16 ? launch_gemm_nt<bf16_t>(g, (hipStream_t)stream) : launch_gemm_nt<float>(g, (hipStream_t)stream));
}
int echo_op_presplit_weights(float* w, int64_t rows, int64_t ld, void* stream) {
  return op_status(launch_presplit_w(w, rows, ld, (hipStream_t)stream));
}
int echo_op_pack_rows(const void* src, int sdt, int64_t sld, void* dst, int ddt, int64_t dld, int rows, int cols, int dst_row0,
                      int swiglu_half, void* stream) {
  return op_status(launch_pack_rows(src, sdt, sld, dst, ddt, dld, rows, cols, dst_row0, swiglu_half, (hipStream_t)stream));
}
int echo_op_attention_bf16(const echo_attn_desc* d, void* stream) {
  AttnArgs a;
  memset(&a, 0, sizeof(a));
  a.Q = (const bf16_t*)d->Q; a.q_ld = d->q_ld; a.q_row_stride = d->q_row_stride;
  a.O = (bf16_t*)d->O; a.o_ld = d->o_ld; a.o_row_stride = d->o_row_stride;
  a.G = (const bf16_t*)d->G; a.g_ld = d->g_ld; a.g_row_stride = d->g_row_stride;
  a.S = d->S; a.H = d->H; a.rows = d->rows; a.nseg = d->nseg; a.causal = d->causal; a.scale = d->scale; a.prof = d->prof;
  a.O8 = (uint8_t*)d->O8; a.o8_ld = d->o8_ld; a.o8_row_stride = d->o8_row_stride; a.o8_inv = d->o8_inv; a.g_act = d->g_activated;
  for (int s = 0; s < d->nseg && s < 4; ++s) {
    a.seg[s].K = (const bf16_t*)d->seg[s].K; a.seg[s].k_ld = d->seg[s].k_ld; a.seg[s].k_row_stride = d->seg[s].k_row_stride;
    a.seg[s].k_head_stride = d->seg[s].k_head_stride;
    a.seg[s].Vt = (const bf16_t*)d->seg[s].Vt; a.seg[s].vt_ld = d->seg[s].vt_ld; a.seg[s].vt_row_stride = d->seg[s].vt_row_stride;
    a.seg[s].vt_head_stride = d->seg[s].vt_head_stride;
    a.seg[s].nkeys = d->seg[s].nkeys; a.seg[s].bias = d->seg[s].bias; a.seg[s].bias_row_stride = d->seg[s].bias_row_stride;
    a.seg[s].kv_mod = d->seg[s].kv_mod;
  }
  // range-report words of the fast joint-attention kernel (common.h AttnArgs::redo): the caller may pass its own buffer
  // (needed when several streams run attention at once); otherwise one buffer per host thread and device
  a.redo = (int*)d->redo;
  if (!a.redo && a.rows > 0 && a.H > 0 && a.S > 0) {
    static thread_local DevBuf redo_buf[64];
    int dev = 0; (void)hipGetDevice(&dev);
    DevBuf& rb = redo_buf[dev & 63];
    if (hipError_t e = rb.reserve((size_t)attn_redo_words(a.rows, a.H, a.S) * sizeof(int)); e != hipSuccess) return op_status(e);
    a.redo = rb.as<int>();
  }
  return op_status(launch_attention_bf16(a, (hipStream_t)stream));
}
int echo_op_norm(int dtype, int mode, const void* x, int64_t ldx, void* y, int64_t ldy, int rows, int D, float eps, const void* w0,
                 const void* w1, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  return op_status(dtype == ECHO_BF16
                       ? launch_norm<bf16_t>(mode, (const bf16_t*)x, ldx, (bf16_t*)y, ldy, rows, D, eps, (const bf16_t*)w0, (const bf16_t*)w1, st)
                       : launch_norm<float>(mode, (const float*)x, ldx, (float*)y, ldy, rows, D, eps, (const float*)w0, (const float*)w1, st));
}
int echo_op_headnorm_rope(int dtype, void* x, int64_t ldx, int64_t t_stride, int nt, int rows, int S, int H, const void* w,
                          int64_t w_stride, float eps, int do_norm, int rope_heads, const void* rope, int pos0, int pos_mul, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  return op_status(dtype == ECHO_BF16
                       ? launch_headnorm_rope_nt<bf16_t>((bf16_t*)x, ldx, t_stride, nt, rows, S, H, (const bf16_t*)w, w_stride, eps, do_norm,
                                                         rope_heads, (const float2*)rope, pos0, pos_mul, st)
                       : launch_headnorm_rope_nt<float>((float*)x, ldx, t_stride, nt, rows, S, H, (const float*)w, w_stride, eps, do_norm,
                                                        rope_heads, (const float2*)rope, pos0, pos_mul, st));
}
int echo_op_transpose_heads(int dtype, const void* v, int64_t ldv, void* vt, int64_t vt_ld, int64_t vt_b_stride, int B, int S, int H,
                            int HD, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  return op_status(dtype == ECHO_BF16
                       ? launch_transpose_heads<bf16_t>((const bf16_t*)v, ldv, (bf16_t*)vt, vt_ld, vt_b_stride, B, S, H, HD, st)
                       : launch_transpose_heads<float>((const float*)v, ldv, (float*)vt, vt_ld, vt_b_stride, B, S, H, HD, st));
}

// ---- on-device post-processing (postproc.hip)
int echo_op_find_flattening_point(const float* latent, int64_t item_stride, int B, int T, int W, int window, float target, float std_threshold,
                                  int32_t* out_dev, void* stream) {
  return op_status(launch_flatten_point(latent, item_stride, B, T, W, window, target, std_threshold, out_dev, (hipStream_t)stream));
}
int echo_op_trailing_quiet(const float* const* chunks_host, const int64_t* lens_host, int n, int max_window, float threshold, int32_t* out_dev,
                           void* stream) {
  if (!chunks_host || !lens_host || n < 1 || n > ECHO_MAX_CHUNKS) return op_status(hipErrorInvalidValue);
  long lens[ECHO_MAX_CHUNKS];
  for (int i = 0; i < n; ++i) lens[i] = (long)lens_host[i];
  return op_status(launch_trailing_quiet(chunks_host, lens, n, max_window, threshold, out_dev, (hipStream_t)stream));
}
int echo_op_assemble_chunks(const float* const* src_host, const int64_t* start_host, const int64_t* len_host, const int64_t* valid_host,
                            const int32_t* overlap_host, int n, float* out_dev, int64_t total, void* stream) {
  if (!src_host || !start_host || !len_host || !valid_host || !overlap_host || n < 1 || n > ECHO_MAX_CHUNKS) return op_status(hipErrorInvalidValue);
  long st_[ECHO_MAX_CHUNKS], ln_[ECHO_MAX_CHUNKS], va_[ECHO_MAX_CHUNKS]; int ov_[ECHO_MAX_CHUNKS];
  for (int i = 0; i < n; ++i) { st_[i] = (long)start_host[i]; ln_[i] = (long)len_host[i]; va_[i] = (long)valid_host[i]; ov_[i] = overlap_host[i]; }
  return op_status(launch_assemble_chunks(src_host, st_, ln_, va_, ov_, n, out_dev, (long)total, (hipStream_t)stream));
}

int echo_op_resample(const float* x_dev, int64_t n, const float* bank_dev, int taps, int up, int down, int width, float* out_dev, int64_t n_out,
                     void* stream) {
  return op_status(launch_resample(x_dev, (long)n, bank_dev, taps, up, down, width, out_dev, (long)n_out, (hipStream_t)stream));
}

}  // extern "C"
