/* libechohip — C ABI of the MI355X-native Echo-TTS hot path.
 *
 * The reference (sruckh/echo-tts) is pure Python/PyTorch and has no FFI of its own; the boundary it
 * offers is its Python call signatures.  Each entry point below replaces the body of
 * one of those Python functions; the Python host in `echo-tts_amd/` keeps the reference signatures
 * and calls these through ctypes (INTEGRATION.md shows the binding).
 *
 * Conventions
 *   - every function returns 0 on success, non-zero on failure; echo_last_error() gives the text;
 *   - all tensor arguments are raw DEVICE pointers owned by the caller unless a parameter says
 *     "host"; the library never frees caller memory; it owns packed weights, KV caches and
 *     workspaces inside echo_ctx (grown on demand, never inside the sampler's step loop once warm);
 *   - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream); all work
 *     is enqueued on it and nothing synchronises unless stated;
 *   - dtype codes: ECHO_F32 = 0, ECHO_BF16 = 1.  "T" below is the context's activation type:
 *     bf16 (production) or f32 (parity mode: fp32 weights, activations and MFMA);
 *   - one echo_ctx per device, not thread-safe (the reference is single-threaded, one job at a time).
 */
#ifndef ECHO_HIP_H
#define ECHO_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define ECHO_F32 0
#define ECHO_BF16 1
#define ECHO_ABI_VERSION 7

typedef struct echo_ctx echo_ctx;

/* Sizes of EchoDiT (reference: inference.py:16-24, model.py:472-559) and of the Fish S1-DAC decode
 * path (autoencoder.py:1144-1192).  head_dim must be 128 for the DiT and its encoders. */
typedef struct {
  int precision;                 /* ECHO_BF16 or ECHO_F32 */
  int latent_size, model_size, num_layers, num_heads, intermediate_size;
  float norm_eps;
  int text_vocab_size, text_model_size, text_num_layers, text_num_heads, text_intermediate_size;
  int speaker_patch_size, speaker_model_size, speaker_num_layers, speaker_num_heads, speaker_intermediate_size;
  int timestep_embed_size, adaln_rank;
  int has_latent_encoder;        /* 0 when the checkpoint was loaded with delete_blockwise_modules (inference.py:28-34) */
  /* DAC decode path (fp32) */
  int dac_latent_dim, dac_decoder_dim, dac_n_rates, dac_rates[8];
  int dac_post_layers, dac_post_heads, dac_post_head_dim, dac_post_ffn, dac_post_window;
  int dac_n_up, dac_up_factors[4];
  float dac_norm_eps;
  /* DAC encode path (speaker reference -> latents; autoencoder.py:903-929, 376-464, 117-158): Encoder channel base,
   * strides, transformer layers per block (window dac_enc_window, heads = C / 64, ffn = 3 C), VQ stack sizes.
   * dac_enc_dim == 0: no encode path in this context. */
  int dac_enc_dim, dac_enc_n_rates, dac_enc_rates[8], dac_enc_tlayers[8], dac_enc_window;
  int dac_n_codebooks, dac_codebook_size, dac_codebook_dim, dac_semantic_size;
  /* BASELINE config C5 ("fp8 MFMA weight path for DiT GEMMs"): 1 = the four large linears of every EchoDiT block (QKV+gate,
   * wo, w1|w3, w2) run on OCP e4m3 operands — weights quantised once at echo_finalize_dit with one scale per output row,
   * activations per token row in front of each GEMM — on the block-scaled MFMA at twice the bf16 rate; accumulation, tails,
   * attention, norms and the encoders stay as in the bf16 engine.  precision must be ECHO_BF16. */
  int dit_fp8;
} echo_config;

int echo_abi_version(void);
const char* echo_last_error(echo_ctx* ctx);            /* ctx may be NULL: last error of a failed create */

/* handler.py:323-417 `_load_models` / inference.py:14-47,56-76: context + weights */
int echo_ctx_create(const echo_config* cfg, int device, echo_ctx** out);
void echo_ctx_destroy(echo_ctx* ctx);

/* Register one checkpoint tensor under its reference state-dict name (SURVEY.md §A.5 for EchoDiT;
 * INTEGRATION.md lists the DAC names, which are the reference names with weight-norm folded and the
 * conv kernels reshaped to GEMM form by the Python loader).  `data` may be a host or a device pointer. */
int echo_load_tensor(echo_ctx* ctx, const char* name, const void* data, int dtype, int ndim, const int64_t* shape,
                     int data_on_device);
int echo_finalize_dit(echo_ctx* ctx, void* stream);    /* pack EchoDiT weights for the kernels, drop the raw copies */
int echo_finalize_dac(echo_ctx* ctx, void* stream);
/* encode-path tensors ("enc.*", prepared by the Python loader: weight-norm folded, conv kernels in GEMM form, codebooks
 * normalised, from_codes operands concatenated; INTEGRATION.md lists them).  Needs echo_finalize_dac first. */
int echo_finalize_dac_encoder(echo_ctx* ctx, void* stream);

/* Position tables computed by the host with the reference's own expressions (exactness for free):
 * rope: (npos, 64) float2 = (cos, sin) of model.py:9-14 for head_dim 128;
 * ae_rope: (npos, head_dim/2, 2) fp32 copy of the bf16 cache of autoencoder.py:805-813. */
int echo_set_rope_table(echo_ctx* ctx, const void* table_dev, int npos);
int echo_set_ae_rope_table(echo_ctx* ctx, const void* table_dev, int npos);

/* model.py:606-613 EchoDiT.get_kv_cache_text.  ids (B,Tt) int32; key_bias: NULL when every mask is a
 * prefix, else (B,Tt) f32 additive 0/-inf; nkeys_host[b] = 1 + index of the last attended token. */
int echo_encode_text(echo_ctx* ctx, const int32_t* ids, const float* key_bias, const int32_t* nkeys_host, int B, int Tt,
                     void* stream);
/* model.py:615-621 get_kv_cache_speaker.  latent (B,Ts,latent) of T; masks are over the Ts/patch keys. */
int echo_encode_speaker(echo_ctx* ctx, const void* latent, const float* key_bias, const int32_t* nkeys_host, int B, int Ts,
                        void* stream);
/* model.py:623-636 get_kv_cache_latent for the first n_latents (multiple of patch) prefix latents. */
int echo_encode_latent_prefix(echo_ctx* ctx, const void* latent, int B, int n_latents, long row_stride_elems, void* stream);
/* inference.py:408-414 _multiply_kv_cache on the speaker cache (K and V of layers < max_layers) */
int echo_scale_speaker_kv(echo_ctx* ctx, float scale, int max_layers, void* stream);

/* Per-voice cache.  The reference re-encodes the reference voice for every text chunk (handler.py:750-758 ->
 * inference.py:333-340 -> model.py:615-621).  echo_voice_capture copies the context's current speaker cache (K / V / V^T of all
 * layers, key counts, optional key bias) into an object owned by the caller; echo_voice_bind makes it the context's speaker
 * cache again (device copy, bit-identical to a fresh echo_encode_speaker) on any context of the same model, precision and
 * device.  The speaker cache may have batch size 1 while the text cache has B: every row then shares the one voice. */
typedef struct echo_voice echo_voice;
int echo_voice_capture(echo_ctx* ctx, echo_voice** out, void* stream);
int echo_voice_bind(echo_ctx* ctx, const echo_voice* voice, void* stream);
int64_t echo_voice_bytes(const echo_voice* voice);
void echo_voice_destroy(echo_voice* voice);

/* SURVEY.md §8b "echo_workspace_bytes": device bytes of KV caches + workspaces the context currently holds, and a call that
 * grows them for a request geometry up front (B utterances per sampler call, S latents, Tt text tokens, Ts speaker latents,
 * T_dac frames per decode) so that the first request allocates nothing. */
int64_t echo_workspace_bytes(echo_ctx* ctx);
/* fp8 engine (echo_config.dit_fp8, BASELINE config C5): calibration of static activation scales for the two block operands no producer
 * can quantise per row (attention output -> wo, SwiGLU output -> w2); SURVEY.md 8f-4 "fp8 calibration".  echo_fp8_calibrate(ctx, 1)
 * clears the per-block maxima and records, during every forward until echo_fp8_calibrate(ctx, 0), the largest dynamic row scale
 * (amax / 448) each block sees; echo_fp8_calibration copies the 2 * num_layers values out ([2 l] attention output, [2 l + 1] SwiGLU
 * output); echo_fp8_set_static_scales installs 2 * num_layers scales (the caller applies its margin; n = 0: back to dynamic row scales):
 * the attention epilogue and the SwiGLU tail then write the e4m3 operands themselves and both quantisation passes of a block go away.
 * Values beyond 448 * scale saturate.  The reference has no fp8 path: this is this engine's own stated arithmetic (oracle:
 * set_fp8_block_linears(True, act_static=...)). */
int echo_fp8_calibrate(echo_ctx* ctx, int on);
int echo_fp8_calibration(echo_ctx* ctx, float* out, int n);
int echo_fp8_set_static_scales(echo_ctx* ctx, const float* scales, int n);
int echo_reserve_workspace(echo_ctx* ctx, int B, int S, int Tt, int Ts, int T_dac);

/* model.py:563-604 EchoDiT.forward for `rows` = R*B rows that share one timestep.
 * x (rows*S, latent) of T; temb (1, timestep_embed) of T (model.py:27-43 evaluated by the host);
 * row r uses text/speaker KV of batch item r % B; row_text_on/row_spk_on (host, rows) switch the
 * segments per row (the CFG "uncond" rows); v_out (rows*S, latent) fp32. */
int echo_dit_forward(echo_ctx* ctx, const void* x, const void* temb, int rows, int B, int S, int start_pos, int use_latent,
                     const int32_t* row_text_on, const int32_t* row_spk_on, float* v_out, void* stream);

/* ABI 7: the same forward with PER-ROW timesteps (model.py:563-604 takes any t (R,); model.py:27-43): temb (n_t, timestep_embed) of T
 * holds the embeddings of the n_t distinct timesteps, row_t (host, rows) names the one each row uses.  Rows are processed together;
 * only the launches that read the AdaLN modulation run once per run of consecutive rows with equal row_t. */
int echo_dit_forward_t(echo_ctx* ctx, const void* x, const void* temb, int n_t, const int32_t* row_t, int rows, int B, int S, int start_pos,
                       int use_latent, const int32_t* row_text_on, const int32_t* row_spk_on, float* v_out, void* stream);

typedef struct {
  int has_cfg;                   /* inference.py:484 */
  float dt;                      /* fp32 (t_next - t), inference.py:515 */
  int rescale;                   /* inference.py:507-508, coefficients of inference.py:421-423 evaluated by the host */
  float r_inv1mt, r_ratio, r_1mt;
  int kv_unscale_after;          /* inference.py:511-513 */
} echo_step;

typedef struct {
  int B, S, num_steps;
  int start_pos, use_latent;     /* blockwise: inference_blockwise.py:91-94 */
  float cfg_scale_text, cfg_scale_speaker;
  int has_truncation;            /* ABI 7: 0 = truncation_factor is None (init_scale ignored: a zero-initialised struct samples from the
                                  * un-truncated noise); 1 = x_t = x_t * init_scale (inference.py:478-479), a literal 0.0 included */
  float init_scale;              /* truncation_factor; read only when has_truncation != 0 */
  float kv_scale; int kv_max_layers;   /* used by kv_unscale_after steps: multiply by 1/kv_scale */
  const echo_step* steps;        /* host, num_steps entries */
  const void* temb;              /* device, (num_steps, timestep_embed) of T */
} echo_sampler_params;

/* inference.py:427-517 sample_euler_cfg_independent_guidances, from the noise draw on.
 * x_init (B,S,latent) fp32 noise (the host draws it exactly like inference.py:457,477);
 * latent_out (B,S,latent) fp32.  Text/speaker(/latent) KV must have been encoded on this ctx. */
int echo_sample_euler(echo_ctx* ctx, const echo_sampler_params* p, const float* x_init, float* latent_out, void* stream);

/* inference.py:226-229 ae_decode -> autoencoder.py:1128-1132 DAC.decode_zq, one batch item:
 * latent (T, latent_size) fp32 -> wav (T * hop) fp32. */
int echo_dac_decode(echo_ctx* ctx, const float* latent, int T, float latent_scale, float* wav_out, void* stream);
/* autoencoder.py:1128-1132 DAC.decode_zq alone: z (T, dac_latent_dim) fp32 CHANNELS-LAST (= z_q[b].T) -> wav (T * hop). */
int echo_dac_decode_zq(echo_ctx* ctx, const float* z, int T, float* wav_out, void* stream);
/* ABI 6: B utterances of T frames each (latent: B x T x latent_size, contiguous) -> wav_out + b * wav_stride (floats).  Same result as B
 * echo_dac_decode calls up to the fp32 summation order of the row-wise GEMMs: the PCA inverse, the post_module transformer and its norm
 * run once on the B * T stacked rows (attention per item and head), the convolution stack per item.  inference.py:226-229 on a batch. */
int echo_dac_decode_batch(echo_ctx* ctx, const float* latent, int B, int T, float latent_scale, float* wav_out, int64_t wav_stride, void* stream);
/* Streaming decode (reference: every conv of autoencoder.py:264-331 is causal, gradio_app.py:43 decodes whole
 * utterances): latent (T, latent_size) fp32 are ALL frames generated so far; the window-limited post_module transformer
 * runs over all of them, the convolutional stack only over frames first_frame.. ; wav_out receives (T - first_frame) * 2048
 * samples.  Samples further than the stack's receptive field (< 10 frames) behind first_frame equal the
 * whole-utterance decode; the caller (DACStream in autoencoder.py) discards that context. */
int echo_dac_decode_tail(echo_ctx* ctx, const float* latent, int T, int first_frame, float latent_scale, float* wav_out, void* stream);
/* inference.py:86-99 PCAState: w = pca_components transposed, (dac_latent_dim, latent_size) row-major fp32; mean (dac_latent_dim). */
int echo_set_pca(echo_ctx* ctx, const float* w, const float* mean, int on_device, void* stream);
int echo_dac_hop(echo_ctx* ctx);
/* inference.py:218-224 ae_encode = DAC.encode_zq (autoencoder.py:1080-1126) + PCA projection, one item:
 * audio (n_samples) fp32 device pointer, n_samples a multiple of the frame length (hop * 4 = 2048; the host pads);
 * latent_out (T, latent_size) fp32, T = n_samples / frame; optional codes_out (1 + n_codebooks, T) int32 and
 * zq_out (T, latent_dim) fp32 (channels-last) for parity checks.  PCA operands: echo_set_pca_encode. */
int echo_dac_encode(echo_ctx* ctx, const float* audio, long n_samples, float* latent_out, int32_t* codes_out, float* zq_out,
                    void* stream);
/* w (latent_size, latent_dim) = pca_components, bias (latent_size) = -(pca_mean @ pca_components^T), scale = latent_scale */
int echo_set_pca_encode(echo_ctx* ctx, const float* w, const float* bias, float scale, int on_device, void* stream);

/* ---- single-kernel entry points (unit tests and micro-benchmarks; same kernels the engine launches) ---- */
typedef struct {
  const void* A; const void* W; void* C; void* C2;
  int M, N, K, Npad; int64_t lda, ldw, ldc;
  int taps, tap_base, tap_shift;
  int nbatch, nbi; int64_t a_bo, a_bi, w_bo, w_bi, c_bo, c_bi;
  float acc_scale;
  const void* bias; int64_t bias_bo, bias_bi; int vec_mod;
  float div; int act;
  const void* colscale;
  const void* res; int64_t ldres, res_bo, res_bi;
  const void* snake_alpha;
  int store_main, swiglu;
  int cfg;                       /* plan: 0..4, 6..9 tile configurations (csrc/gemm.hip TileCfg table), 5 = bf16 ping-pong kernel */
  int ksplit; void* ws; int64_t ws_bytes;   /* split-K: fp32 workspace of ksplit * roundup(M,768) * Npad * 4 bytes */
  int split3;                    /* fp32 only: 3 x bf16 MFMA per product (hi/lo operand splitting), ~1e-5 relative error */
  int fp8;                       /* dtype ECHO_BF16, cfg 5 only: A and W are e4m3 bytes, y = acc * a_scale[m] * w_scale[n]; K % 128 == 0 */
  const float* a_scale; const float* w_scale;
  /* fused QKV(G) tail (model.py:217-232): the N axis is [q | k | v | gate] x qkv_D; q and k get the per-head RMSNorm (qk_w =
   * [q_norm | k_norm], each qkv_D, eps qk_eps) and interleaved-pair RoPE on heads < rope_heads at position pos0 + (m % qkv_S)
   * (rope: (npos, 64) float2 cos/sin); v is written transposed to vt[(m / qkv_S)][h * 128 + d][m % qkv_S] (pitches vt_ld,
   * vt_row_stride in elements); gate is stored as is.  cfg 0-5 only, qkv_D % 256 == 0. */
  int qkv_mode, qkv_D, qkv_S, rope_heads, pos0; float qk_eps;
  const void* qk_w; const void* rope; void* vt; int64_t vt_ld, vt_row_stride;
  int w_presplit;                /* split3 only: W was reformatted in place by echo_op_presplit_weights (static weights) */
  /* ABI 6, fp8 with a static (calibrated) activation scale: a_scale == NULL -> every A row has the scale a_scale_const (> 0);
   * c8 != NULL (fp8 + swiglu only): the SwiGLU output leaves as e4m3(bf16(out) * c8_inv) bytes at c8 (row pitch c8_ld bytes, % 8 == 0,
   * saturating at +-448) instead of bf16 at C - the A operand of the next fp8 GEMM without a quantisation pass */
  float a_scale_const; void* c8; int64_t c8_ld; float c8_inv;
  int qkv_gate_act;              /* ABI 6, qkv_mode + bf16: the gate section is stored as bf16(sigmoid(bf16(acc))) (v_exp + v_rcp form) - the value the
                                  * attention epilogue derives from a raw gate; pass echo_attn_desc.g_activated = 1 to the attention that reads it */
} echo_gemm_desc;
int echo_op_gemm(int dtype, const echo_gemm_desc* d, void* stream);
/* In-place reformat of an fp32 weight matrix (rows x ld, ld % 32 == 0) for w_presplit: every aligned block of 32 floats becomes
 * 32 bf16 hi = bf16(x) followed by 32 bf16 lo = bf16(x - hi), the values the split3 kernels otherwise compute per fragment.  Results of a
 * split3 GEMM are bit-identical with and without it.  (New in ABI 5; the engine applies it to the Fish S1-DAC decoder's conv weights.) */
int echo_op_presplit_weights(float* w, int64_t rows, int64_t ld, void* stream);
/* bf16 rows -> OCP e4m3 bytes + one fp32 scale per row (amax / 448): the operand format of the fp8 GEMM */
int echo_op_quant_rows_fp8(const void* x, int64_t ldx, void* q, int64_t ldq, float* scale, int rows, int K, void* stream);
/* ABI 7: the fp8 engine's AdaLN-apply (model.py:76-83 on bf16 rows: x_hat * scale1p + shift, rounded to bf16) whose output leaves as
 * e4m3 bytes + one fp32 scale per row (amax / 448) - echo_op_norm(mode 0) followed by echo_op_quant_rows_fp8 in one pass.
 * D % 8 == 0, D <= 4096. */
int echo_op_norm_adaln_fp8(const void* x, int64_t ldx, void* q, int64_t ldq, float* scale, int rows, int D, float eps,
                           const void* scale1p, const void* shift, void* stream);
int echo_op_pack_rows(const void* src, int src_dtype, int64_t src_ld, void* dst, int dst_dtype, int64_t dst_ld, int rows,
                      int cols, int dst_row0, int swiglu_half, void* stream);

typedef struct {
  const void* K; int64_t k_ld, k_row_stride, k_head_stride;
  const void* Vt; int64_t vt_ld, vt_row_stride, vt_head_stride;
  const int32_t* nkeys; const float* bias; int64_t bias_row_stride; int kv_mod;
} echo_attn_seg;
typedef struct {
  const void* Q; int64_t q_ld, q_row_stride;
  void* O; int64_t o_ld, o_row_stride;
  const void* G; int64_t g_ld, g_row_stride;
  int S, H, rows, nseg; echo_attn_seg seg[4]; int causal; float scale;
  void* prof;                    /* diagnostic build only: (workgroups*4, 4) uint64 cycle sums; NULL normally */
  void* redo;                    /* optional device scratch of rows * H * ceil(S / 256) int32 for the fast kernel's range report (DESIGN.md
                                    3.2); NULL: the library uses a buffer of its own per host thread and device */
  /* ABI 6: O8 != NULL -> the (gated) output is written as e4m3(bf16(out) * o8_inv) bytes to O8 (same indexing as O, pitches in bytes,
   * o8_ld % 4 == 0, saturating) INSTEAD of bf16 to O: wo's fp8 operand under a static activation scale (echo_fp8_set_static_scales) */
  void* O8; int64_t o8_ld, o8_row_stride; float o8_inv;
  int g_activated;               /* ABI 6: G already holds bf16(sigmoid(gate)) (echo_gemm_desc.qkv_gate_act): the epilogue only multiplies */
} echo_attn_desc;
int echo_op_attention_bf16(const echo_attn_desc* d, void* stream);

int echo_op_norm(int dtype, int mode, const void* x, int64_t ldx, void* y, int64_t ldy, int rows, int D, float eps,
                 const void* w0, const void* w1, void* stream);
int echo_op_headnorm_rope(int dtype, void* x, int64_t ldx, int64_t t_stride, int nt, int rows, int S, int H, const void* w,
                          int64_t w_stride, float eps, int do_norm, int rope_heads, const void* rope, int pos0, int pos_mul,
                          void* stream);
int echo_op_transpose_heads(int dtype, const void* v, int64_t ldv, void* vt, int64_t vt_ld, int64_t vt_b_stride, int B, int S,
                            int H, int HD, void* stream);

/* ---- on-device post-processing; `*_host` arguments are host arrays of n <= 64 entries ----
 * inference.py:288-296 find_flattening_point for B latents (T, W) fp32 (item_stride floats apart): out_dev[b] = first frame whose
 * zero-extended `window` frames have unbiased std < std_threshold and |mean - target| < 0.1, T if none. */
int echo_op_find_flattening_point(const float* latent, int64_t item_stride, int B, int T, int W, int window, float target,
                                  float std_threshold, int32_t* out_dev, void* stream);
/* handler.py:199-211: out_dev[c] = number of trailing samples below `threshold` among the last min(len, max_window) of chunk c */
int echo_op_trailing_quiet(const float* const* chunks_host, const int64_t* lens_host, int n, int max_window, float threshold,
                           int32_t* out_dev, void* stream);
/* handler.py:126-170 crossfade_chunks over chunks already trimmed / zero-extended by handler.py:213-232: chunk i supplies
 * len[i] samples from output position start[i] (the first valid[i] from src[i], zeros after), overlapping chunk i + 1 by
 * overlap[i] samples mixed as result * linspace(1,0,n) + next * linspace(0,1,n). */
int echo_op_assemble_chunks(const float* const* src_host, const int64_t* start_host, const int64_t* len_host, const int64_t* valid_host,
                            const int32_t* overlap_host, int n, float* out_dev, int64_t total, void* stream);

/* ABI 7, SURVEY.md 8f-4 `load_audio` (inference.py:104-113: torchaudio.functional.resample to 44.1 kHz): polyphase FIR resampling of one mono
 * signal on the device.  bank (up, taps) fp32 = the windowed-sinc filters of the `up` output phases (built by the host:
 * echo_tts_amd.inference.sinc_resample_bank restates torchaudio's published sinc_interp_hann kernel); out[f * up + p] =
 * sum_k bank[p][k] * x[f * down + k - width], x = 0 outside [0, n).  The caller crops to ceil(up * n / down) samples. */
int echo_op_resample(const float* x_dev, int64_t n, const float* bank_dev, int taps, int up, int down, int width, float* out_dev, int64_t n_out,
                     void* stream);

/* test hook: copy one DiT layer's cached K and V (which: 0 text, 1 speaker, 2 latent) as fp32 (B, T, model_size);
 * K is post-k_norm(/RoPE), V as projected.  Synchronous.  *B_out / *T_out receive the cache geometry. */
int echo_debug_get_kv(echo_ctx* ctx, int which, int layer, float* k_out, float* v_out, int* B_out, int* T_out);

/* ABI 7, test instrument ("does the parity test have teeth?"): the nth plain-store GEMM launch with M >= 512 and N >= 512 from now on
 * (in an EchoDiT forward: in_proj, then wo and w2 of every block, whatever tile kernel runs them) has its output tile (rows 256..511,
 * columns 256..511) negated right after the launch - what one wrong entry in a kernel's tile walk would produce.  nth = 0 disarms.
 * tests/test_gpu_engine.py shows that its full-depth checks reject such a forward. */
int echo_debug_corrupt_tile(echo_ctx* ctx, int nth);

/* timing of the engine's phases, filled by the last echo_sample_euler / echo_dac_decode when
 * echo_set_profiling(ctx, 1) was called (HIP events on the caller's stream; adds synchronisation) */
typedef struct {
  float ms_mod, ms_steps, ms_total;
  float ms_gemm_sum; int n_gemm;   /* sum / count of gemm_nt launch durations of the last profiled run */
  /* the dominant kernel alone: gemm_pp_kernel launches (plan cfg 5) of that run, their HIP-event time on the launch
     stream and their algorithmic FLOPs 2 * M * N * K * taps (bench.py "roofline") */
  float ms_pp_sum; int n_pp;
  double flops_pp;
  float ms_attn_sum; int n_attn;   /* attn_kernel launches of that run (HIP events on the launch stream) */
} echo_profile;
int echo_set_profiling(echo_ctx* ctx, int on);
int echo_get_profile(echo_ctx* ctx, echo_profile* out);

#ifdef __cplusplus
}
#endif
#endif
