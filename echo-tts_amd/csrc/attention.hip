// Segmented flash attention, bf16, head_dim 128, for gfx950.
//
// Replaces the three F.scaled_dot_product_attention call sites that run in bf16 on the hot path
// (reference: model.py:148 encoder self-attention, model.py:255 joint attention) together with
// the torch.cat of K/V segments (model.py:246-247), the boolean key mask (model.py:249-253), the
// 3x batch replication of the text/speaker KV (inference.py:471-472) and the sigmoid output gate
// (model.py:157, 264).
//
// Keys come in up to 4 segments given by pointer (self | latent | text | speaker).  A segment may
// be shared by all CFG rows (stride 0 / kv_mod), clipped per row (nkeys[row]; 0 disables the
// segment for that row, which is how the "uncond" rows drop text or speaker) and optionally
// masked per key by an additive 0/-inf bias.  Masked keys contribute exactly 0, like -inf
// masking in the reference.
//
// Work split: one workgroup = 4 waves = 128 query rows of one (row, head); each wave owns 32
// queries.  K tile (64 keys x 128) and Vᵀ tile (128 x 64 keys) are DMA'd into LDS
// (global_load_lds_dwordx4, source-side XOR swizzle), double buffered, one barrier per tile.
// Sᵀ = K·Qᵀ is computed with the key on the MFMA row; K rows are fed in the order
// pi(i) = i with bits 2,3 swapped, which makes each lane's 16 accumulator registers hold 2 runs
// of 8 consecutive keys = exactly the B-operand fragments of the following Oᵀ += Vᵀ·Pᵀ MFMAs, so
// P never leaves registers (cdna_hip_programming.md §3 "accumulator tile as the next operand").
#include "common.h"

namespace {

constexpr int QT = 128;      // queries per workgroup
constexpr int KT = 64;       // keys per tile
constexpr int HD = 128;
constexpr int K_TILE_BYTES = KT * HD * 2;    // 16 KiB, rows of 256 B
constexpr int V_TILE_BYTES = HD * KT * 2;    // 16 KiB, rows of 128 B
constexpr int STAGE = K_TILE_BYTES + V_TILE_BYTES;
constexpr int SMEM = 2 * STAGE;

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
__device__ __forceinline__ void glds16(const char* src, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)lds_wave_base, 16, 0, 0);
}

__device__ __forceinline__ bf16x8 pack8(const f32x16& s, int base) {
  bf16x8 r;
#pragma unroll
  for (int j = 0; j < 8; ++j) r[j] = (short)f2bf(s[base + j]);
  return r;
}

template <bool CAUSAL>
__global__ void __launch_bounds__(256, 2) attn_kernel(const AttnArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int row = blockIdx.z, head = blockIdx.y;
  const int qbase = blockIdx.x * QT;
  const int fr = lane & 31, fh = lane >> 5;
  int q = qbase + wid * 32 + fr;
  const bool q_ok = q < p.S;
  const int qc = q_ok ? q : p.S - 1;

  // ---- Q fragments (B operand of Sᵀ = K·Qᵀ): lane holds Q[q][16kk + 8fh + 0..7]
  bf16x8 qf[8];
  {
    const bf16_t* qp = p.Q + (long)row * p.q_row_stride + (long)qc * p.q_ld + head * HD + 8 * fh;
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) qf[kk] = *(const bf16x8*)(qp + 16 * kk);
  }

  // ---- per-segment tile counts (wave-uniform)
  const int q_hi = min(qbase + QT, p.S) - 1;  // last query of this workgroup
  auto seg_keys = [&](int s) -> int {
    if (s >= p.nseg) return 0;
    int nk = p.seg[s].nkeys[row];
    if (CAUSAL) nk = min(nk, q_hi + 1);
    return nk < 0 ? 0 : nk;
  };
  const int nk0 = seg_keys(0), nk1 = seg_keys(1), nk2 = seg_keys(2), nk3 = seg_keys(3);
  // explicit selects instead of runtime-indexed arrays (those would live in scratch)
  auto NK = [&](int s) -> int { return s == 0 ? nk0 : s == 1 ? nk1 : s == 2 ? nk2 : s == 3 ? nk3 : 0; };
  auto NT = [&](int s) -> int { return (NK(s) + KT - 1) / KT; };
  const int total_tiles = NT(0) + NT(1) + NT(2) + NT(3);

  // ---- staging roles: wave w issues K pieces 4w..4w+3 (4 keys x 256 B each) and V pieces (8 d-rows x 128 B)
  const int k_row_in_piece = lane >> 4, k_slot = lane & 15;
  const int v_row_in_piece = lane >> 3, v_slot = lane & 7;

  auto stage = [&](int buf, int seg, int tile) {
    const AttnSeg& sg = p.seg[seg];
    const int kvrow = sg.kv_mod ? row % sg.kv_mod : row;
    const int nk = NK(seg);
    const int k0 = tile * KT;
    char* kb = smem + buf * STAGE + wid * 4096;
    char* vb = smem + buf * STAGE + K_TILE_BYTES + wid * 4096;
    const char* kbase = (const char*)(sg.K + (long)kvrow * sg.k_row_stride + (long)head * sg.k_head_stride);
    const char* vbase = (const char*)(sg.Vt + (long)kvrow * sg.vt_row_stride + (long)head * sg.vt_head_stride);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = (wid * 4 + i) * 4 + k_row_in_piece;       // key row inside the tile
      int key = k0 + r;
      key = key < nk ? key : nk - 1;                           // stay inside the valid rows
      const int chunk = k_slot ^ (r & 15);
      glds16(kbase + ((long)key * sg.k_ld) * 2 + chunk * 16, kb + i * 1024);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int d = (wid * 4 + i) * 8 + v_row_in_piece;
      const int chunk = v_slot ^ ((d >> 1) & 7);
      glds16(vbase + ((long)d * sg.vt_ld + k0) * 2 + chunk * 16, vb + i * 1024);
    }
  };

  // running state; scores are kept in the log2 domain (scale * log2(e) folded in)
  const float c = p.scale * 1.4426950408889634f;
  float m_i = -1e30f, l_i = 0.0f;
  f32x16 o[4];
#pragma unroll
  for (int d = 0; d < 4; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[d][r] = 0.0f;

  // LDS fragment addresses
  const int pi_row = (fr & 0x13) | ((fr & 4) << 1) | ((fr & 8) >> 1);  // swap bits 2 and 3
  const int sw_v = (lane >> 1) & 7;

  int seg = 0, tile = 0;
  while (seg < 4 && NT(seg) == 0) ++seg;
  if (total_tiles > 0) stage(0, seg, 0);
  for (int it = 0; it < total_tiles; ++it) {
    __syncthreads();
    // next tile
    int nseg_ = seg, ntile = tile + 1;
    if (ntile >= NT(seg)) { ntile = 0; ++nseg_; while (nseg_ < 4 && NT(nseg_) == 0) ++nseg_; }
    if (it + 1 < total_tiles) stage((it + 1) & 1, nseg_, ntile);

    const char* sk = smem + (it & 1) * STAGE;
    const char* sv = sk + K_TILE_BYTES;
    const int nk = NK(seg);
    const int k0 = tile * KT;

    // ---- Sᵀ (64 keys x 32 queries per wave)
    f32x16 s[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int r = 0; r < 16; ++r) s[ks][r] = 0.0f;
      const int krow = ks * 32 + pi_row;
      const char* kr = sk + krow * 256;
      const int sw_k = krow & 15;
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) {
        const bf16x8 kf = *(const bf16x8*)(kr + (((2 * kk + fh) ^ sw_k) << 4));
        s[ks] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[kk], s[ks], 0, 0, 0);
      }
    }

    // ---- scale, mask, online softmax.  Register r of s[ks] is key k0 + 32ks + 16(r>>3) + 8fh + (r&7).
    const AttnSeg& sg = p.seg[seg];
    const bool need_mask = (k0 + KT > nk) || (sg.bias != nullptr) || (CAUSAL && (k0 + KT - 1 > qbase + wid * 32));
    float mx = -INFINITY;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int r = 0; r < 16; ++r) s[ks][r] *= c;
    if (need_mask) {
      const float* bias = sg.bias ? sg.bias + (long)(sg.kv_mod ? row % sg.kv_mod : row) * sg.bias_row_stride : nullptr;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = k0 + 32 * ks + 16 * (r >> 3) + 8 * fh + (r & 7);
          bool ok = key < nk;
          if (CAUSAL) ok = ok && (key <= q);
          float x = s[ks][r];
          if (bias && ok) x += bias[key] * 1.4426950408889634f;
          s[ks][r] = ok ? x : -INFINITY;
        }
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[ks][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_i, mx);
    const float alpha = __builtin_amdgcn_exp2f(m_i - m_new);
    float rs = 0.0f;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float pv = __builtin_amdgcn_exp2f(s[ks][r] - m_new);
        s[ks][r] = pv;
        rs += pv;
      }
    rs += __shfl_xor(rs, 32, 64);
    l_i = l_i * alpha + rs;
    m_i = m_new;
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[d][r] *= alpha;

    // ---- Oᵀ += Vᵀ · Pᵀ : 4 k-steps of 16 keys, 4 d sub-tiles of 32
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int st = 0; st < 2; ++st) {
        const bf16x8 pf = pack8(s[ks], 8 * st);
        const int chunk = 4 * ks + 2 * st + fh;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          const bf16x8 vf = *(const bf16x8*)(sv + (d * 32 + fr) * 128 + ((chunk ^ sw_v) << 4));
          o[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, o[d], 0, 0, 0);
        }
      }
    seg = nseg_; tile = ntile;
  }

  // ---- epilogue: lane holds O[q][32d + 8g + 4fh + 0..3]
  if (!q_ok) return;
  const float inv_l = 1.0f / l_i;
  bf16_t* op = p.O + (long)row * p.o_row_stride + (long)q * p.o_ld + head * HD;
  const bf16_t* gp = p.G ? p.G + (long)row * p.g_row_stride + (long)q * p.g_ld + head * HD : nullptr;
#pragma unroll
  for (int d = 0; d < 4; ++d)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int col = 32 * d + 8 * g + 4 * fh;
      float y[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) y[i] = bf2f(f2bf(o[d][4 * g + i] * inv_l));
      if (gp) {
        float gv[4];
        Vec4<bf16_t>::unpack(*(const uint2*)(gp + col), gv);
#pragma unroll
        for (int i = 0; i < 4; ++i) y[i] = y[i] * bf2f(f2bf(sigmoid_f(gv[i])));
      }
      *(uint2*)(op + col) = Vec4<bf16_t>::pack(y);
    }
}

}  // namespace

hipError_t launch_attention_bf16(const AttnArgs& a, hipStream_t st) {
  if (a.nseg < 1 || a.nseg > 4 || a.S < 1 || a.H < 1 || a.rows < 1) return hipErrorInvalidValue;
  if (a.causal && a.nseg != 1) return hipErrorInvalidValue;
  for (int s = 0; s < a.nseg; ++s)
    if ((a.seg[s].vt_ld & 7) || (a.seg[s].k_ld & 7) || !a.seg[s].nkeys) return hipErrorInvalidValue;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)attn_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute((const void*)attn_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  dim3 grid((a.S + QT - 1) / QT, a.H, a.rows);
  if (a.causal) hipLaunchKernelGGL(attn_kernel<true>, grid, dim3(256), SMEM, st, a);
  else hipLaunchKernelGGL(attn_kernel<false>, grid, dim3(256), SMEM, st, a);
  return hipGetLastError();
}
