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
#include <cstdlib>
#include <type_traits>

namespace {

constexpr int QT = 128;      // queries per workgroup
constexpr int KT = 64;       // keys per tile
constexpr int HD = 128;
constexpr float RESCALE_LOG2 = 4.0f;   // deferred-rescale threshold, in log2 units of the softmax argument
constexpr int K_TILE_BYTES = KT * HD * 2;    // 16 KiB, rows of 256 B
constexpr int V_TILE_BYTES = HD * KT * 2;    // 16 KiB, rows of 128 B
constexpr int MERGE_BYTES = 4 * 64 * 66 * 4;   // final merge of the two key halves: 4 query blocks x 64 lanes x (64 O + m + l) floats
constexpr int SMEM = MERGE_BYTES > 2 * (K_TILE_BYTES + V_TILE_BYTES) ? MERGE_BYTES : 2 * (K_TILE_BYTES + V_TILE_BYTES);

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
__device__ __forceinline__ void glds16(const char* src, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)lds_wave_base, 16, 0, 0);
}

__device__ __forceinline__ bf16x8 pack8(const f32x16& s, int base) {
  f32x8 v;
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = s[base + j];
  return __builtin_bit_cast(bf16x8, __builtin_convertvector(v, hbf16x8));   // 4 x v_cvt_pk_bf16_f32
}

// A stream walks the key tiles of the active segments in order.  Everything that changes per tile is one scalar pointer
// advanced by a scalar step; the per-lane part of a DMA source address is a 32-bit offset that is recomputed only when
// the stream enters a segment (and clamped on a segment's ragged last tile).  Three streams run 0 / 1 / 2 tiles ahead of
// the compute tile: `cur` (mask information), `vs` (V of tile t + 1) and `ks` (K of tile t + 2).
struct Stream { int seg, k0, nk; const char* ptr; int step; };

// Software-pipelined by one tile: the Sᵀ = K·Qᵀ MFMAs of tile t+1 are issued before the softmax (VALU) of tile t
// so the two overlap inside one wave; K tiles therefore run one tile ahead of V tiles in the LDS rings.
// fp8 engine with static activation scales (AttnArgs.O8): four outputs -> e4m3(bf16(y) * inv), saturating, one 32-bit store
__device__ __forceinline__ void store_o4_fp8(uint8_t* dst, const float (&y)[4], float inv) {
  float z[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) z[i] = __builtin_amdgcn_fmed3f(bf2f(f2bf(y[i])) * inv, -448.0f, 448.0f);
  int w = __builtin_amdgcn_cvt_pk_fp8_f32(z[0], z[1], 0, false);
  w = __builtin_amdgcn_cvt_pk_fp8_f32(z[2], z[3], w, true);
  *(int*)dst = w;
}
__device__ __forceinline__ unsigned long long stamp() {
  unsigned long long t;
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  __builtin_amdgcn_sched_barrier(0);
  return t;
}

// Work split (one workgroup per CU, 8 waves = two per SIMD): wave w owns the 32 queries of block qb = w & 3 and,
// inside every 64-key tile, the key half kh = w >> 2.  Each wave keeps its own online-softmax state (m, l, O) over its
// key subset; the two halves of a query block are merged once through LDS after the loop.  Two co-resident waves per
// SIMD let one wave's softmax VALU / LDS reads / DMA issue overlap the other's MFMAs, and every wave issues only 4 of
// the tile's 32 LDS-DMA pieces.
// One 128-query block (bxq) of (row, head); `smem` = K ring [2][16 KiB] | V ring [2][16 KiB] (+ merge area).  A device function so that
// attn_kernel runs it for its own blockIdx and attn_redo_kernel for the blocks the fast kernels flagged; a wave that `return`s here has
// only left this function.
template <bool CAUSAL, bool BIAS, bool PROF>
__device__ __forceinline__ void attn_body(const AttnArgs& p, char* const smem, const int bxq, const int head, const int row, const long prof_wg) {
  unsigned long long pt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long k_t0 = 0, k_r0 = 0, pa = 0, pb = 0;
  if (PROF) { k_t0 = __builtin_amdgcn_s_memtime(); k_r0 = __builtin_amdgcn_s_memrealtime(); }
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int qb = wid & 3, kh = wid >> 2;
  const int qbase = bxq * QT;
  const int fr = lane & 31, fh = lane >> 5;
  const int q = qbase + qb * 32 + fr;
  const bool q_ok = q < p.S;
  const int qc = q_ok ? q : p.S - 1;

  // ---- Q fragments (B operand of Sᵀ = K·Qᵀ): lane holds Q[q][16kk + 8fh + 0..7]
  bf16x8 qf[8];
  {
    const bf16_t* qp = p.Q + (long)row * p.q_row_stride + (long)qc * p.q_ld + head * HD + 8 * fh;
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) qf[kk] = *(const bf16x8*)(qp + 16 * kk);
  }

  // ---- per-segment key counts (wave-uniform); explicit selects instead of runtime-indexed arrays
  const int q_hi = min(qbase + QT, p.S) - 1;
  // all four counts are requested at once (independent scalar loads; inactive segments read segment 0's slot and are
  // zeroed afterwards): a branchy per-segment version serialised four dependent ~1 k-cycle round trips per workgroup
  int nkraw[4];
#pragma unroll
  for (int sgi = 0; sgi < 4; ++sgi) nkraw[sgi] = (sgi < p.nseg ? p.seg[sgi].nkeys : p.seg[0].nkeys)[row];
  auto seg_keys = [&](int sgi) -> int {
    int nk = sgi < p.nseg ? nkraw[sgi] : 0;
    if (CAUSAL) nk = min(nk, q_hi + 1);
    return nk < 0 ? 0 : nk;
  };
  const int nk0 = seg_keys(0), nk1 = seg_keys(1), nk2 = seg_keys(2), nk3 = seg_keys(3);
  auto NK = [&](int s) -> int { return s == 0 ? nk0 : s == 1 ? nk1 : s == 2 ? nk2 : s == 3 ? nk3 : 0; };
  auto NT = [&](int s) -> int { return (NK(s) + KT - 1) / KT; };
  const int total_tiles = NT(0) + NT(1) + NT(2) + NT(3);
  // ---- per-segment operands resolved ONCE into scalar registers (indexing the kernel-argument struct inside the tile
  // loop costs a chain of s_load + s_waitcnt per use: measured 1800 cycles per tile)
  const char* kb_[4]; const char* vb_[4]; long kld_[4], vld_[4]; const float* bias_[4];
  int mod0 = 0;     // the shared-KV segments all use the same modulus (the batch size): divide once
#pragma unroll
  for (int sgi = 0; sgi < 4; ++sgi)
    if (sgi < p.nseg && mod0 == 0 && p.seg[sgi].kv_mod > 0) mod0 = p.seg[sgi].kv_mod;
  const int rmod0 = mod0 > 0 ? row % mod0 : row;
#pragma unroll
  for (int sgi = 0; sgi < 4; ++sgi) {
    kb_[sgi] = nullptr; vb_[sgi] = nullptr; kld_[sgi] = 0; vld_[sgi] = 0; bias_[sgi] = nullptr;
    if (sgi < p.nseg) {
      const AttnSeg& sg = p.seg[sgi];
      const int kvrow = sg.kv_mod == 0 ? row : (sg.kv_mod == mod0 ? rmod0 : row % sg.kv_mod);   // one division per workgroup in practice
      kb_[sgi] = (const char*)(sg.K + (long)kvrow * sg.k_row_stride + (long)head * sg.k_head_stride);
      vb_[sgi] = (const char*)(sg.Vt + (long)kvrow * sg.vt_row_stride + (long)head * sg.vt_head_stride);
      kld_[sgi] = sg.k_ld * 2; vld_[sgi] = sg.vt_ld * 2;
      bias_[sgi] = sg.bias ? sg.bias + (long)kvrow * sg.bias_row_stride : nullptr;
    }
  }
#define SEL4(arr, s) ((s) == 0 ? arr[0] : (s) == 1 ? arr[1] : (s) == 2 ? arr[2] : arr[3])
  auto next_seg = [&](int sg) -> int {          // next segment with keys, 4 = none
    ++sg;
    while (sg < 4 && NK(sg) == 0) ++sg;
    return sg;
  };

  char* const kring = smem;
  char* const vring = smem + 2 * K_TILE_BYTES;
  // DMA roles: a tile is 16 K pieces (4 keys x 256 B) + 16 V pieces (8 d-rows x 128 B); wave w issues pieces 2w, 2w+1 of each
  int k_r[2], k_c[2], v_d[2], v_c[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    k_r[i] = (wid * 2 + i) * 4 + (lane >> 4);
    k_c[i] = ((lane & 15) ^ (k_r[i] & 15)) << 4;
    v_d[i] = (wid * 2 + i) * 8 + (lane >> 3);
    v_c[i] = ((lane & 7) ^ ((v_d[i] >> 1) & 7)) << 4;
  }
  // K stream: ptr = first key row of the tile; lane offset = (row of the piece, clamped to the segment's last key) * kld + chunk
  Stream ks, vs, cur;
  unsigned k_off[2], v_off[2];
  auto k_offsets = [&]() {
    const int kld = (int)SEL4(kld_, ks.seg), last = ks.nk - 1 - ks.k0;      // last >= 0
#pragma unroll
    for (int i = 0; i < 2; ++i) k_off[i] = (unsigned)(min(k_r[i], last) * kld + k_c[i]);
  };
  auto k_enter = [&](int sg) {
    ks.seg = sg; ks.k0 = 0; ks.nk = NK(sg); ks.ptr = SEL4(kb_, sg); ks.step = (int)SEL4(kld_, sg) * KT;
    k_offsets();
  };
  auto k_advance = [&]() {                       // past the end the stream stays on the last tile
    if (ks.k0 + KT < ks.nk) {
      ks.k0 += KT; ks.ptr += ks.step;
      if (ks.k0 + KT > ks.nk) k_offsets();       // ragged last tile of the segment: clamp the rows
    } else {
      const int sg = next_seg(ks.seg);
      if (sg < 4) k_enter(sg);
    }
  };
  auto v_enter = [&](int sg) {
    vs.seg = sg; vs.k0 = 0; vs.nk = NK(sg); vs.ptr = SEL4(vb_, sg); vs.step = KT * 2;
    const int vld = (int)SEL4(vld_, sg);
#pragma unroll
    for (int i = 0; i < 2; ++i) v_off[i] = (unsigned)(v_d[i] * vld + v_c[i]);
  };
  auto v_advance = [&]() {
    if (vs.k0 + KT < vs.nk) { vs.k0 += KT; vs.ptr += vs.step; }
    else { const int sg = next_seg(vs.seg); if (sg < 4) v_enter(sg); }
  };
  const float* cur_bias = nullptr;
  auto c_enter = [&](int sg) { cur.seg = sg; cur.k0 = 0; cur.nk = NK(sg); cur_bias = SEL4(bias_, sg); };
  auto c_advance = [&]() {
    if (cur.k0 + KT < cur.nk) cur.k0 += KT;
    else { const int sg = next_seg(cur.seg); if (sg < 4) c_enter(sg); }
  };
  auto stage_k = [&](int slot) {
    char* kb = kring + slot * K_TILE_BYTES + wid * 2048;
    glds16(ks.ptr + k_off[0], kb); glds16(ks.ptr + k_off[1], kb + 1024);
  };
  auto stage_v = [&](int slot) {
    char* vb = vring + slot * V_TILE_BYTES + wid * 2048;
    glds16(vs.ptr + v_off[0], vb); glds16(vs.ptr + v_off[1], vb + 1024);
  };

  const int pi_row = (fr & 0x13) | ((fr & 4) << 1) | ((fr & 8) >> 1);  // K rows are fed with bits 2,3 swapped
  const int sw_v = (lane >> 1) & 7;
  const int krow = kh * 32 + pi_row;           // this wave's key sub-tile
  const int sw_k = krow & 15;

  const float c = p.scale * 1.4426950408889634f;   // scores are exponentiated in the log2 domain
  float m_i = -1e30f, l_i = 0.0f;                   // reference point of the RAW scores, running denominator
  f32x16 o[4];
#pragma unroll
  for (int d = 0; d < 4; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[d][r] = 0.0f;

  auto load_k = [&](const char* sk, bf16x8 (&kf)[8]) {
    const char* kr = sk + krow * 256;
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) kf[kk] = *(const bf16x8*)(kr + (((2 * kk + fh) ^ sw_k) << 4));
  };
  auto load_v = [&](const char* sv, int st, bf16x8 (&vf)[4]) {   // keys 32kh + 16st .. +15 of the tile
    const int chunk = 4 * kh + 2 * st + fh;
#pragma unroll
    for (int d = 0; d < 4; ++d) vf[d] = *(const bf16x8*)(sv + (d * 32 + fr) * 128 + ((chunk ^ sw_v) << 4));
  };

  // masks this wave's half of the current tile in place (register r is key k0 + 32kh + 16(r>>3) + 8fh + (r&7)); rare path
  auto apply_mask = [&](f32x16& sc) {
    const int nk = cur.nk, k0 = cur.k0;
    const float* bias = cur_bias;
    const float inv_scale = 1.0f / p.scale;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = k0 + 32 * kh + 16 * (r >> 3) + 8 * fh + (r & 7);
      bool ok = key < nk;
      if (CAUSAL) ok = ok && (key <= q);
      float x = sc[r];
      if (BIAS) { if (bias) x += bias[ok ? key : 0] * inv_scale; }   // bias is added to the scaled score in the reference
      sc[r] = ok ? x : -INFINITY;
    }
  };
  auto tile_needs_mask = [&]() -> bool {
    return (cur.k0 + KT > cur.nk) || (BIAS && cur_bias != nullptr) || (CAUSAL && (cur.k0 + KT - 1 > qbase + qb * 32));
  };

  // One pipeline step: Sᵀ(t+1) for this wave's keys (8 MFMAs) paired with the first 8 exponentials of tile t, the first
  // PV k-step (4 MFMAs) paired with the other 8, then the second PV k-step.  sched_barrier(0) pins each {MFMA, VALU
  // shadow work, at most one DMA piece} group: left alone hipcc hoists the LDS-DMA pieces into a burst (which blocks
  // the wave) and clusters the MFMAs.  On the last tile the "next" K slot holds a stale but valid tile whose Sᵀ is
  // computed and dropped (no branch, small code).
  auto compute = [&](f32x16& scur, f32x16& snext, const char* sk_next, const char* sv, char* kdst, char* vdst) {
    bf16x8 kf[8], vf0[4], vf1[4];
    load_k(sk_next, kf);
    load_v(sv, 0, vf0);
    load_v(sv, 1, vf1);
    asm volatile("" : "+v"(scur));   // keep the score tile in architectural VGPRs (no v_accvgpr_read per VALU use)
    float mx = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) mx = fmaxf(mx, scur[r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    // deferred rescale (cdna_hip_programming.md T13): the reference point only moves when a row's maximum grew by
    // more than 2^RESCALE_LOG2; P may then exceed 1 (bf16 is floating point; l and O accumulate in fp32)
    const float m_new = fmaxf(m_i, mx);
    if (__any((m_new - m_i) * c > RESCALE_LOG2)) {
      const float alpha = __builtin_amdgcn_exp2f((m_i - m_new) * c);
      l_i *= alpha;
#pragma unroll
      for (int d = 0; d < 4; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[d][r] *= alpha;
      m_i = m_new;
    }
    const float mc = m_i * c;
#pragma unroll
    for (int r = 0; r < 16; ++r) snext[r] = 0.0f;
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
      snext = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[kk], qf[kk], snext, 0, 0, 0);
      // no consumer right behind the exponential (an exp -> add chain stalls the in-order wave); the row sum is formed at the end
      scur[kk] = __builtin_amdgcn_exp2f(__builtin_fmaf(scur[kk], c, -mc));
      if (kk == 1) glds16(ks.ptr + k_off[0], kdst);            // K(t+2)
      if (kk == 3) glds16(ks.ptr + k_off[1], kdst + 1024);
      if (kk == 5) glds16(vs.ptr + v_off[0], vdst);            // V(t+1)
      if (kk == 7) glds16(vs.ptr + v_off[1], vdst + 1024);
      __builtin_amdgcn_sched_barrier(0);
    }
    {
      const bf16x8 pf = pack8(scur, 0);
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        o[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf0[d], pf, o[d], 0, 0, 0);
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const int r = 8 + 2 * d + e;
          scur[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(scur[r], c, -mc));
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    {
      const bf16x8 pf = pack8(scur, 8);
#pragma unroll
      for (int d = 0; d < 4; ++d) o[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf1[d], pf, o[d], 0, 0, 0);
    }
    // row sum in the shadow of the last MFMAs: four independent chains, then a tree
    float r0 = scur[0] + scur[4], r1 = scur[1] + scur[5], r2 = scur[2] + scur[6], r3 = scur[3] + scur[7];
    r0 += scur[8]; r1 += scur[9]; r2 += scur[10]; r3 += scur[11];
    r0 += scur[12]; r1 += scur[13]; r2 += scur[14]; r3 += scur[15];
    float rs = (r0 + r1) + (r2 + r3);
    rs += __shfl_xor(rs, 32, 64);
    l_i += rs;
  };

  if (PROF) pa = stamp() - k_t0;      // Q fragments requested, key counts and segment operands resolved
  if (total_tiles > 0) {
    {
      int s0 = 0;
      while (s0 < 3 && NK(s0) == 0) ++s0;
      k_enter(s0); v_enter(s0); c_enter(s0);
    }
    stage_k(0);                       // K(0)
    stage_v(0);                       // V(0)
    k_advance();
    if (total_tiles > 1) stage_k(1);  // K(1)
    __syncthreads();   // vmcnt(0) + barrier
    if (PROF) pb = stamp() - k_t0;    // first K / V tiles landed
    f32x16 sa, sb;
    {
      bf16x8 kf[8];
      load_k(kring, kf);
#pragma unroll
      for (int r = 0; r < 16; ++r) sa[r] = 0.0f;
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) sa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[kk], qf[kk], sa, 0, 0, 0);
    }
    k_advance();                      // ks -> tile 2, vs -> tile 1, cur = tile 0
    v_advance();
    if (PROF) pt[7] = __builtin_amdgcn_s_memtime() - k_t0;   // prologue
    // One step: tile t is in `scur`; Sᵀ of tile t+1 is produced into `snext`.  The loop is unrolled by two with the score
    // tiles exchanging roles and the ring slots as compile-time constants (PAR = t & 1): no 16-register copy per tile and
    // the LDS read / DMA addresses are a per-lane offset + an immediate.
    auto step = [&](auto par, f32x16& scur, f32x16& snext) __attribute__((always_inline)) {
      constexpr int PAR = decltype(par)::value;
      unsigned long long t0 = 0, t1 = 0, t2 = 0;
      if (PROF) t0 = stamp();
      // Every wave's DMA pieces of K(t+1) / V(t) must have landed before anyone reads them.  hipcc does NOT put the
      // vmcnt wait into __syncthreads() here (it only tracked the prologue's DMAs): without this explicit wait the
      // kernel was non-deterministic at full size (caught by tests/test_gpu_kernels.py::test_attention_is_deterministic).
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (PROF) { const unsigned long long tv = stamp(); pt[4] += tv - t0; }
      __syncthreads();   // K(t+1), V(t) landed; every wave finished QK(t) and PV(t-1)
      if (PROF) t1 = stamp();
      // next DMA targets: K(t+2) -> K slot t&1, V(t+1) -> V slot (t+1)&1.  Past the end the streams stay on the last
      // tile: the redundant pieces land in slots nobody reads again (and are drained before the merge).
      char* kdst = kring + PAR * K_TILE_BYTES + wid * 2048;
      char* vdst = vring + (PAR ^ 1) * V_TILE_BYTES + wid * 2048;
      if (tile_needs_mask()) apply_mask(scur);
      if (PROF) t2 = stamp();
      compute(scur, snext, kring + (PAR ^ 1) * K_TILE_BYTES, vring + PAR * V_TILE_BYTES, kdst, vdst);
      if (PROF) { const unsigned long long t3 = stamp(); pt[0] += t1 - t0; pt[1] += t2 - t1; pt[2] += t3 - t2; pt[3] += 1; }
      k_advance(); v_advance(); c_advance();
    };
    int t = 0;
#pragma unroll 1
    for (; t + 1 < total_tiles; t += 2) {
      step(std::integral_constant<int, 0>{}, sa, sb);
      step(std::integral_constant<int, 1>{}, sb, sa);
    }
    if (t < total_tiles) step(std::integral_constant<int, 0>{}, sa, sb);   // odd tile count: the last tile has even parity
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no LDS-DMA may be in flight when the rings are reused / the workgroup ends
  if (PROF && lane == 0) {
    unsigned long long* dst = (unsigned long long*)p.prof + (prof_wg * 8 + wid) * 8;
    dst[0] = pt[0]; dst[1] = pt[1]; dst[2] = pt[2]; dst[3] = pt[3]; dst[4] = pt[4];
    dst[5] = __builtin_amdgcn_s_memtime() - k_t0; dst[6] = __builtin_amdgcn_s_memrealtime() - k_r0; dst[7] = pt[7] | (pa << 20) | (pb << 40);   // up to the end of the tile loop
  }

  // the output gate values of this lane's 64 columns are requested now (one batch of 16 x 8 bytes, kh = 0 waves only): issued
  // one by one inside the store loop each load would wait behind the previous store (a single in-order vmcnt) - 16 L2 round
  // trips per workgroup instead of one that is hidden by the merge below
  uint2 gq[4][4];
  {
    const bf16_t* gp0 = p.G ? p.G + (long)row * p.g_row_stride + (long)qc * p.g_ld + head * HD + 4 * fh : nullptr;
    if (gp0 && kh == 0) {
#pragma unroll
      for (int d = 0; d < 4; ++d)
#pragma unroll
        for (int g = 0; g < 4; ++g) gq[d][g] = *(const uint2*)(gp0 + 32 * d + 8 * g);
    }
  }
  // ---- merge the two key halves of each query block: waves kh = 1 park (O, m, l) in LDS, waves kh = 0 combine
  __syncthreads();
  float* mo = (float*)smem + qb * (64 * 66);      // per query block: 64 lanes x (64 O values + m + l), lane-major stride 66
  if (kh == 1) {
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
      for (int r = 0; r < 16; ++r) mo[(d * 16 + r) * 64 + lane] = o[d][r];
    mo[64 * 64 + lane] = m_i;
    mo[65 * 64 + lane] = l_i;
  }
  __syncthreads();
  if (kh == 1) return;
  {
    const float m_b = mo[64 * 64 + lane], l_b = mo[65 * 64 + lane];
    const float m = fmaxf(m_i, m_b);
    const float fa = __builtin_amdgcn_exp2f((m_i - m) * c), fb = __builtin_amdgcn_exp2f((m_b - m) * c);
    l_i = l_i * fa + l_b * fb;
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[d][r] = o[d][r] * fa + mo[(d * 16 + r) * 64 + lane] * fb;
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
        Vec4<bf16_t>::unpack(gq[d][g], gv);
#pragma unroll
        for (int i = 0; i < 4; ++i) y[i] = y[i] * (p.g_act ? gv[i] : bf2f(f2bf(sigmoid_fast(gv[i]))));
      }
      if (p.O8) store_o4_fp8(p.O8 + (long)row * p.o8_row_stride + (long)q * p.o8_ld + head * HD + col, y, p.o8_inv);
      else *(uint2*)(op + col) = Vec4<bf16_t>::pack(y);
    }
}

template <bool CAUSAL, bool BIAS, bool PROF>
__global__ void __launch_bounds__(512, 2) attn_kernel(const AttnArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];   // K ring [2][16 KiB] | V ring [2][16 KiB]
  // (ECHO_ATTN_REDO_FULL=1, A/B aid: the second pass as a full grid behind this filter instead of attn_redo_kernel)
  if (p.redo_filter && p.redo_filter[((long)blockIdx.z * gridDim.y + blockIdx.y) * p.redo_nb + ((blockIdx.x + p.q_block0) >> (p.q128 ? 0 : 1))] == 0) return;
  attn_body<CAUSAL, BIAS, PROF>(p, smem, blockIdx.x + p.q_block0, blockIdx.y, blockIdx.z, ((long)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x);
}

// Second pass behind attn4_kernel / attn5_kernel: the blocks whose scores left the fast kernels' range (normally none) are done again
// with per-tile rescaling.  A handful of workgroups, each scanning its share of the report words in one coalesced read and leaving at
// once when none is set: as a full attn_kernel grid behind a filter, the pass put one 8-wave, 100 KiB-LDS workgroup per block (2 880 for a
// CFG step of 24 utterances) into the queue behind every attention launch - 14-32 us each on the stream's critical path when the
// other stream's GEMM holds the CUs (1.3 % of the kernel time of profiles/r02_bench_kernel_stats_v8.csv).
__global__ void __launch_bounds__(512, 2) attn_redo_kernel(const AttnArgs p, const int nwords, const int per_wg) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int base = blockIdx.x * per_wg;
  const int n = min(per_wg, nwords - base);
  int flag = 0;
  for (int i = threadIdx.x; i < n; i += 512) flag |= p.redo_filter[base + i];
  if (!__syncthreads_or(flag)) return;
  const int nsub = p.q128 ? 1 : 2;            // a report word covers one 128-query block, or the two of a 256-query workgroup
  for (int i = 0; i < n; ++i) {
    if (p.redo_filter[base + i] == 0) continue;          // workgroup-uniform
    const int w = base + i, blk = w % p.redo_nb, rh = w / p.redo_nb;
    for (int sb = 0; sb < nsub; ++sb) {
      const int bxq = blk * nsub + sb;
      if (bxq * QT >= p.S) break;
      attn_body<false, false, false>(p, smem, bxq, rh % p.H, rh / p.H, 0);
      __syncthreads();                                   // the rings and the merge area are reused by the next block
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// attn4_kernel: the joint attention of the EchoDiT steps (no causal mask, no per-key bias) as 4 waves x 64 queries, ONE wave
// per SIMD on the 512-entry register file (cdna_hip_programming.md Appendix B "4-wave, one-wave-per-SIMD"; round-1 DESIGN §3.2:
// attn_kernel's two waves per SIMD overlap their MFMA / VALU / LDS streams only partly and end with an LDS merge).
//
// One workgroup = 256 queries of one (row, head); wave w owns queries 64 w .. 64 w + 63 as two INDEPENDENT 32-query streams
// q = 0, 1 that run half a tile apart inside one instruction stream: while the matrix pipe works on one stream's 16 MFMAs, the
// vector ALU runs the other stream's softmax in the MFMA gaps.  MFMA order per 64-key tile t:
//     S0(t) = K(t) Q0^T     PV1(t-1)     S1(t) = K(t) Q1^T     PV0(t)           (16 x v_mfma_f32_32x32x16_bf16 each)
// softmax of stream 0 / tile t occupies the 32 gaps of PV1(t-1) and S1(t), softmax of stream 1 those of PV0(t) and S0(t+1): every
// softmax is complete before the PV phase that consumes its P.  Live score registers: 2 x 32.  Fragments come from LDS one per
// MFMA, two MFMAs ahead.  Same LDS images, LDS-DMA, segment streams, key permutation and in-register P as attn_kernel; rings of
// three K and three V^T tiles (K two tiles ahead, V one), one workgroup barrier per tile, the loop unrolled by three so that ring
// slots are immediates.
//
// The tile loop contains NO branch that modifies score or output registers: hipcc answers such a branch (a conditional rescale of
// O, a conditional masking of S) with v_accvgpr round trips of the accumulators on the hot path - measured 2x SLOWER than
// attn_kernel.  Instead:
//  * masking is an MFMA operand: every score chain starts from cz[kb] = 0 for valid keys, -inf for the keys past a segment's end
//    (S = K Q^T + cz), recomputed branch-free every step in place of the zero the chains would need anyway;
//  * the reference point m of a stream is FIXED at its first tile's row maximum and every tile is exponentiated against it: exact
//    as long as no later score exceeds it by more than 2^FAST_LOG2_RANGE in the log2 domain (P <= 2^40: bf16 has fp32's exponent
//    range, l and O accumulate in fp32) - RMS-normalised q, k stay far inside.  A workgroup that sees a score beyond the range
//    reports it in `redo`, and launch_attention_bf16 runs attn_kernel (per-tile rescaling) right behind for exactly those
//    workgroups: always correct, fast whenever the scores behave.
constexpr float FAST_LOG2_RANGE = 40.0f;
constexpr int SMEM4 = 3 * (K_TILE_BYTES + V_TILE_BYTES) + 16;

template <int I, int N, typename F>
__device__ __forceinline__ void sfor(F&& f) {
  if constexpr (I < N) { f(std::integral_constant<int, I>{}); sfor<I + 1, N>(f); }
}
__device__ __forceinline__ float half_max(float v) {      // max over lanes l and l ^ 32
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float half_sum(float v) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// max of s[h .. h + 7] as one statement: hipcc pads a wait state behind every asm statement whose output a VALU reads next
__device__ __forceinline__ float max8(const f32x16& s, int h) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3\n\tv_max3_f32 %0, %0, %4, %5\n\tv_max3_f32 %0, %0, %6, %7\n\tv_max_f32 %0, %0, %8"
      : "=&v"(r) : "v"(s[h]), "v"(s[h + 1]), "v"(s[h + 2]), "v"(s[h + 3]), "v"(s[h + 4]), "v"(s[h + 5]), "v"(s[h + 6]), "v"(s[h + 7]));
  return r;
}
__device__ __forceinline__ float max3(float a, float b, float c) {   // no canonicalising v_max in front (fmaxf on an MFMA / asm result gets one)
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
// Score MFMAs in the VGPR form (D / C in arch VGPRs, Q in the accumulator file): the softmax reads the scores with the vector ALU, and
// hipcc's AGPR-form chain costs a v_accvgpr_read per element (80 of the 488 VALU issues of a step).  hipcc pads no hazards of an asm
// MFMA (cdna_hip_programming.md §5.7 item 2): the first reader of a finished chain is a softmax slot at least two MFMA issues
// (>= 64 cycles) later by construction of the phases - audit the ISA for compiler v_mov of these registers after every edit.
__device__ __forceinline__ void mfma_s_first(f32x16& s, const bf16x8& kf, const bf16x8& q) {
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(s) : "v"(kf), "a"(q));
}
__device__ __forceinline__ void mfma_s_next(f32x16& s, const bf16x8& kf, const bf16x8& q) {
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(s) : "v"(kf), "a"(q));
}

// LDS fragment read hidden from hipcc's wait-count bookkeeping (cdna_hip_programming.md §5.7 item 1, form iii): hipcc waits
// lgkmcnt(0) in front of every third MFMA of a rolling three-fragment window - i.e. also for the read it has just issued - which
// exposes a full LDS round trip 21 times per tile (measured: 2.7 us per tile instead of ~1.2).  Here every MFMA gap issues exactly
// one read and lds_wait<2>() in front of MFMA g retires exactly fragment g (reads return in order; the loop has no other LGKM user).
__device__ __forceinline__ bf16x8 lds_read16(unsigned addr, int off) {
  bf16x8 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "i"(off));
  return v;
}
template <int N> __device__ __forceinline__ void lds_wait() {
  asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
  __builtin_amdgcn_sched_barrier(0);          // no MFMA may be hoisted above the wait (guide rule 18)
}

struct Soft {            // one query stream of a wave
  f32x16 s[2];           // raw scores of the current tile, [kb]: register r = key 32 kb + 16 (r >> 3) + 8 fh + (r & 7); P after the exponentials
  bf16x8 pf[4];          // P^T as the B operand of the four PV k-steps
  float m, l, mc;        // reference point of the raw scores, running denominator, m * c
  float mxp[4], rs[4];
};

// DIAG bits (timing experiments only, wrong results): 1 = no softmax VALU work, 2 = no MFMAs, 4 = no LDS-DMA in the loop, 8 = no barrier / vmcnt wait per tile
template <int DIAG>
__global__ void __launch_bounds__(256, 1) attn4_kernel(const AttnArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];   // K ring [3][16 KiB] | V ring [3][16 KiB] | 4 range flags
  constexpr int QW = 256;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int row = blockIdx.z, head = blockIdx.y;
  const int qbase = blockIdx.x * QW;
  const int fr = lane & 31, fh = lane >> 5;
  const bool wave_on = qbase + wid * 64 < p.S;          // wave-uniform: a wave without queries only stages tiles
  char* const kring = smem;
  char* const vring = smem + 3 * K_TILE_BYTES;
  int* const wflags = (int*)(smem + 3 * (K_TILE_BYTES + V_TILE_BYTES));

  // ---- Q fragments (B operand of S^T = K Q^T): lane holds Q[q][16 kk + 8 fh + 0..7] for both streams
  bf16x8 qf[2][8];
  int qi[2]; bool q_ok[2];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    qi[qb] = qbase + wid * 64 + qb * 32 + fr;
    q_ok[qb] = qi[qb] < p.S;
    const int qc = q_ok[qb] ? qi[qb] : p.S - 1;
    const bf16_t* qp = p.Q + (long)row * p.q_row_stride + (long)qc * p.q_ld + head * HD + 8 * fh;
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) qf[qb][kk] = *(const bf16x8*)(qp + 16 * kk);
  }

  // ---- per-segment key counts and operands, resolved once into scalar registers (see attn_kernel)
  int nkraw[4];
#pragma unroll
  for (int sgi = 0; sgi < 4; ++sgi) nkraw[sgi] = (sgi < p.nseg ? p.seg[sgi].nkeys : p.seg[0].nkeys)[row];
  auto seg_keys = [&](int sgi) -> int { const int nk = sgi < p.nseg ? nkraw[sgi] : 0; return nk < 0 ? 0 : nk; };
  const int nk0 = seg_keys(0), nk1 = seg_keys(1), nk2 = seg_keys(2), nk3 = seg_keys(3);
  auto NK = [&](int s) -> int { return s == 0 ? nk0 : s == 1 ? nk1 : s == 2 ? nk2 : s == 3 ? nk3 : 0; };
  auto NT = [&](int s) -> int { return (NK(s) + KT - 1) / KT; };
  const int total_tiles = NT(0) + NT(1) + NT(2) + NT(3);
  const char* kb_[4]; const char* vb_[4]; long kld_[4], vld_[4];
  int mod0 = 0;
#pragma unroll
  for (int sgi = 0; sgi < 4; ++sgi)
    if (sgi < p.nseg && mod0 == 0 && p.seg[sgi].kv_mod > 0) mod0 = p.seg[sgi].kv_mod;
  const int rmod0 = mod0 > 0 ? row % mod0 : row;
#pragma unroll
  for (int sgi = 0; sgi < 4; ++sgi) {
    kb_[sgi] = nullptr; vb_[sgi] = nullptr; kld_[sgi] = 0; vld_[sgi] = 0;
    if (sgi < p.nseg) {
      const AttnSeg& sg = p.seg[sgi];
      const int kvrow = sg.kv_mod == 0 ? row : (sg.kv_mod == mod0 ? rmod0 : row % sg.kv_mod);
      kb_[sgi] = (const char*)(sg.K + (long)kvrow * sg.k_row_stride + (long)head * sg.k_head_stride);
      vb_[sgi] = (const char*)(sg.Vt + (long)kvrow * sg.vt_row_stride + (long)head * sg.vt_head_stride);
      kld_[sgi] = sg.k_ld * 2; vld_[sgi] = sg.vt_ld * 2;
    }
  }
  auto next_seg = [&](int sg) -> int { ++sg; while (sg < 4 && NK(sg) == 0) ++sg; return sg; };

  // DMA roles: a tile is 16 K pieces (4 keys x 256 B) + 16 V pieces (8 d-rows x 128 B); wave w issues pieces 4 w .. 4 w + 3 of each
  int k_r[4], k_c[4], v_d[4], v_c[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    k_r[i] = (wid * 4 + i) * 4 + (lane >> 4);
    k_c[i] = ((lane & 15) ^ (k_r[i] & 15)) << 4;
    v_d[i] = (wid * 4 + i) * 8 + (lane >> 3);
    v_c[i] = ((lane & 7) ^ ((v_d[i] >> 1) & 7)) << 4;
  }
  // ONE tile walk, three readers: d2 = tile t + 2 (its K is staged during step t), d1 = tile t + 1 (its V), d0 = tile t (masking).
  // Only d2 is advanced (one compare and two pointer increments on the common path, the segment switch out of line); d1 and d0 are
  // last step's d2 and d1.  All scalar: every wave computes the same values.
  struct Tile { int seg, k0, nk; const char* kptr; const char* vptr; int kstep; };
  Tile d2, d1;
  int d0_valid = 0, vseg = -1;
  unsigned k_off[4], v_off[4];
  auto k_offsets = [&]() {
    const int kld = (int)SEL4(kld_, d2.seg), last = d2.nk - 1 - d2.k0;      // rows past the segment's last key repeat it (masked in the softmax)
#pragma unroll
    for (int i = 0; i < 4; ++i) k_off[i] = (unsigned)(min(k_r[i], last) * kld + k_c[i]);
  };
  auto enter = [&](int sg) {
    d2.seg = sg; d2.k0 = 0; d2.nk = NK(sg); d2.kptr = SEL4(kb_, sg); d2.vptr = SEL4(vb_, sg); d2.kstep = (int)SEL4(kld_, sg) * KT;
    k_offsets();
  };
  auto advance = [&]() {                         // past the end the walk stays on the last tile
    if (__builtin_expect(d2.k0 + 2 * KT <= d2.nk, 1)) { d2.k0 += KT; d2.kptr += d2.kstep; d2.vptr += KT * 2; }
    else if (d2.k0 + KT < d2.nk) { d2.k0 += KT; d2.kptr += d2.kstep; d2.vptr += KT * 2; k_offsets(); }
    else { const int sg = next_seg(d2.seg); if (sg < 4) enter(sg); }
  };
  auto v_offsets = [&]() {                       // per-lane offsets of the V^T pieces of d1's segment
    if (__builtin_expect(d1.seg != vseg, 0)) {
      vseg = d1.seg;
      const int vld = (int)SEL4(vld_, d1.seg);
#pragma unroll
      for (int i = 0; i < 4; ++i) v_off[i] = (unsigned)(v_d[i] * vld + v_c[i]);
    }
  };
  auto dma_k = [&](int i, char* kdst) __attribute__((always_inline)) { glds16(d2.kptr + k_off[i], kdst + i * 1024); };
  auto dma_v = [&](int i, char* vdst) __attribute__((always_inline)) { glds16(d1.vptr + v_off[i], vdst + i * 1024); };

  const int pi_row = (fr & 0x13) | ((fr & 4) << 1) | ((fr & 8) >> 1);  // K rows are fed with bits 2,3 swapped (see attn_kernel)
  const int sw_v = (lane >> 1) & 7;
  const int sw_k = pi_row & 15;
  // fragment read addresses (bytes from the start of a tile image), + kb * 8192 / + db * 4096 as immediates
  // (the ring base is part of the per-lane pointer, so that slot, kb and db offsets fold into the ds_read's offset field)
  unsigned ka[8], va[4];
  {
    const unsigned kbase = (unsigned)(unsigned long)(lptr_t)kring, vbase = (unsigned)(unsigned long)(lptr_t)vring;
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) ka[kk] = kbase + pi_row * 256 + (((2 * kk + fh) ^ sw_k) << 4);
#pragma unroll
    for (int kss = 0; kss < 4; ++kss) va[kss] = vbase + fr * 128 + (((2 * kss + fh) ^ sw_v) << 4);
  }
  const float c = p.scale * 1.4426950408889634f;
  f32x16 o[2][4];
  Soft st[2];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[qb][d][r] = 0.0f;
    st[qb].m = -1e30f; st[qb].l = 0.0f; st[qb].mc = -1e30f * c;
    // "tile -1" (makes the first step uniform): P = 0 everywhere.  The elements the skipped slots 6-15 would have exponentiated
    // (e < 20: kb 0, and registers 0-3 of kb 1) are already 0, the others are raw -inf and become 0 in slots 16-21
#pragma unroll
    for (int r = 0; r < 16; ++r) { st[qb].s[0][r] = 0.0f; st[qb].s[1][r] = r < 4 ? 0.0f : -INFINITY; }
#pragma unroll
    for (int i = 0; i < 4; ++i) { st[qb].pf[i] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0}; st[qb].mxp[i] = 0.f; st[qb].rs[i] = 0.f; }
  }
  int ovf = 0;             // a score left the fixed reference's range (wave-uniform)
  bool ragged = false;     // the current tile has fewer than KT valid keys (wave-uniform)
  float mask_x = 0.0f;     // valid keys of the current tile - 8 fh

  // ---- softmax of stream q, cut into 32 slots that ride in the MFMA gaps of the other stream
  auto soft_slot = [&](auto qc_, auto ic_) __attribute__((always_inline)) {
    constexpr int q = decltype(qc_)::value, I = decltype(ic_)::value;
    Soft& S = st[q];
    if constexpr (I < 4) {
      constexpr int kb = I >> 1, h = 8 * (I & 1);
      if (__builtin_expect(ragged, 0)) {            // keys past the segment's end: a hugely negative addend (wave-uniform, the last tile of a segment only)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float mk = fminf(0.0f, __builtin_fmaf(mask_x, 1e30f, -((float)(32 * kb + 16 * ((h + j) >> 3) + ((h + j) & 7)) + 0.5f) * 1e30f));
          asm volatile("v_add_f32 %0, %0, %1" : "+v"(S.s[kb][h + j]) : "v"(mk));
        }
      }
      if constexpr (DIAG & 32) {
        const float a = fmaxf(fmaxf(S.s[kb][h], S.s[kb][h + 1]), S.s[kb][h + 2]);
        const float b = fmaxf(fmaxf(S.s[kb][h + 3], S.s[kb][h + 4]), S.s[kb][h + 5]);
        S.mxp[I] = fmaxf(fmaxf(a, b), fmaxf(S.s[kb][h + 6], S.s[kb][h + 7]));
      } else S.mxp[I] = max8(S.s[kb], h);
    } else if constexpr (I == 4) {
      if constexpr (DIAG & 32) S.mxp[0] = half_max(fmaxf(fmaxf(S.mxp[0], S.mxp[1]), fmaxf(S.mxp[2], S.mxp[3])));
      else S.mxp[0] = half_max(max3(max3(S.mxp[0], S.mxp[1], S.mxp[2]), S.mxp[3], S.mxp[3]));
    } else if constexpr (I == 5) {
      const float mx = S.mxp[0];
      S.m = S.m == -1e30f ? mx : S.m;                        // the first tile sets the reference; O and l are still 0
      if (__any((mx - S.m) * c > FAST_LOG2_RANGE)) ovf = 1;  // only a scalar is written in here
      S.mc = S.m * c;
    } else if constexpr (I < 22) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        constexpr int e0 = 2 * (I - 6);
        const int e = e0 + u;
        const int kss = e >> 3, kb = kss >> 1, r = 8 * (kss & 1) + (e & 7);
        S.s[kb][r] = __builtin_amdgcn_exp2f(__builtin_fmaf(S.s[kb][r], c, -S.mc));
      }
      if constexpr (((2 * (I - 6) + 1) & 7) == 7) {         // the 8 values of PV k-step kss are exponentiated: pack them
        constexpr int kss = (2 * (I - 6)) >> 3;
        S.pf[kss] = pack8(S.s[kss >> 1], 8 * (kss & 1));
      }
    } else if constexpr (I < 30) {
      constexpr int g = I - 22;                             // elements 4 g .. 4 g + 3
      constexpr int kss = g >> 1, kb = kss >> 1, r0 = 8 * (kss & 1) + 4 * (g & 1);
      const float t = (S.s[kb][r0] + S.s[kb][r0 + 1]) + (S.s[kb][r0 + 2] + S.s[kb][r0 + 3]);
      if constexpr (g < 4) S.rs[g] = t; else S.rs[g - 4] += t;
    } else if constexpr (I == 30) {
      S.rs[0] = half_sum((S.rs[0] + S.rs[1]) + (S.rs[2] + S.rs[3]));
    } else {
      S.l += S.rs[0];
    }
  };

  // ---- MFMA phases.  VQ: the stream whose softmax rides in the gaps (-1 = none), slots SL0 .. SL0 + 15.  Every MFMA takes its
  // LDS fragment from the rolling window fq (two MFMAs ahead); the last two gaps of a phase already read the first two fragments of
  // the NEXT phase (nxt0 / nxt1: their LDS byte addresses), so no phase starts with an exposed LDS round trip.
  // FW fragments deep: with 3 in flight per wave (12 KiB per CU) the kernel ran at the LDS LATENCY, 1.36 us per tile for the reads alone
  constexpr int FW = 8;
  bf16x8 fq[FW];
  // fragment j of a phase: kind 0 = K of S_q (kb = j >> 3, kk = j & 7), kind 1 = V^T of PV_q (kss = j >> 2, db = j & 3); slot = ring slot (a literal)
  auto frag = [&](auto kindc, const int slot, auto jc) __attribute__((always_inline)) -> bf16x8 {
    constexpr int kind = decltype(kindc)::value, j = decltype(jc)::value;
    if constexpr (kind == 0) return lds_read16(ka[j & 7], slot * K_TILE_BYTES + (j >> 3) * 8192);
    else return lds_read16(va[j >> 2], slot * V_TILE_BYTES + (j & 3) * 4096);
  };
  // one phase of 16 MFMAs.  KIND 0: S_q(t) = K(t) Q_q^T + cz (kb, kk); KIND 1: PV_q: O_q^T += V^T(tile) P_q^T (kss, db).  slot: its ring slot;
  // nslot: the ring slot of the NEXT phase (which is of the other kind), whose first FW - 1 fragments the last gaps already read.
  // DMA (S phases): this wave's 4 K + 4 V pieces at g = 0, 2, .. 14.
  auto phase = [&](auto kindc, auto qc_, const int slot, const int nslot, auto vq_, auto sl0_, auto dmac_, char* kdst, char* vdst) __attribute__((always_inline)) {
    constexpr int KIND = decltype(kindc)::value, q = decltype(qc_)::value, VQ = decltype(vq_)::value, SL0 = decltype(sl0_)::value;
    constexpr bool DMA = decltype(dmac_)::value != 0;
    sfor<0, 16>([&](auto gc) __attribute__((always_inline)) {
      constexpr int g = decltype(gc)::value, j = g + FW - 1;
      if constexpr (j < 16) fq[j % FW] = frag(kindc, slot, std::integral_constant<int, (j < 16 ? j : 0)>{});
      else fq[j % FW] = frag(std::integral_constant<int, 1 - KIND>{}, nslot, std::integral_constant<int, (j >= 16 ? j - 16 : 0)>{});
      lds_wait<FW - 1>();
      if constexpr (DIAG & 2) asm volatile("" :: "v"(fq[g % FW]));
      else if constexpr (KIND == 0) {
        constexpr int kb = g >> 3, kk = g & 7;
        if constexpr (DIAG & 16) {
          if constexpr (kk == 0) st[q].s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fq[g % FW], qf[q][kk], f32x16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, 0, 0, 0);
          else st[q].s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fq[g % FW], qf[q][kk], st[q].s[kb], 0, 0, 0);
        } else if constexpr (kk == 0) mfma_s_first(st[q].s[kb], fq[g % FW], qf[q][kk]);
        else mfma_s_next(st[q].s[kb], fq[g % FW], qf[q][kk]);
      } else {
        constexpr int kss = g >> 2, db = g & 3;
        o[q][db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fq[g % FW], st[q].pf[kss], o[q][db], 0, 0, 0);
      }
      if constexpr (!(DIAG & 64)) __builtin_amdgcn_sched_barrier(0);          // the gap's vector work stays BEHIND its MFMA (hipcc hoisted it in front of every second one)
      if constexpr (VQ >= 0 && !(DIAG & 1)) soft_slot(std::integral_constant<int, (VQ < 0 ? 0 : VQ)>{}, std::integral_constant<int, SL0 + g>{});
      if constexpr (DMA && (g & 1) == 0 && !(DIAG & 4)) { if constexpr (g < 8) dma_k(g >> 1, kdst); else dma_v((g - 8) >> 1, vdst); }
      __builtin_amdgcn_sched_barrier(0);
    });
  };
  // The look-ahead reads at the end of the LAST step (and of the drain) have no consumer: without a use behind the wait hipcc treats
  // their destination registers as free while the reads are still in flight and puts live values there (seen: exponentials of
  // stream 1 overwritten by the late LDS return whenever total_tiles % 3 == 2).  A use of the whole window behind every wait.
  auto keep_window = [&]() __attribute__((always_inline)) {
    static_assert(FW == 8, "lists the whole window");
    asm volatile("" ::"v"(fq[0]), "v"(fq[1]), "v"(fq[2]), "v"(fq[3]), "v"(fq[4]), "v"(fq[5]), "v"(fq[6]), "v"(fq[7]));
  };
  typedef std::integral_constant<int, 0> I0;
  typedef std::integral_constant<int, 1> I1;
  typedef std::integral_constant<int, 16> I16;
  typedef std::integral_constant<int, -1> IN;

  if (total_tiles > 0) {
    {
      int s0 = 0;
      while (s0 < 3 && NK(s0) == 0) ++s0;
      enter(s0);
    }
    d1 = d2; v_offsets();
    d0_valid = d2.nk - d2.k0;
#pragma unroll
    for (int i = 0; i < 4; ++i) dma_k(i, kring + wid * 4096);                          // K(0) -> K slot 0
#pragma unroll
    for (int i = 0; i < 4; ++i) dma_v(i, vring + wid * 4096);                          // V(0) -> V slot 0
#pragma unroll
    for (int i = 0; i < 4; ++i) dma_v(i, vring + 2 * V_TILE_BYTES + wid * 4096);       // and -> V slot 2: finite operands for PV1(-1), whose P is 0
    advance();                        // d2 = tile 1
    if (total_tiles > 1) {
#pragma unroll
      for (int i = 0; i < 4; ++i) dma_k(i, kring + K_TILE_BYTES + wid * 4096);         // K(1) -> K slot 1
    }
    d1 = d2; v_offsets();             // d1 = tile 1
    advance();                        // d2 = tile 2
    // step t (slot sl = t % 3, a literal): reads K(t) [K slot sl], V(t-1) [V slot sl + 2], V(t) [V slot sl]; stages K(t+2) -> K slot
    // sl + 2 and V(t+1) -> V slot sl + 1.  The fragment window enters phase j of step sl at offset (4 sl + j) % 3 = (sl + j) % 3.
    auto step = [&](const int sl) __attribute__((always_inline)) {
      if constexpr (!(DIAG & 8)) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's DMA pieces of K(t+1) / V(t) have landed ...
        __syncthreads();                                    // ... and everyone's; every wave finished step t - 1
      }
      const int sl1 = sl == 2 ? 0 : sl + 1, sl2 = sl == 0 ? 2 : sl - 1;
      char* kdst = kring + sl2 * K_TILE_BYTES + wid * 4096;
      char* vdst = vring + sl1 * V_TILE_BYTES + wid * 4096;
      if (wave_on) {
        ragged = d0_valid < KT;
        mask_x = (float)(d0_valid - 8 * fh);
        phase(I0{}, I0{}, sl, sl2, I1{}, I16{}, I1{}, kdst, vdst);     // S0(t)    | softmax 1 (t-1), slots 16-31   (next: PV1(t-1) from V slot sl2)
        phase(I1{}, I1{}, sl2, sl, I0{}, I0{}, I0{}, kdst, vdst);      // PV1(t-1) | softmax 0 (t),   slots 0-15    (next: S1(t)    from K slot sl)
        phase(I0{}, I1{}, sl, sl, I0{}, I16{}, I0{}, kdst, vdst);      // S1(t)    | softmax 0 (t),   slots 16-31   (next: PV0(t)   from V slot sl)
        phase(I1{}, I0{}, sl, sl1, I1{}, I0{}, I0{}, kdst, vdst);      // PV0(t)   | softmax 1 (t),   slots 0-15    (next: S0(t+1)  from K slot sl1)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the look-ahead reads of the next step's S0: back in hipcc's books
        keep_window();
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) dma_k(i, kdst);
#pragma unroll
        for (int i = 0; i < 4; ++i) dma_v(i, vdst);
      }
      d0_valid = d1.nk - d1.k0; d1 = d2; v_offsets(); advance();
    };
    // the first two fragments of S0(0): K(0) has to be visible to every wave first
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    sfor<0, FW - 1>([&](auto jc) __attribute__((always_inline)) { fq[decltype(jc)::value] = frag(I0{}, 0, jc); });
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    int t = 0;
#pragma unroll 1
    for (; t + 2 < total_tiles; t += 3) { step(0); step(1); step(2); }
    const int rem = total_tiles - t;
    if (rem >= 1) step(0);
    if (rem >= 2) step(1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no LDS-DMA may be in flight when the workgroup ends
    // ---- drain: softmax 1 (T-1) slots 16-31, then PV1(T-1) from V slot (T-1) % 3
    if (wave_on) {
      sfor<16, 32>([&](auto ic) __attribute__((always_inline)) { soft_slot(I1{}, ic); });
      const int last = (total_tiles - 1) % 3;
      auto drain = [&](const int vsl) __attribute__((always_inline)) {
        sfor<0, FW - 1>([&](auto jc) __attribute__((always_inline)) { fq[decltype(jc)::value] = frag(I1{}, vsl, jc); });
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        phase(I1{}, I1{}, vsl, 0, IN{}, I0{}, I0{}, kring, vring);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        keep_window();
      };
      if (last == 0) drain(0); else if (last == 1) drain(1); else drain(2);
    }
  }
  // ---- range report: one word per workgroup, always written (launch_attention_bf16 runs attn_kernel for the flagged ones)
  if (lane == 0) wflags[wid] = ovf;
  __syncthreads();
  if (tid == 0 && p.redo) p.redo[((long)row * gridDim.y + head) * p.redo_nb + blockIdx.x] = wflags[0] | wflags[1] | wflags[2] | wflags[3];
  if (!wave_on) return;

  // ---- epilogue: lane holds O[q = 32 qb + fr][32 d + 8 g + 4 fh + 0..3]; gate values requested as one batch per query stream
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    if (!q_ok[qb]) continue;
    const int q = qi[qb];
    uint2 gq[4][4];
    const bf16_t* gp = p.G ? p.G + (long)row * p.g_row_stride + (long)q * p.g_ld + head * HD + 4 * fh : nullptr;
    if (gp) {
#pragma unroll
      for (int d = 0; d < 4; ++d)
#pragma unroll
        for (int g = 0; g < 4; ++g) gq[d][g] = *(const uint2*)(gp + 32 * d + 8 * g);
    }
    const float inv_l = 1.0f / st[qb].l;
    bf16_t* op = p.O + (long)row * p.o_row_stride + (long)q * p.o_ld + head * HD + 4 * fh;
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float y[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) y[i] = bf2f(f2bf(o[qb][d][4 * g + i] * inv_l));
        if (gp) {
          float gv[4];
          Vec4<bf16_t>::unpack(gq[d][g], gv);
#pragma unroll
          for (int i = 0; i < 4; ++i) y[i] = y[i] * (p.g_act ? gv[i] : bf2f(f2bf(sigmoid_fast(gv[i]))));      // wave-uniform
        }
        *(uint2*)(op + 32 * d + 8 * g) = Vec4<bf16_t>::pack(y);
      }
  }
}

// =====================================================================================================================================
// attn5_kernel: 4 waves x 64 queries like attn4_kernel, but the two 32-query streams of a wave run in LOCKSTEP and share every LDS
// fragment: one ds_read_b128 feeds two MFMAs (stream 0 and stream 1).  attn4_kernel (and attn_kernel) read one 1 KiB fragment per
// MFMA - and a ds_read_b128 with four waves on the LDS port costs its wave ~19 issue cycles (tools/micro/mfma_fill.hip): attn4_kernel's
// "no MFMA, no softmax" ablation build still takes 120-147 us of the 250.  Here it is one read per MFMA pair.
//   step t:   A(t+1): S0,S1(t+1) = K(t+1) Q0^T, K(t+1) Q1^T   (16 K fragments, 32 MFMAs)  | late(t), both streams
//             B(t):   O0,O1 += V^T(t) P0,P1(t)                 (16 V fragments, 32 MFMAs)  | early(t+1)
// Scores are double-buffered in arch VGPRs (asm MFMAs in the VGPR form: 2 x 64 registers), P is packed into 32 more; O (128), Q (64)
// and a 16-fragment window of K / V^T fragments (64, written by ds_read directly) fill the accumulator file.
// Softmax WITHOUT a reference point: Q is pre-multiplied by scale * log2(e) (bf16), so the MFMA result is the exp2 argument and
// P = exp2(S) - no row maximum, no subtraction, no rescale: per element one v_exp_f32, one add (row sum) and half a v_cvt_pk.  fp32 and
// bf16 share the exponent range, so this is exact as long as the row sum l stays inside [2^-64, 2^64] (RMS-normalised q, k: |S| < ~20);
// a workgroup whose l leaves that range (or is not finite) reports it in `redo` and attn_kernel redoes it with the online softmax.
//   early(t): exp2 + row sum of the kb = 0 half of tile t (rides B(t-1));  late(t): the kb = 1 half, the packing of P, l (rides A(t+1)).
// Rings: K four tiles, V^T three: step t stages K(t+4) and V^T(t+2), i.e. every tile two steps before its first reader (the look-ahead
// reads at the end of step t already touch K(t+2)), so the per-step wait is vmcnt(8) - the previous step's eight pieces stay in flight.
// Ring slots are not literals (the loop is unrolled by two, for the score buffers): the eight K and four V^T read addresses move by
// one v_add each per step.  No LGKM drain at a step's end: the eight look-ahead reads stay in flight across the barrier.
// The last step computes a dummy A(T) on whatever the K ring holds (a separate last-step path made hipcc spill): 32 MFMAs per workgroup.
// This wave's eight LDS-DMA pieces of a step go out in phase B (phase A carries the packing of P).  Workgroup shapes: 256 queries (two
// streams per wave), or 128 queries with one stream per wave (NS = 1: a short last block, and every block of a grid of fewer than 256
// workgroups, AttnArgs.q128).  The per-segment operands live in an LDS table (segtab) that only the walk's rare paths read.
constexpr int K5_SLOTS = 4, V5_SLOTS = 3;
constexpr int SMEM5 = K5_SLOTS * K_TILE_BYTES + V5_SLOTS * V_TILE_BYTES + 16 + 4 * 32;      // rings | 4 range flags | segment table

__device__ __forceinline__ bf16x8 lds_read16a(unsigned addr, int off) {      // LDS -> accumulator file
  bf16x8 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=a"(v) : "v"(addr), "i"(off));
  return v;
}
__device__ __forceinline__ void mfma_sa_first(f32x16& s, const bf16x8& kf, const bf16x8& q) {
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(s) : "a"(kf), "a"(q));
}
__device__ __forceinline__ void mfma_sa_next(f32x16& s, const bf16x8& kf, const bf16x8& q) {
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(s) : "a"(kf), "a"(q));
}
__device__ __forceinline__ void mfma_pv(f32x16& o, const bf16x8& vf, const bf16x8& pf) {
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(o) : "a"(vf), "v"(pf));
}

// eight fp32 -> bf16x8 with exactly four v_cvt_pk_bf16_f32 (hipcc sometimes converts the elements one by one and merges with v_perm_b32)
__device__ __forceinline__ bf16x8 pack8_asm(const f32x16& s, int base) {
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  u32x4 r;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    unsigned u;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u) : "v"(s[base + 2 * j]), "v"(s[base + 2 * j + 1]));
    r[j] = u;
  }
  return __builtin_bit_cast(bf16x8, r);
}

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
struct Soft5 {           // softmax state of one query stream (the scores live in sc[buf][q][kb])
  u32x4 pf[4];           // P^T (bf16 pairs) as the B operand of the four PV k-steps
  float l, rs;           // denominator; row sum of the tile in flight
};

// DIAG bits (timing experiments): 1 = no softmax VALU work, 2 = no MFMAs, 8 = linear (trivially conflict-free) fragment addresses (all three:
// wrong results), 16 = no LDS-DMA in the loop, 32 = no LDS reads, 4 = s_memtime stamps per phase into p.prof
// NS = query streams per wave: 2 (256 queries per workgroup), or 1 for a last block of at most 128 queries (four waves x 32 queries, one MFMA per
// fragment: no wave idles; the same code with the second stream compiled out)
template <int DIAG, int NS>
__device__ __forceinline__ void attn5_body(const AttnArgs& p) {
  unsigned long long pt[6] = {0, 0, 0, 0, 0, 0}, k_t0 = 0, k_r0 = 0, t_end = 0;
  if constexpr (DIAG & 4) k_t0 = __builtin_amdgcn_s_memtime();
  extern __shared__ __attribute__((aligned(16))) char smem[];   // K ring [4][16 KiB] | V ring [3][16 KiB] | 4 range flags
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int row = blockIdx.z, head = blockIdx.y;
  const int qbase = blockIdx.x * (p.q128 ? 128 : 256);
  const int fr = lane & 31, fh = lane >> 5;
  constexpr int WQ = 32 * NS;                            // queries per wave
  const bool wave_on = qbase + wid * WQ < p.S;          // wave-uniform: a wave without queries only stages tiles
  char* const kring = smem;
  char* const vring = smem + K5_SLOTS * K_TILE_BYTES;
  int* const wflags = (int*)(smem + K5_SLOTS * K_TILE_BYTES + V5_SLOTS * V_TILE_BYTES);

  bf16x8 qf[2][8];
#pragma unroll
  for (int qb = 0; qb < NS; ++qb) {
    const int qi = qbase + wid * WQ + qb * 32 + fr;
    const int qc = qi < p.S ? qi : p.S - 1;
    const bf16_t* qp = p.Q + (long)row * p.q_row_stride + (long)qc * p.q_ld + head * HD + 8 * fh;
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) qf[qb][kk] = *(const bf16x8*)(qp + 16 * kk);      // requested here, scaled behind the first tiles' DMA issue
  }
  auto scale_q = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int qb = 0; qb < NS; ++qb)
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) {
        f32x8 v = __builtin_convertvector(__builtin_bit_cast(hbf16x8, qf[qb][kk]), f32x8);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] *= p.scale * 1.4426950408889634f;
        qf[qb][kk] = __builtin_bit_cast(bf16x8, __builtin_convertvector(v, hbf16x8));
        asm volatile("" : "+a"(qf[qb][kk]));          // from here on an accumulator-file value: no per-use copies
      }
  };

  int nkraw[4];
#pragma unroll
  for (int sgi = 0; sgi < 4; ++sgi) nkraw[sgi] = (sgi < p.nseg ? p.seg[sgi].nkeys : p.seg[0].nkeys)[row];
  auto seg_keys = [&](int sgi) __attribute__((always_inline)) -> int { const int nk = sgi < p.nseg ? nkraw[sgi] : 0; return nk < 0 ? 0 : nk; };
  const int nk0 = seg_keys(0), nk1 = seg_keys(1), nk2 = seg_keys(2), nk3 = seg_keys(3);
  auto NK = [&](int s) __attribute__((always_inline)) -> int { return s == 0 ? nk0 : s == 1 ? nk1 : s == 2 ? nk2 : s == 3 ? nk3 : 0; };
  auto NT = [&](int s) __attribute__((always_inline)) -> int { return (NK(s) + KT - 1) / KT; };
  const int total_tiles = NT(0) + NT(1) + NT(2) + NT(3);
  // per-segment operands: resolved once, parked in LDS (segtab) and fetched by the rare paths of the tile walk (a segment switch, a ragged
  // tile) - kept in scalar registers they (24 of ~100) pushed the walk's own state into VGPR lanes (95 v_readlane in the kernel)
  struct SegEnt { const char* kb; const char* vb; int kld, vld, pad0, pad1; };      // 32 bytes
  SegEnt* const segtab = (SegEnt*)(smem + K5_SLOTS * K_TILE_BYTES + V5_SLOTS * V_TILE_BYTES + 16);
  int mod0 = 0;
#pragma unroll
  for (int sgi = 0; sgi < 4; ++sgi)
    if (sgi < p.nseg && mod0 == 0 && p.seg[sgi].kv_mod > 0) mod0 = p.seg[sgi].kv_mod;
  const int rmod0 = mod0 > 0 ? row % mod0 : row;
#pragma unroll
  for (int sgi = 0; sgi < 4; ++sgi) {
    if (sgi < p.nseg && tid == 0) {
      const AttnSeg& sg = p.seg[sgi];
      const int kvrow = sg.kv_mod == 0 ? row : (sg.kv_mod == mod0 ? rmod0 : row % sg.kv_mod);
      segtab[sgi].kb = (const char*)(sg.K + (long)kvrow * sg.k_row_stride + (long)head * sg.k_head_stride);
      segtab[sgi].vb = (const char*)(sg.Vt + (long)kvrow * sg.vt_row_stride + (long)head * sg.vt_head_stride);
      segtab[sgi].kld = (int)sg.k_ld * 2; segtab[sgi].vld = (int)sg.vt_ld * 2;
    }
  }
  __syncthreads();
  // one table entry into scalar registers (all LGKM traffic drained: the rare paths only)
  typedef unsigned u32x4t __attribute__((ext_vector_type(4)));
  struct SegVal { const char* kb; const char* vb; int kld, vld; };
  auto seg_fetch = [&](int sg) __attribute__((always_inline)) -> SegVal {
    const unsigned addr = (unsigned)(unsigned long)(lptr_t)segtab + (unsigned)sg * 32u;
    u32x4t e0, e1;
    asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:16\n\ts_waitcnt lgkmcnt(0)" : "=&v"(e0), "=&v"(e1) : "v"(addr) : "memory");
    SegVal r;
    r.kb = (const char*)((unsigned long)(unsigned)__builtin_amdgcn_readfirstlane((int)e0[0]) | ((unsigned long)(unsigned)__builtin_amdgcn_readfirstlane((int)e0[1]) << 32));
    r.vb = (const char*)((unsigned long)(unsigned)__builtin_amdgcn_readfirstlane((int)e0[2]) | ((unsigned long)(unsigned)__builtin_amdgcn_readfirstlane((int)e0[3]) << 32));
    r.kld = __builtin_amdgcn_readfirstlane((int)e1[0]);
    r.vld = __builtin_amdgcn_readfirstlane((int)e1[1]);
    return r;
  };
  auto next_seg = [&](int sg) __attribute__((always_inline)) -> int { ++sg; while (sg < 4 && NK(sg) == 0) ++sg; return sg; };

  // ---- the tile walk: w = tile t + 4 (its K is staged during step t), x3 = tile t + 3, x2 = tile t + 2 (its V^T is staged during step
  // t), valid1 = key count of tile t + 1 (masks the first softmax half).  Only w is advanced; the others are last step's w, x3, x2.  DMA roles as in attn4_kernel;
  // the per-lane role constants are recomputed where an offset changes (segment switch, ragged tile) instead of being kept in registers.
  struct Tile { int seg, k0, nk; const char* kptr; const char* vptr; int kstep; };      // kstep = the segment's K row pitch x KT
  struct Lag { int vld, valid; const char* vptr; };      // what the two followers need of a tile (vld: the segment's V^T row pitch in bytes)
  Tile w;
  int w_vld = 0;
  Lag x3, x2;
  int valid1 = 0;
  auto lag_of = [&]() __attribute__((always_inline)) -> Lag { return Lag{w_vld, w.nk - w.k0, w.vptr}; };
  unsigned k_off[4], v_off[4];
  auto k_offsets = [&]() __attribute__((always_inline)) {
    const int kld = w.kstep / KT, last = w.nk - 1 - w.k0;        // rows past the segment's last key repeat it (masked in the softmax)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = (wid * 4 + i) * 4 + (lane >> 4);
      k_off[i] = (unsigned)(min(r, last) * kld + (((lane & 15) ^ (r & 15)) << 4));
    }
  };
  auto enter = [&](int sg) __attribute__((always_inline)) {
    const SegVal e = seg_fetch(sg);
    w.seg = sg; w.k0 = 0; w.nk = NK(sg); w.kptr = e.kb; w.vptr = e.vb; w.kstep = e.kld * KT; w_vld = e.vld;
    k_offsets();
  };
  auto advance = [&]() __attribute__((always_inline)) {                         // past the end the walk stays on the last tile
    if (__builtin_expect(w.k0 + 2 * KT <= w.nk, 1)) { w.k0 += KT; w.kptr += w.kstep; w.vptr += KT * 2; }
    else if (w.k0 + KT < w.nk) { w.k0 += KT; w.kptr += w.kstep; w.vptr += KT * 2; k_offsets(); }
    else { const int sg = next_seg(w.seg); if (sg < 4) enter(sg); }
  };
  auto v_offsets = [&]() __attribute__((always_inline)) {                       // per-lane offsets of the V^T pieces of x2's segment: four v_mad, no branch
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int d = (wid * 4 + i) * 8 + (lane >> 3);
      v_off[i] = (unsigned)(d * x2.vld + (((lane & 7) ^ ((d >> 1) & 7)) << 4));
    }
  };
  auto dma_k = [&](int i, char* kdst) __attribute__((always_inline)) { glds16(w.kptr + k_off[i], kdst + i * 1024); };
  auto dma_v = [&](int i, char* vdst) __attribute__((always_inline)) { glds16(x2.vptr + v_off[i], vdst + i * 1024); };

  const int pi_row = (fr & 0x13) | ((fr & 4) << 1) | ((fr & 8) >> 1);  // K rows are fed with bits 2,3 swapped (see attn_kernel)
  const int sw_v = (lane >> 1) & 7;
  const int sw_k = pi_row & 15;
  unsigned ka[8], va[4];     // fragment read addresses incl. the CURRENT K / V ring slot (moved by kdelta / vdelta once per step)
  {
    const unsigned kbase = (unsigned)(unsigned long)(lptr_t)kring, vbase = (unsigned)(unsigned long)(lptr_t)vring;
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) ka[kk] = (DIAG & 8) ? kbase + lane * 16 + kk * 1024 : kbase + pi_row * 256 + (((2 * kk + fh) ^ sw_k) << 4);
#pragma unroll
    for (int kss = 0; kss < 4; ++kss) va[kss] = (DIAG & 8) ? vbase + lane * 16 + kss * 1024 : vbase + fr * 128 + (((2 * kss + fh) ^ sw_v) << 4);
  }
  f32x16 o[2][4];            // accumulator file (asm "+a")
  f32x16 sc[2][2][2];        // [buffer][stream][kb] exp2 arguments / P of a tile: arch VGPRs (asm "=&v" / "+v")
  Soft5 st[2];
#pragma unroll
  for (int qb = 0; qb < NS; ++qb) {
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[qb][d][r] = 0.0f;
    st[qb].l = 0.0f; st[qb].rs = 0.0f;
#pragma unroll
    for (int i = 0; i < 4; ++i) st[qb].pf[i] = u32x4{0, 0, 0, 0};
  }
  bool ragged = false;     // the tile whose early half runs next has fewer than KT valid keys (wave-uniform)
  int mask_l = 0;          // its valid keys - 8 fh

  // keys past a segment's end (the last tile of a segment only): exp2 argument -1e30.  Register r of sc[..][kb] = key 32 kb + 16 (r >> 3)
  // + 8 fh + (r & 7).  Runs in front of early(): directly behind the tile's last score MFMA, hence the wait states.
  auto mask_tile = [&](auto bc_) __attribute__((always_inline)) {
    constexpr int b = decltype(bc_)::value;
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    const float gone = -1e30f;
    sfor<0, 32 * NS>([&](auto ec) __attribute__((always_inline)) {
      constexpr int e = decltype(ec)::value, q = e >> 5, kb = (e >> 4) & 1, r = e & 15;
      float x = sc[b][q][kb][r];
      asm volatile("v_cmp_gt_i32 vcc, %1, %2\n\tv_cndmask_b32 %0, %3, %0, vcc" : "+v"(x) : "v"(mask_l), "n"(32 * kb + 16 * (r >> 3) + (r & 7)), "v"(gone) : "vcc");
      sc[b][q][kb][r] = x;
    });
  };
  // early(): element g of the kb = 0 half: P = exp2(S), row sum.  late(): element g of the kb = 1 half, one v_cvt_pk of finished pairs
  // (kb 0 pairs in gaps 0-7, kb 1 pair g - 8 in gaps 8-15), and the denominator in gap 15.
  // The pieces are placed by hand (tools/micro/mfma_fill.hip: one wave issues in order - MFMA 8 cycles, v_exp_f32 10, v_add / v_cvt_pk
  // 4.5, ds_read_b128 + counted wait 19 with four waves reading, an LDS-DMA piece ~24 - and an MFMA takes 32: what sits between two
  // MFMAs must add up to <= 24): behind the first MFMA of a pair exp(stream 0), add(stream 0), exp(stream 1); behind the second
  // add(stream 1), the packing / DMA piece, and the next pair's fragment read.
  auto early_exp = [&](auto qc_, auto bc_, auto gc_) __attribute__((always_inline)) {
    constexpr int q = decltype(qc_)::value, b = decltype(bc_)::value, g = decltype(gc_)::value;
    sc[b][q][0][g] = __builtin_amdgcn_exp2f(sc[b][q][0][g]);
  };
  auto early_add = [&](auto qc_, auto bc_, auto gc_) __attribute__((always_inline)) {      // one element late: no wait state behind the v_exp
    constexpr int q = decltype(qc_)::value, b = decltype(bc_)::value, g = decltype(gc_)::value;
    if constexpr (g == 1) st[q].rs = sc[b][q][0][0];
    else if constexpr (g > 1) st[q].rs += sc[b][q][0][g - 1];
  };
  auto cvt_pair = [&](auto qc_, auto bc_, auto kbc_, auto jc_) __attribute__((always_inline)) {      // pair j (elements 2 j, 2 j + 1) of half kb
    constexpr int q = decltype(qc_)::value, b = decltype(bc_)::value, kb = decltype(kbc_)::value, j = decltype(jc_)::value;
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef __bf16 hbf16x2 __attribute__((ext_vector_type(2)));
    const f32x2 v = {sc[b][q][kb][2 * j], sc[b][q][kb][2 * j + 1]};
    st[q].pf[2 * kb + (j >> 2)][j & 3] = __builtin_bit_cast(unsigned, __builtin_convertvector(v, hbf16x2));   // one v_cvt_pk_bf16_f32
  };
  auto late_exp = [&](auto qc_, auto bc_, auto gc_) __attribute__((always_inline)) {
    constexpr int q = decltype(qc_)::value, b = decltype(bc_)::value, g = decltype(gc_)::value;
    sc[b][q][1][g] = __builtin_amdgcn_exp2f(sc[b][q][1][g]);
  };
  auto late_add = [&](auto qc_, auto bc_, auto gc_) __attribute__((always_inline)) {
    constexpr int q = decltype(qc_)::value, b = decltype(bc_)::value, g = decltype(gc_)::value;
    if constexpr (g == 0) st[q].rs += sc[b][q][0][15]; else st[q].rs += sc[b][q][1][g - 1];
  };
  auto late_cvt = [&](auto qc_, auto bc_, auto gc_) __attribute__((always_inline)) {       // kb 0 pairs in gaps 0-7, kb 1 pair g - 8 in gaps 8-15
    constexpr int q = decltype(qc_)::value, b = decltype(bc_)::value, g = decltype(gc_)::value;
    if constexpr (g < 8) cvt_pair(qc_, bc_, std::integral_constant<int, 0>{}, gc_);
    else cvt_pair(qc_, bc_, std::integral_constant<int, 1>{}, std::integral_constant<int, (g >= 8 ? g - 8 : 0)>{});
    if constexpr (g == 15) {
      cvt_pair(qc_, bc_, std::integral_constant<int, 1>{}, std::integral_constant<int, 7>{});
      st[q].l += half_sum(st[q].rs + sc[b][q][1][15]);
    }
  };

  // ---- the 16-fragment window (accumulator file).  Fragment f of a phase sits in fw[f]; it is read LA gaps ahead: fragments LA..15 of a
  // phase in its own gaps 0..15-LA, fragments 0..LA-1 of the NEXT phase in gaps 16-LA..15 (slot f was consumed 16 - LA gaps earlier).
  bf16x8 fw[16];
  constexpr int LA = 8;      // look-ahead of the fragment reads, in fragments (= gap pairs); 12 makes hipcc spill Q fragments to scratch
  auto read_k = [&](auto jc) __attribute__((always_inline)) -> bf16x8 {           // K fragment j = (kb, kk) of the current K slot
    constexpr int j = decltype(jc)::value;
    if constexpr (DIAG & 32) { bf16x8 z; asm volatile("" : "=a"(z)); return z; }
    return lds_read16a(ka[j & 7], (j >> 3) * 8192);
  };
  auto read_v = [&](auto jc) __attribute__((always_inline)) -> bf16x8 {           // V^T fragment j = (kss, db) of the current V slot
    constexpr int j = decltype(jc)::value;
    if constexpr (DIAG & 32) { bf16x8 z; asm volatile("" : "=a"(z)); return z; }
    return lds_read16a(va[j >> 2], (j & 3) * 4096);
  };
  auto keep_window = [&]() __attribute__((always_inline)) {    // see attn4_kernel: a use of every in-flight read's destination behind the wait
    static_assert(LA == 8, "the look-ahead reads without a consumer are fragments 0 .. LA - 1 of a phase that does not run");
    asm volatile("" ::"a"(fw[0]), "a"(fw[1]), "a"(fw[2]), "a"(fw[3]), "a"(fw[4]), "a"(fw[5]), "a"(fw[6]), "a"(fw[7]));
  };
  typedef std::integral_constant<int, 0> I0;
  typedef std::integral_constant<int, 1> I1;
  int kdelta = K_TILE_BYTES, vdelta = V_TILE_BYTES;       // what moves ka / va to the ring slot of the next A / B phase

  // phase A: scores of one tile for both streams into buffer NB.  SOFT: second softmax half on buffer 1 - NB in the gaps.  NEXTA: the
  // next MFMA phase is again an A phase (prologue), else phase B.
  auto phase_a = [&](auto nbc, auto softc, auto nextac) __attribute__((always_inline)) {
    constexpr int NB = decltype(nbc)::value;
    constexpr bool SOFT = decltype(softc)::value != 0 && !(DIAG & 1), NEXTA = decltype(nextac)::value != 0;
    typedef std::integral_constant<int, 1 - NB> OB;
    sfor<0, 16>([&](auto fc) __attribute__((always_inline)) {
      constexpr int f = decltype(fc)::value, kb = f >> 3, kk = f & 7;
      constexpr int j = f + LA, nj = j >= 16 ? j - 16 : 0;
      if constexpr (j < 16) fw[j] = read_k(std::integral_constant<int, (j < 16 ? j : 0)>{});
      else if constexpr (NEXTA) {
        if constexpr (nj < 8) ka[nj] += kdelta;
        fw[nj] = read_k(std::integral_constant<int, nj>{});
      } else fw[nj] = read_v(std::integral_constant<int, nj>{});
      lds_wait<LA>();
      if constexpr (!(DIAG & 2)) {
        if constexpr (kk == 0) mfma_sa_first(sc[NB][0][kb], fw[f], qf[0][kk]); else mfma_sa_next(sc[NB][0][kb], fw[f], qf[0][kk]);
      }
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (SOFT) { late_exp(I0{}, OB{}, fc); late_add(I0{}, OB{}, fc); if constexpr (NS == 2) late_exp(I1{}, OB{}, fc); else late_cvt(I0{}, OB{}, fc); }
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (!(DIAG & 2) && NS == 2) {
        if constexpr (kk == 0) mfma_sa_first(sc[NB][1][kb], fw[f], qf[1][kk]); else mfma_sa_next(sc[NB][1][kb], fw[f], qf[1][kk]);
      } else if constexpr (DIAG & 2) asm volatile("" ::"a"(fw[f]));
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (SOFT && NS == 2) { late_add(I1{}, OB{}, fc); late_cvt(I0{}, OB{}, fc); late_cvt(I1{}, OB{}, fc); }
      if constexpr (!NEXTA && f >= 16 - LA && f < 24 - LA) ka[f - (16 - LA)] += kdelta;      // every read of this K slot is issued: on to the next one
      __builtin_amdgcn_sched_barrier(0);
    });
  };
  // phase B: O += V^T P for both streams; SOFT: first softmax half on buffer SB in the gaps.  The next phase is an A phase.
  auto phase_b = [&](auto sbc, auto softc, auto lastc, auto dmac, char* kdst, char* vdst) __attribute__((always_inline)) {
    constexpr int SB = decltype(sbc)::value;
    constexpr bool SOFT = decltype(softc)::value != 0 && !(DIAG & 1), LASTB = decltype(lastc)::value != 0, DMA = decltype(dmac)::value != 0;
    if constexpr (SOFT) { if (__builtin_expect(ragged, 0)) mask_tile(sbc); }
    sfor<0, 16>([&](auto fc) __attribute__((always_inline)) {
      constexpr int f = decltype(fc)::value, kss = f >> 2, db = f & 3;
      constexpr int j = f + LA, nj = j >= 16 ? j - 16 : 0;
      if constexpr (j < 16) fw[j] = read_v(std::integral_constant<int, (j < 16 ? j : 0)>{});
      else if constexpr (!LASTB) fw[nj] = read_k(std::integral_constant<int, nj>{});
      lds_wait<(LASTB && 15 - f < LA ? 15 - f : LA)>();
      if constexpr (!(DIAG & 2)) mfma_pv(o[0][db], fw[f], __builtin_bit_cast(bf16x8, st[0].pf[kss]));
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (SOFT) { early_exp(I0{}, sbc, fc); early_add(I0{}, sbc, fc); if constexpr (NS == 2) early_exp(I1{}, sbc, fc); }
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (!(DIAG & 2) && NS == 2) mfma_pv(o[1][db], fw[f], __builtin_bit_cast(bf16x8, st[1].pf[kss])); else if constexpr (DIAG & 2) asm volatile("" ::"a"(fw[f]));
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (SOFT && NS == 2) early_add(I1{}, sbc, fc);
      if constexpr (DMA && (f & 1) == 0 && !(DIAG & 16)) { if constexpr (f < 8) dma_k(f >> 1, kdst); else dma_v((f - 8) >> 1, vdst); }     // this wave's 4 K + 4 V^T pieces
      if constexpr (!LASTB && f >= 16 - LA && f < 20 - LA) va[f - (16 - LA)] += vdelta;        // every read of this V slot is issued: on to the next one
      __builtin_amdgcn_sched_barrier(0);
    });
  };

  // gate values of the epilogue; stream 0's are requested in the last step between A and B (both streams' at once spill)
  uint2 gq[2][4][4];
  auto load_gate = [&](auto qbc) __attribute__((always_inline)) {
    if (p.G) {
      {
        constexpr int qb = decltype(qbc)::value;
        const int qi = qbase + wid * WQ + qb * 32 + fr;
        const bf16_t* gp = p.G + (long)row * p.g_row_stride + (long)(qi < p.S ? qi : p.S - 1) * p.g_ld + head * HD + 4 * fh;
#pragma unroll
        for (int d = 0; d < 4; ++d)
#pragma unroll
          for (int g = 0; g < 4; ++g) gq[qb][d][g] = *(const uint2*)(gp + 32 * d + 8 * g);
      }
    }
  };
  unsigned long long pa = 0, pb = 0, pc2 = 0;
  if constexpr (DIAG & 4) pa = stamp() - k_t0;       // arguments, key counts, segment operands resolved; Q requested
  if (total_tiles > 0) {
    {
      int s0 = 0;
      while (s0 < 3 && NK(s0) == 0) ++s0;
      enter(s0);
    }
    x2 = lag_of(); v_offsets();
    const int valid0 = x2.valid;
#pragma unroll
    for (int i = 0; i < 4; ++i) dma_k(i, kring + wid * 4096);                          // K(0) -> K slot 0
#pragma unroll
    for (int i = 0; i < 4; ++i) dma_v(i, vring + wid * 4096);                          // V(0) -> V slot 0
    advance();
#pragma unroll
    for (int i = 0; i < 4; ++i) dma_k(i, kring + K_TILE_BYTES + wid * 4096);           // K(1) -> K slot 1 (past the end: the last tile again)
    x2 = lag_of(); v_offsets();
    valid1 = x2.valid;
#pragma unroll
    for (int i = 0; i < 4; ++i) dma_v(i, vring + V_TILE_BYTES + wid * 4096);           // V(1) -> V slot 1
    advance();
#pragma unroll
    for (int i = 0; i < 4; ++i) dma_k(i, kring + 2 * K_TILE_BYTES + wid * 4096);       // K(2) -> K slot 2
    x2 = lag_of(); v_offsets();                                                         // x2 = tile 2
    advance();
#pragma unroll
    for (int i = 0; i < 4; ++i) dma_k(i, kring + 3 * K_TILE_BYTES + wid * 4096);       // K(3) -> K slot 3
    x3 = lag_of();                                                                      // x3 = tile 3
    advance();                                                                          // w  = tile 4
    if constexpr (DIAG & 4) { pt[4] = 0; pb = stamp() - k_t0; }     // first tiles' DMA issued
    scale_q();
    if constexpr (DIAG & 4) pc2 = stamp() - k_t0;                   // Q arrived and scaled                        // the Q loads were issued before the DMA pieces: their wait leaves those in flight
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if constexpr (DIAG & 4) k_r0 = stamp() - k_t0;     // (reused) setup + first tiles landed
    // ---- prologue: A(0) alone into buffer 0 (reads ahead into A(1): K slot 1), then early(0) alone
    if (wave_on) {
      sfor<0, LA>([&](auto jc) __attribute__((always_inline)) { fw[decltype(jc)::value] = read_k(jc); });
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      kdelta = K_TILE_BYTES;
      phase_a(I0{}, I0{}, I1{});
      asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");          // MFMA results -> vector ALU: hipcc pads nothing behind an asm MFMA
      mask_l = valid0 - 8 * fh;
      if (valid0 < KT) mask_tile(I0{});
      if constexpr (!(DIAG & 1)) sfor<0, 16>([&](auto ic) __attribute__((always_inline)) { early_exp(I0{}, I0{}, ic); early_add(I0{}, I0{}, ic); if constexpr (NS == 2) { early_exp(I1{}, I0{}, ic); early_add(I1{}, I0{}, ic); } });
    }
    if constexpr (DIAG & 4) pt[5] = stamp() - k_t0;      // prologue: setup, first tiles landed, A(0), early(0)
    int t = 0;
    int kst = 0, vst = 0;    // t % 4, t % 3: K(t + 4) is staged into K slot kst, V^T(t + 2) into V slot (vst + 2) % 3
    // step t (P = t & 1): A(t+1) -> buffer 1 - P from the K slot ka points at | second half(t) on buffer P;  B(t) from the V slot va points at | first half(t+1)
    auto step = [&](auto pc) __attribute__((always_inline)) {
      constexpr int P = decltype(pc)::value;
      unsigned long long s0 = 0, s1 = 0, s2 = 0, s3 = 0;
      if constexpr ((DIAG & 4) && !(DIAG & 128)) { s0 = stamp(); if (t_end) pt[4] += s0 - t_end; }
      // this wave's pieces of the tiles staged two steps ago (K(t+2), V(t)) have landed - last step's eight stay in flight - ...
      // (past the last tile the walk stays on it: the same eight pieces per step, a branch around them cost 440 cycles per step)
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      __builtin_amdgcn_s_barrier();                       // ... and everyone's; every wave finished step t - 1
      asm volatile("" ::: "memory");
      if constexpr ((DIAG & 4) && !(DIAG & 128)) s1 = stamp();
      char* kdst = kring + kst * K_TILE_BYTES + wid * 4096;
      char* vdst = vring + (vst == 0 ? 2 : vst - 1) * V_TILE_BYTES + wid * 4096;
      if (__builtin_expect(wave_on, 1)) {
        ragged = valid1 < KT;
        mask_l = valid1 - 8 * fh;
        kdelta = kst == 2 ? -(K5_SLOTS - 1) * K_TILE_BYTES : K_TILE_BYTES;
        vdelta = vst == 2 ? -(V5_SLOTS - 1) * V_TILE_BYTES : V_TILE_BYTES;
        phase_a(std::integral_constant<int, 1 - P>{}, I1{}, I0{});
        // the last step: the epilogue's gate values are requested here (the registers of the P just packed are free), B(T-1) covers part of their latency
        if (__builtin_expect(t + 1 >= total_tiles, 0)) load_gate(I0{});
        if constexpr ((DIAG & 4) && !(DIAG & 128)) s2 = stamp();
        phase_b(std::integral_constant<int, 1 - P>{}, I1{}, I0{}, I1{}, kdst, vdst);
        if constexpr ((DIAG & 4) && !(DIAG & 128)) { s3 = stamp(); pt[0] += s1 - s0; pt[1] += s2 - s1; pt[2] += s3 - s2; pt[3] += 1; t_end = s3; }
        else if constexpr (DIAG & 4) pt[3] += 1;
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) dma_k(i, kdst);
#pragma unroll
        for (int i = 0; i < 4; ++i) dma_v(i, vdst);
      }
      valid1 = x2.valid; x2 = x3; x3 = lag_of(); v_offsets(); advance();
      kst = kst == 3 ? 0 : kst + 1;
      vst = vst == 2 ? 0 : vst + 1;
      ++t;
    };
#pragma unroll 1
    while (t + 1 < total_tiles) { step(I0{}); step(I1{}); }
    if (t < total_tiles) step(I0{});
    if constexpr (DIAG & 4) t_end = stamp() - k_t0;      // end of the tile loop
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");            // the look-ahead reads of a phase that does not run
    if (wave_on) keep_window();
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");            // the last PV MFMAs -> the epilogue's accumulator reads
  }
  // ---- range report: one word per workgroup, always written (launch_attention_bf16 runs attn_kernel for the flagged ones)
  int ovf = total_tiles > 0 ? 0 : 1;       // no key at all: left to attn_kernel
  if (wave_on && total_tiles > 0) {
    const bool bad0 = !(st[0].l > 5.4e-20f && st[0].l < 1.8e19f), on0 = qbase + wid * WQ + fr < p.S;     // 2^-64 .. 2^64, NaN is bad
    bool bad = bad0 && on0;
    if constexpr (NS == 2) bad = bad || (!(st[1].l > 5.4e-20f && st[1].l < 1.8e19f) && qbase + wid * WQ + 32 + fr < p.S);
    ovf = __any(bad) ? 1 : 0;
    if constexpr (DIAG & 3) ovf = 0;          // timing builds: no second pass
  }
  if (lane == 0) wflags[wid] = ovf;
  __syncthreads();
  if (tid == 0 && p.redo) p.redo[((long)row * gridDim.y + head) * p.redo_nb + blockIdx.x] = wflags[0] | wflags[1] | wflags[2] | wflags[3];
  if (!wave_on) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); return; }      // no LDS-DMA may be in flight when a wave ends
  if (total_tiles == 0) load_gate(I0{});
  if constexpr (NS == 2) load_gate(I1{});                 // stream 1's gate values: their latency is covered by stream 0's rows

  // ---- epilogue: lane holds O[q = 32 qb + fr][32 d + 8 g + 4 fh + 0..3]; gate values requested as one batch per query stream
#pragma unroll
  for (int qb = 0; qb < NS; ++qb) {
    const int q = qbase + wid * WQ + qb * 32 + fr;
    if (q >= p.S) continue;
    const bool gp = p.G != nullptr;
    // every LDS-DMA piece of the last steps (tiles past the end) has landed before the first store is issued - the gate values, requested
    // after them, are needed here anyway - so nothing is waited for at the kernel's end (a vmcnt(0) there also waits for the stores)
    if (qb == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const float inv_l = 1.0f / st[qb].l;
    bf16_t* op = p.O + (long)row * p.o_row_stride + (long)q * p.o_ld + head * HD + 4 * fh;
    // GM: 0 = no gate, 1 = raw gate (sigmoid here), 2 = the QKVG tail stored bf16(sigmoid(gate)) already (AttnArgs.g_act): one
    // wave-uniform choice per stream instead of a branch per four outputs
    auto rows_out = [&](auto gmc) __attribute__((always_inline)) {
      constexpr int GM = decltype(gmc)::value;
#pragma unroll
      for (int d = 0; d < 4; ++d)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          float y[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) y[i] = bf2f(f2bf(o[qb][d][4 * g + i] * inv_l));
          if constexpr (GM != 0) {
            float gv[4];
            Vec4<bf16_t>::unpack(gq[qb][d][g], gv);
#pragma unroll
            for (int i = 0; i < 4; ++i) y[i] = y[i] * (GM == 2 ? gv[i] : bf2f(f2bf(sigmoid_fast(gv[i]))));
          }
          if constexpr (DIAG & 256) store_o4_fp8(p.O8 + (long)row * p.o8_row_stride + (long)q * p.o8_ld + head * HD + 4 * fh + 32 * d + 8 * g, y, p.o8_inv);
          else *(uint2*)(op + 32 * d + 8 * g) = Vec4<bf16_t>::pack(y);
        }
    };
    if (!gp) rows_out(I0{});
    else if (p.g_act) rows_out(std::integral_constant<int, 2>{});
    else rows_out(I1{});
  }
  if constexpr (DIAG & 4) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) {
      unsigned long long* dst = (unsigned long long*)p.prof + ((((long)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 4 + wid) * 8;
      dst[0] = pt[0]; dst[1] = pt[1] | (pa << 40); dst[2] = pt[2] | (pb << 40); dst[3] = pt[3]; dst[4] = pt[4] | (pc2 << 40);
      dst[5] = __builtin_amdgcn_s_memtime() - k_t0; dst[6] = k_r0; dst[7] = pt[5] | (t_end << 32);
    }
  }
}

template <int DIAG>
__global__ void __launch_bounds__(256, 1) attn5_kernel(const AttnArgs p) {
  if (p.q128 || p.S - (int)blockIdx.x * 256 <= 128) attn5_body<DIAG, 1>(p);     // 128-query blocks (small grids), or the short last block
  else attn5_body<DIAG, 2>(p);
}

}  // namespace

hipError_t launch_attention_bf16(const AttnArgs& a, hipStream_t st) {
  if (a.nseg < 1 || a.nseg > 4 || a.S < 1 || a.H < 1 || a.rows < 1) return hipErrorInvalidValue;
  if (a.causal && a.nseg != 1) return hipErrorInvalidValue;
  if (a.O8 && (!(a.o8_inv > 0.0f) || (a.o8_ld & 3))) return hipErrorInvalidValue;
  for (int s = 0; s < a.nseg; ++s)
    if ((a.seg[s].vt_ld & 7) || (a.seg[s].k_ld & 7) || !a.seg[s].nkeys) return hipErrorInvalidValue;
  static std::atomic<unsigned long long> prepared[5];
  const void* ks[5] = {(const void*)attn_kernel<false, false, false>, (const void*)attn_kernel<true, false, false>,
                       (const void*)attn_kernel<false, true, false>, (const void*)attn_kernel<true, true, false>,
                       (const void*)attn_kernel<false, false, true>};
  for (int i = 0; i < 5; ++i)
    if (hipError_t e = ensure_dyn_lds(ks[i], SMEM, prepared[i]); e != hipSuccess) return e;
  dim3 grid((a.S + QT - 1) / QT, a.H, a.rows);
  bool bias = false;
  for (int s = 0; s < a.nseg; ++s) bias = bias || a.seg[s].bias != nullptr;
  // ECHO_ATTN=4 / 5 select attn4_kernel / attn5_kernel (4 waves x 64 queries, one wave per SIMD) for the joint attention, then attn_kernel
  // for the (normally zero) workgroups whose scores left the fast kernels' range; both launches on the caller's stream.
  // Default: attn5_kernel with 256-query workgroups (two query streams per wave), or 128-query ones for grids of at most one round (below).
  // Measured (tools/bench_attn4.py, us, against attn_kernel): 24 rows 214-221 vs 268, 12 rows 123 vs 138, 8 rows 105 vs 113, 4 rows 53 vs 75,
  // 3 rows 40.5 vs 41.7, 1 row 34.4 vs 36.8.  ECHO_ATTN=1 / 4 / 5 force a kernel, ECHO_ATTN_Q128=0 / 1 the block size.
  static const int forced = getenv("ECHO_ATTN") ? atoi(getenv("ECHO_ATTN")) : 0;
  static const int forced_q128 = getenv("ECHO_ATTN_Q128") ? atoi(getenv("ECHO_ATTN_Q128")) : -1;
  // 128-query workgroups (one stream per wave; 35 us each against 50) exactly when all of them run at once - one round of the chip.
  // Measured (us, 128- vs 256-query blocks): 1 row 35 / 50, 3 rows 41 / 52, 4 rows 72 / 53, 6 rows 69 / 71, 8 rows 112 / 105, 24 rows 252 / 217
  const long wg128 = (long)((a.S + 127) / 128) * a.H * a.rows;
  const bool q128 = forced_q128 >= 0 ? forced_q128 != 0 : wg128 <= 256;
  const int variant = forced && !(a.O8 && forced == 4) ? forced : 5;      // attn4_kernel has no e4m3 output
  if (!a.causal && !bias && (!a.prof || forced == 5) && (variant == 4 || variant == 5) && a.redo) {
    static std::atomic<unsigned long long> prep4[6];
    static const int diag = getenv("ECHO_ATTN_DIAG") ? atoi(getenv("ECHO_ATTN_DIAG")) : 0;      // timing experiments (tools/bench_attn4.py)
    // (a last block of at most 128 queries runs the kernel's one-stream body: S = 640 is two 256-query workgroups + one of 128)
    const bool q128m = variant == 5 && q128;
    const int nb256 = q128m ? (a.S + 127) / 128 : (a.S + 255) / 256;        // blocks of the fast kernel per (row, head)
    const int nfast = nb256;
    AttnArgs f = a;
    f.redo_nb = nb256;
    f.q_block0 = 0;
    f.q128 = q128m ? 1 : 0;
    const dim3 g4(nfast, a.H, a.rows);
    if (a.O8) {            // e4m3 output (fp8 engine, static activation scales): attn5_kernel<256>
      static std::atomic<unsigned long long> prep8{0};
      if (hipError_t e = ensure_dyn_lds((const void*)attn5_kernel<256>, SMEM5, prep8); e != hipSuccess) return e;
      hipLaunchKernelGGL(attn5_kernel<256>, g4, dim3(256), SMEM5, st, f);
    } else if (variant == 4) {
      if (hipError_t e = ensure_dyn_lds((const void*)attn4_kernel<0>, SMEM4, prep4[0]); e != hipSuccess) return e;
      hipLaunchKernelGGL(attn4_kernel<0>, g4, dim3(256), SMEM4, st, f);
    } else {
      const int di = a.prof ? (diag == 132 ? 2 : diag == 6 ? 3 : 5) : diag == 3 ? 4 : 1;
      const void* k5[6] = {nullptr, (const void*)attn5_kernel<0>, (const void*)attn5_kernel<132>, (const void*)attn5_kernel<6>, (const void*)attn5_kernel<3>,
                           (const void*)attn5_kernel<4>};
      if (hipError_t e = ensure_dyn_lds(k5[di], SMEM5, prep4[di]); e != hipSuccess) return e;
      switch (di) {
        case 2: hipLaunchKernelGGL(attn5_kernel<132>, g4, dim3(256), SMEM5, st, f); break;
        case 3: hipLaunchKernelGGL(attn5_kernel<6>, g4, dim3(256), SMEM5, st, f); break;
        case 4: hipLaunchKernelGGL(attn5_kernel<3>, g4, dim3(256), SMEM5, st, f); break;
        case 5: hipLaunchKernelGGL(attn5_kernel<4>, g4, dim3(256), SMEM5, st, f); break;
        default: hipLaunchKernelGGL(attn5_kernel<0>, g4, dim3(256), SMEM5, st, f);
      }
    }
    if (hipError_t e = hipGetLastError(); e != hipSuccess) return e;
    AttnArgs b = f;
    b.prof = nullptr;
    if (nfast < nb256) {                         // the short last block: attn_kernel, its blocks 2 nfast ..
      b.q_block0 = 2 * nfast;
      hipLaunchKernelGGL((attn_kernel<false, false, false>), dim3(grid.x - 2 * nfast, a.H, a.rows), dim3(512), SMEM, st, b);
      if (hipError_t e = hipGetLastError(); e != hipSuccess) return e;
    }
    b.q_block0 = 0;                              // second pass over the fast kernels' blocks: only the flagged ones do anything
    b.redo_filter = a.redo;
    static const bool redo_full = getenv("ECHO_ATTN_REDO_FULL") && atoi(getenv("ECHO_ATTN_REDO_FULL")) != 0;
    if (redo_full) {
      hipLaunchKernelGGL((attn_kernel<false, false, false>), dim3(min((int)grid.x, q128m ? nfast : 2 * nfast), a.H, a.rows), dim3(512), SMEM, st, b);
    } else {
      static std::atomic<unsigned long long> prep_redo{0};
      if (hipError_t e = ensure_dyn_lds((const void*)attn_redo_kernel, SMEM, prep_redo); e != hipSuccess) return e;
      const int nwords = a.rows * a.H * nb256;
      const int nwg = nwords < 32 ? nwords : 32;
      hipLaunchKernelGGL(attn_redo_kernel, dim3(nwg), dim3(512), SMEM, st, b, nwords, (nwords + nwg - 1) / nwg);
    }
    return hipGetLastError();
  }
  AttnArgs a0 = a;
  a0.q_block0 = 0; a0.redo_nb = 0; a0.redo_filter = nullptr; a0.q128 = 0;
  if (a.prof && !a.causal && !bias) hipLaunchKernelGGL((attn_kernel<false, false, true>), grid, dim3(512), SMEM, st, a0);   // s_memtime stamps
  else if (a.causal && bias) hipLaunchKernelGGL((attn_kernel<true, true, false>), grid, dim3(512), SMEM, st, a0);
  else if (a.causal) hipLaunchKernelGGL((attn_kernel<true, false, false>), grid, dim3(512), SMEM, st, a0);
  else if (bias) hipLaunchKernelGGL((attn_kernel<false, true, false>), grid, dim3(512), SMEM, st, a0);
  else hipLaunchKernelGGL((attn_kernel<false, false, false>), grid, dim3(512), SMEM, st, a0);
  return hipGetLastError();
}
