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
template <bool CAUSAL, bool BIAS, bool PROF>
__global__ void __launch_bounds__(512, 2) attn_kernel(const AttnArgs p) {
  unsigned long long pt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long k_t0 = 0, k_r0 = 0, pa = 0, pb = 0;
  if (PROF) { k_t0 = __builtin_amdgcn_s_memtime(); k_r0 = __builtin_amdgcn_s_memrealtime(); }
  extern __shared__ __attribute__((aligned(16))) char smem[];   // K ring [2][16 KiB] | V ring [2][16 KiB]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int qb = wid & 3, kh = wid >> 2;
  const int row = blockIdx.z, head = blockIdx.y;
  const int qbase = blockIdx.x * QT;
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
    unsigned long long* dst = (unsigned long long*)p.prof + ((((long)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 8 + wid) * 8;
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
        for (int i = 0; i < 4; ++i) y[i] = y[i] * bf2f(f2bf(sigmoid_fast(gv[i])));
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
  static std::atomic<unsigned long long> prepared[5];
  const void* ks[5] = {(const void*)attn_kernel<false, false, false>, (const void*)attn_kernel<true, false, false>,
                       (const void*)attn_kernel<false, true, false>, (const void*)attn_kernel<true, true, false>,
                       (const void*)attn_kernel<false, false, true>};
  for (int i = 0; i < 5; ++i)
    if (hipError_t e = ensure_dyn_lds(ks[i], SMEM, prepared[i]); e != hipSuccess) return e;
  dim3 grid((a.S + QT - 1) / QT, a.H, a.rows);
  bool bias = false;
  for (int s = 0; s < a.nseg; ++s) bias = bias || a.seg[s].bias != nullptr;
  if (a.prof && !a.causal && !bias) hipLaunchKernelGGL((attn_kernel<false, false, true>), grid, dim3(512), SMEM, st, a);   // s_memtime stamps
  else if (a.causal && bias) hipLaunchKernelGGL((attn_kernel<true, true, false>), grid, dim3(512), SMEM, st, a);
  else if (a.causal) hipLaunchKernelGGL((attn_kernel<true, false, false>), grid, dim3(512), SMEM, st, a);
  else if (bias) hipLaunchKernelGGL((attn_kernel<false, true, false>), grid, dim3(512), SMEM, st, a);
  else hipLaunchKernelGGL((attn_kernel<false, false, false>), grid, dim3(512), SMEM, st, a);
  return hipGetLastError();
}
