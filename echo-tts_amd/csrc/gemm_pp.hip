// gemm_pp_kernel (plan cfg 5): bf16 / fp8-e4m3, persistent 256x256 "ping-pong" kernel - every large EchoDiT linear (QKVG with its fused
// head-norm / RoPE / V^T tail, wo, SwiGLU w1||w3, w2).  Split from gemm.hip so that the two translation units compile in parallel.
#include "gemm_common.h"

namespace {

// ---------------------------------------------------------------------------------------------------------------
// gemm_pp_kernel: bf16, 256x256 output tile, K-tile 64, 8 waves (4 M x 2 N, 64 x 128 outputs each) on
// v_mfma_f32_16x16x32_bf16, written for one workgroup per CU (160 KiB LDS, <= 256 VGPRs).
//
// The two waves of every SIMD (wave w and w + 4) run the same program HALF A PHASE APART ("ping-pong"): while one is
// inside a cluster of 16 MFMAs, its partner issues the LDS fragment reads and the LDS-DMA of its next phase, and they
// swap roles at every workgroup barrier, so the matrix pipe of the SIMD always has one wave feeding it
// (MI355X_MICROARCH.md "Two waves per SIMD"; cdna_hip_programming.md §5 "256² 8-phase").
//
// A K-tile is 4 phases; phase q computes one 32 x 64 quadrant of the wave's outputs over K = 64:
//     q0 (m-half 0, n-half 0)   q1 (0, 1)   q2 (1, 1)   q3 (1, 0)
// and reads only the fragments it does not hold yet: q0 A(m-half 0) + W(n-half 0), q1 W(n-half 1), q2 A(m-half 1).
// The LDS image of a K-tile is cut the same way into four 16 KiB UNITS, staged one per phase in the order they are
// first read:  j0 = A rows of m-half 0 (all four wave rows),  j1 = W rows of n-half 0 (both wave columns),
// j2 = W n-half 1,  j3 = A m-half 1.  Unit u = 4t + j is issued in phase u - LEAD (2 LDS-DMA instructions per wave)
// into a ring of two K-tiles.  Ordering, with G0 = waves 0-3 and G1 = waves 4-7 one barrier behind:
//   RAW  every wave waits for its own DMA of the units first read in phase r with a counted vmcnt in the load part of
//        phase r - 1 (at most 2 (LEAD - 2) younger DMAs stay in flight), then passes the barrier that ends that
//        part; both groups have done so before either reads (G0 reads two barriers, G1 one barrier later);
//   WAR  unit u overwrites unit u - 8, last read in phase u - 8 - {0,1,1,1}[j]; those reads are retired by the
//        lgkmcnt(0) in front of that phase's first barrier, so restaging two phases later is safe: LEAD <= 6.
// Never vmcnt(0) in the loop: the look-ahead units of the last phases re-stage the final K-tile into dead ring slots.
//
// The kernel is PERSISTENT: gridDim.x workgroups (one per CU) walk the tile list with stride gridDim.x, and the unit
// stream simply continues from the last K-tile of one output tile into the first K-tile of the next, so the LDS-DMA of
// the next tile is in flight while the accumulators of the current one go through the epilogue.
//
// Epilogue (pp_epilogue): no workgroup barriers.  Tails that work per output column or per (row, head) run on the
// accumulator layout in registers (SwiGLU pairs, q/k head RMSNorm + RoPE: a wave owns 128 columns = one head); then
// every wave turns its own 16-row x 64-column pieces through a PRIVATE 4 KiB LDS area (behind the ring) into rows of
// 8 consecutive columns per lane, applies the row-layout tail (bias / activation / column scale / residual / Snake) and
// stores 16 bytes per lane: each store instruction writes eight whole 128-byte lines.  The V section of the fused QKV
// projection takes the same route with rows and columns exchanged (Vᵀ is written 8 tokens per lane).
#ifndef PP_LEAD
#define PP_LEAD 5
#endif
#ifndef PP_GN
#define PP_GN 4
#endif
#ifndef PP_RES_AHEAD
#define PP_RES_AHEAD 1
#endif
template <int V> struct IC { static constexpr int value = V; };

// lane id of this thread inside its wave, from the hardware (two VALU operations, no register carried from kernel entry)
// (the mask goes through an opaque asm: mbcnt is a pure function of constants to hipcc, which otherwise computes it once at kernel entry, keeps -
// i.e. spills - the result and reloads it at every use behind a vmcnt(0))
__device__ __forceinline__ int hw_lane() {
  unsigned m = ~0u;
  asm volatile("" : "+s"(m));
  return (int)__builtin_amdgcn_mbcnt_hi(m, __builtin_amdgcn_mbcnt_lo(m, 0u));
}


__device__ __forceinline__ uint4 pack8_bf16(const float* y) {
  uint4 r;
  r.x = pack_bf16x2(y[0], y[1]); r.y = pack_bf16x2(y[2], y[3]); r.z = pack_bf16x2(y[4], y[5]); r.w = pack_bf16x2(y[6], y[7]);
  return r;
}
__device__ __forceinline__ void unpack8_bf16(uint4 r, float* f) {
  f[0] = __uint_as_float(r.x << 16); f[1] = __uint_as_float(r.x & 0xffff0000u);
  f[2] = __uint_as_float(r.y << 16); f[3] = __uint_as_float(r.y & 0xffff0000u);
  f[4] = __uint_as_float(r.z << 16); f[5] = __uint_as_float(r.z & 0xffff0000u);
  f[6] = __uint_as_float(r.w << 16); f[7] = __uint_as_float(r.w & 0xffff0000u);
}

// global stores the compiler's wait-count pass does not see (edge tiles of the branch-free tails, see gemm_pp_kernel's fast tail): the hardware
// vmcnt then runs ahead of the compiler's count, so its counted waits wait for at least what they meant to.  s_nop: the store-data hazard slots
// the compiler's hazard recogniser would have kept (2 wait states on gfx940+ for a store of more than 64 bits, none needed for 16 bits).
__device__ __forceinline__ void hidden_store16(void* dst, uint4 pk) {
  const i32x4 pv = {(int)pk.x, (int)pk.y, (int)pk.z, (int)pk.w};
  asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" :: "v"(dst), "v"(pv) : "memory");      // 2 wait states before a VALU may overwrite the data of a > 64-bit store (gfx940+)
}
__device__ __forceinline__ void hidden_store2(void* dst, bf16_t v) {
  const int iv = v;
  asm volatile("global_store_short %0, %1, off\n\ts_nop 0" :: "v"(dst), "v"(iv) : "memory");
}

// Start of one epilogue FORM (a branch-free copy of a tail for one case): the lane coordinates are re-derived behind an asm that is different
// in every form, so that nothing computed from them is "the same instruction" in two forms - hipcc otherwise hoists the loads (and the unpacking)
// that two forms share above the branch that picks the form and carries them, spilled, through it.
template <int FORM> __device__ __forceinline__ void form_lane(int& lane_e, int& fr_e, int& fg_e) {
  asm volatile("; epilogue form %1" : "+v"(lane_e) : "n"(FORM));
  fr_e = lane_e & 15; fg_e = lane_e >> 4;
}

template <int FORM> __device__ __forceinline__ void form_touch(f32x4& v) { asm volatile("; form %1" : "+v"(v) : "n"(FORM)); }      // the same for an accumulator block

// gemm_tail on 8 consecutive bf16 columns (same operation order and rounding points); GELU / Snake / the second output
// belong to the fp32 DAC path and are rejected at launch for this kernel
__device__ __forceinline__ void gemm_tail8(const GemmArgs& p, int m, int n0, float (&y)[8], int zo, int zi, bf16_t* C) {
  typedef bf16_t T;
  if (p.acc_scale != 1.0f) {
#pragma unroll
    for (int i = 0; i < 8; ++i) y[i] *= p.acc_scale;
  }
  const int nv = p.vec_mod ? n0 % p.vec_mod : n0;
  if (p.bias) {
    float b[8];
    unpack8_bf16(*(const uint4*)((const T*)p.bias + zo * p.bias_bo + zi * p.bias_bi + nv), b);
#pragma unroll
    for (int i = 0; i < 8; ++i) y[i] += b[i];
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) y[i] = Num<T>::rnd(y[i]);
  if (p.div != 0.0f) {
#pragma unroll
    for (int i = 0; i < 8; ++i) y[i] = Num<T>::rnd(y[i] / p.div);
  }
  if (p.act == 1) {
#pragma unroll
    for (int i = 0; i < 8; ++i) y[i] = Num<T>::rnd(silu_f(y[i]));
  }
  if (p.colscale) {
    float c[8];
    unpack8_bf16(*(const uint4*)((const T*)p.colscale + nv), c);
#pragma unroll
    for (int i = 0; i < 8; ++i) y[i] = Num<T>::rnd(y[i] * c[i]);
  }
  if (p.res) {
    float r[8];
    unpack8_bf16(*(const uint4*)((const T*)p.res + zo * p.res_bo + zi * p.res_bi + (long)m * p.ldres + n0), r);
#pragma unroll
    for (int i = 0; i < 8; ++i) y[i] = Num<T>::rnd(y[i] + r[i]);
  }
  *(uint4*)(C + (long)m * p.ldc + n0) = pack8_bf16(y);
}

// DIAG (timing experiments only): 1 = no LDS-DMA in the loop, 2 = no fragment reads, 3 = no MFMAs, 4 = no epilogue
// (1-4 give wrong results); 5 = correct results + per-wave cycle sums (s_memtime) written to p.ws: {total, K loops,
// epilogue, tiles} x 8 waves per workgroup; 6 = correct results + per-wave, per-phase cycle sums {load part, wait at
// barrier 1, MFMA part, wait at barrier 2} x 4 phases; 7 = the epilogue without its global stores (wrong results)
//
// FP8: A and W are OCP e4m3 bytes (K-tile = 128 elements = the same 128-byte rows, so staging, ring and phases are unchanged);
// a phase is 8 v_mfma_scale_f32_16x16x128_f8f6f4 (unit block scales: twice the bf16 FLOPs per cycle) on fragments of 32
// consecutive K bytes per lane, and the accumulators are multiplied by a_scale[m] * w_scale[n] before the bf16 tails.
// TAIL: which epilogues an instantiation carries.  One kernel with every tail (split-K slabs, fused QKV, the row-layout tails) needs the
// union of their registers: the all-tails build of round 2 had 71 VGPR + 108 SGPR spills (272 bytes of scratch per lane) around its tile
// hand-over while the SwiGLU-only instantiation had 2 and the epilogue-less timing build none.  TAIL_ROWS = the row-layout tails only (plain
// store, column scale + residual: wo / w2, and the generic bias / activation form), TAIL_QKV = the fused QKV(G) tail only, TAIL_ALL = everything
// (split-K and the diagnostic builds).  The launcher picks the instantiation from the arguments.
typedef const __attribute__((address_space(4))) GemmArgs* kargs_t;      // the kernel's argument block in the kernarg segment
enum { TAIL_ALL = 0, TAIL_ROWS = 1, TAIL_QKV = 2, TAIL_FAST = 3, TAIL_FASTR = 4 };      // TAIL_FAST: y = T(acc) only (plain store); TAIL_FASTR: y = T(T(T(acc) * colscale) + residual) (wo, w2)
template <bool SWIGLU, int DIAG = 0, int LEAD = PP_LEAD, bool FP8 = false, int TAIL = TAIL_ALL>
__global__ void __launch_bounds__(512, 2) gemm_pp_kernel(const GemmArgs p) {
  typedef bf16_t T;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int BM = 256, BN = 256, KE = FP8 ? 128 : 64, ES = FP8 ? 1 : 2, UNIT = 16384, KTILE = 4 * UNIT;
  static_assert(LEAD >= 2 && LEAD <= 6, "see the WAR/RAW rules above");
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.Npad + BN - 1) / BN;
  const int ntiles = tiles_m * tiles_n;
  const int G = gridDim.x;                       // <= ntiles
  // virtual block id v -> tile: bijective XCD remap (workgroups that share an XCD walk one contiguous run of tiles;
  // G is a multiple of 8 or equals ntiles, so v % 8 == blockIdx.x % 8 for every tile of this workgroup)
  // The linear index b runs through strips of PP_GN tile columns, row by row inside a strip: the 32 tiles an XCD works
  // on at one time form an 8 x 4 block (8 A panels + 4 W panels through its L2 instead of 32 + 1).
  auto tile_of = [&](int v, int& tm, int& tn) __attribute__((always_inline)) {
    const int q = ntiles >> 3, r = ntiles & 7, xcd = v & 7, idx = v >> 3;
    const int b = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    const int GN = p.pp_gn > 0 ? p.pp_gn : PP_GN;
    const int strip = b / (GN * tiles_m), rem = b - strip * (GN * tiles_m);
    const int w = tiles_n - strip * GN < GN ? tiles_n - strip * GN : GN;
    tm = rem / w; tn = strip * GN + (rem - tm * w);
  };
  const int z = blockIdx.y, zo = z / p.nbi, zi = z - zo * p.nbi;
  const long a_z = zo * p.a_bo + zi * p.a_bi, w_z = zo * p.w_bo + zi * p.w_bi, c_z = zo * p.c_bo + zi * p.c_bi;

  const int kb_per_tap = p.K / KE;
  const int nk_total = kb_per_tap * p.taps;
  const int ks = p.ksplit > 1 ? p.ksplit : 1;
  const int split = blockIdx.z;
  const int it0 = (int)((long)nk_total * split / ks), it1 = (int)((long)nk_total * (split + 1) / ks);
  const int nk = it1 - it0;

  // ---- DMA sources: per unit type j, wave w copies rows 16w .. 16w + 15 of the unit as two 8-row pieces.  The per-lane
  // part of the address is a 32-bit byte offset that only changes with the output tile; everything that moves inside a
  // tile (K position, tap) and the batch offset are wave-uniform and live in the 64-bit scalar cursors.
  unsigned voff[4][2];
  const unsigned lda_b = (unsigned)(p.lda * ES), ldw_b = (unsigned)(p.ldw * ES);      // row pitches in bytes
  auto set_stage_tile = [&](int v) __attribute__((always_inline)) {
    int tm, tn;
    tile_of(v, tm, tn);
    // the lane coordinate behind an opaque asm: otherwise the lane-only parts of the eight offsets below (unit rows, swizzled chunks) are loop
    // invariants that hipcc keeps in registers through the K loop - which has none to spare, so it spilled them and reloaded them HERE, in the
    // hand-over K-tile, behind a vmcnt(0) that drained the LDS-DMA ring once per output tile
    int lane_s = hw_lane();      // (from mbcnt, not from `lane`: that variable would be spilled at kernel entry and reloaded here - behind a vmcnt(0))
    asm volatile("; stage tile" : "+v"(lane_s));
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int ur = (wid * 2 + i) * 8 + (lane_s >> 3);       // row inside the unit
        const int chunk = (lane_s & 7) ^ ((ur >> 1) & 7);        // source-side swizzle
        if (j == 0 || j == 3) {                                // A: unit row = wave row (ur >> 5) x 32 rows of this m-half
          int gm = tm * BM + (ur >> 5) * 64 + (j == 3 ? 32 : 0) + (ur & 31);
          gm = gm < p.M ? gm : p.M - 1;
          voff[j][i] = (unsigned)gm * lda_b + (unsigned)(chunk * 16);     // 32-bit on purpose (launch_sw refuses operands whose row offsets need more)
        } else {                                               // W: unit row = wave column (ur >> 6) x 64 rows of this n-half
          int gn = tn * BN + (ur >> 6) * 128 + (j == 2 ? 64 : 0) + (ur & 63);
          gn = gn < p.Npad ? gn : p.Npad - 1;
          voff[j][i] = (unsigned)gn * ldw_b + (unsigned)(chunk * 16);
        }
      }
  };
  // staging cursors (wave-uniform): A units j0, j3 of one K-tile share a_cur, W units j1, j2 share w_cur.  After the
  // last K-tile of an output tile the stream moves on to the next tile of this workgroup; after the last tile the
  // cursors stop, so the look-ahead units of the final phases re-stage the last K-tile into ring slots nobody reads
  // again: every phase issues exactly two DMAs per wave and the counted vmcnt holds to the end.
  const long a_tap_bytes = ((long)p.tap_shift * p.lda - (long)p.K) * ES;
  const int kb0 = it0 % kb_per_tap;
  const char* const a_start = (const char*)p.A + ((long)p.tap_base * p.lda + a_z + (long)(it0 / kb_per_tap) * p.tap_shift * p.lda) * ES + (long)kb0 * KBYTES;
  const char* const w_start = (const char*)p.W + w_z * ES + (long)it0 * KBYTES;
  const char* a_cur = a_start;
  const char* w_cur = w_start;
  int kb = kb0, kt_staged = 0, v_stage = blockIdx.x;
  bool in_loop = false;
  set_stage_tile(v_stage);
  // SWITCH = false: the caller knows that this unit is not the last one of an output tile (the steady K loop), so the
  // tile hand-over (new per-lane offsets, cursor reset) is compiled out of the hot path
  auto issue_unit = [&](auto jc, int buf, auto swc) __attribute__((always_inline)) {
    constexpr int j = decltype(jc)::value;
    constexpr bool SWITCH = decltype(swc)::value != 0;
    char* dst = smem + buf * KTILE + j * UNIT + wid * 2048;
    const char* base = (j == 0 || j == 3) ? a_cur : w_cur;
    // the 32-bit per-lane offsets go through an opaque asm at every use: hipcc otherwise zero-extends the eight of them ONCE, outside the
    // K loop, keeps the 64-bit copies (8 more VGPRs live through the loop) and adds the base on the VALU (two v_lshl_add_u64 per unit); from
    // a 32-bit value it selects `global_load_lds_dwordx4 voff32, s[base]`
    unsigned vo0 = voff[j][0], vo1 = voff[j][1];
    asm volatile("" : "+v"(vo0), "+v"(vo1));
    if (DIAG != 1 || !in_loop) {
      glds16(base + vo0, dst);
      glds16(base + vo1, dst + 1024);
    }
    if constexpr (!SWITCH) {
      if (j == 2) w_cur += KBYTES;
      if (j == 3) {
        ++kt_staged;
        a_cur += KBYTES;
        if (++kb == kb_per_tap) { kb = 0; a_cur += a_tap_bytes; }
      }
    } else {
      if (j == 2 && kt_staged + 1 < nk) w_cur += KBYTES;
      if (j == 3) {
        if (kt_staged + 1 < nk) {
          ++kt_staged;
          a_cur += KBYTES;
          if (++kb == kb_per_tap) { kb = 0; a_cur += a_tap_bytes; }
        } else if (v_stage + G < ntiles) {       // wave-uniform: on to the first K-tile of this workgroup's next tile
          v_stage += G;
          set_stage_tile(v_stage);
          kt_staged = 0; kb = kb0; a_cur = a_start; w_cur = w_start;
        }
      }
    }
  };

  // ---- fragment read addresses: lane (r = lane & 15, g = lane >> 4) reads row r of a 16-row fragment, k-chunk g (+ 4)
  const int wm = wid >> 1, wn = wid & 1;
  const int fr = lane & 15, fg = lane >> 4;
  // bf16: swizzled 16-byte chunk fg of k-half 0, k-half 1 is c0 ^ 64.  fp8: the lane's 32 K bytes are chunks 2 fg, 2 fg + 1
  const int c0 = (((FP8 ? 2 * fg : fg) ^ (fr >> 1)) & 7) << 4;
  constexpr int C1X = FP8 ? 16 : 64;
  const int a_rd0 = (wm * 32 + fr) * KBYTES + c0, a_rd1 = a_rd0 ^ C1X;   // + unit j0 / j3, + 16 i rows
  const int w_rd0 = (wn * 64 + fr) * KBYTES + c0, w_rd1 = w_rd0 ^ C1X;   // + unit j1 / j2, + 16 jn rows

  f32x4 acc[4][8];                 // [m fragment][n fragment]: lane holds C[m = 16 i + fr][n = 16 jn + 4 fg + r]
  bf16x8 af[2][2], wf[2][4][2];    // bf16: A (i, k-half) of the current m-half; W (n-half, jn, k-half)
  i32x8 af8[2], wf8[2][4];         // fp8: the lane's 32 K bytes of A (i) / W (n-half, jn) as ONE 8-register operand, assembled where the two 16-byte halves are
                                   // loaded (assembled at the MFMA instead, hipcc copied every fragment into a fresh tuple: 62 v_mov_b64 per K-tile in the steady loop)

  // ---- prologue: the first LEAD units; units 0 and 1 must have landed everywhere before phase 0 reads them
#pragma unroll
  for (int u = 0; u < LEAD; ++u) {
    if ((u & 3) == 0) issue_unit(IC<0>{}, (u >> 2) & 1, IC<1>{});
    if ((u & 3) == 1) issue_unit(IC<1>{}, (u >> 2) & 1, IC<1>{});
    if ((u & 3) == 2) issue_unit(IC<2>{}, (u >> 2) & 1, IC<1>{});
    if ((u & 3) == 3) issue_unit(IC<3>{}, (u >> 2) & 1, IC<1>{});
  }
  wait_vmcnt<2 * (LEAD - 2)>();
  __builtin_amdgcn_s_barrier();
  in_loop = true;
  if constexpr (DIAG == 2) {
#pragma unroll
    for (int i = 0; i < 2; ++i) { af[i][0] = af[i][1] = bf16x8{1, 2, 3, 4, 5, 6, 7, 8}; }
#pragma unroll
    for (int i = 0; i < 4; ++i) { wf[0][i][0] = wf[0][i][1] = wf[1][i][0] = wf[1][i][1] = bf16x8{1, 2, 3, 4, 5, 6, 7, 8}; }
  }

  // t = K-tile counter of this workgroup over ALL its output tiles (ring parity)
  unsigned long long ph_acc[4][4] = {};   // DIAG 6: [phase][load part, wait at barrier 1 (+ lgkmcnt), MFMA part, wait at barrier 2]
  auto phase = [&](auto qc, int t, auto swc) __attribute__((always_inline)) {
    constexpr int q = decltype(qc)::value;
    constexpr int mh = q >> 1, nh = (q == 1 || q == 2) ? 1 : 0;
    const char* sb = smem + (t & 1) * KTILE;
    unsigned long long s0 = 0, s1 = 0, s2 = 0, s3 = 0, s4 = 0;
    if constexpr (DIAG == 6) s0 = __builtin_amdgcn_s_memtime();
    // ---- load part: fragments first read in this phase
    if constexpr (DIAG != 2 && (q == 0 || q == 1)) {
      const char* wp = sb + (q == 0 ? 1 : 2) * UNIT;
#pragma unroll
      for (int jn = 0; jn < 4; ++jn) {
        if constexpr (FP8) {
          const i32x4 lo = *(const i32x4*)(wp + w_rd0 + jn * 16 * KBYTES), hi = *(const i32x4*)(wp + w_rd1 + jn * 16 * KBYTES);
          wf8[nh][jn] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        } else {
          wf[nh][jn][0] = *(const bf16x8*)(wp + w_rd0 + jn * 16 * KBYTES);
          wf[nh][jn][1] = *(const bf16x8*)(wp + w_rd1 + jn * 16 * KBYTES);
        }
      }
    }
    if constexpr (q == 0) __builtin_amdgcn_sched_barrier(0);
    if constexpr (DIAG != 2 && (q == 0 || q == 2)) {
      const char* ap = sb + (q == 0 ? 0 : 3) * UNIT;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        if constexpr (FP8) {
          const i32x4 lo = *(const i32x4*)(ap + a_rd0 + i * 16 * KBYTES), hi = *(const i32x4*)(ap + a_rd1 + i * 16 * KBYTES);
          af8[i] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        } else {
          af[i][0] = *(const bf16x8*)(ap + a_rd0 + i * 16 * KBYTES);
          af[i][1] = *(const bf16x8*)(ap + a_rd1 + i * 16 * KBYTES);
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- stage the unit LEAD phases ahead
    {
      constexpr int jj = (q + LEAD) & 3, dt = (q + LEAD) >> 2;
      issue_unit(IC<jj>{}, (t + dt) & 1, swc);
    }
    // ---- the units first read in the next phase have landed (this wave's part of them)
    if constexpr (q != 2) wait_vmcnt<2 * (LEAD - 2)>();
    if constexpr (DIAG == 6) s1 = __builtin_amdgcn_s_memtime();
    // the fragments are retired BEFORE the barrier (the wave would only wait at the barrier anyway; measured +3-4 % over
    // waiting behind it), and no s_setprio flips around the MFMA cluster (measured +3 %: the partner wave is in its load
    // part and barely competes for the vector issue port; the sched_barriers keep the cluster between the two barriers)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if constexpr (DIAG == 6) s2 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (DIAG == 3) {
#pragma unroll
      for (int i = 0; i < 2; ++i) asm volatile("" :: "v"(af[i][0]), "v"(af[i][1]));
#pragma unroll
      for (int jn = 0; jn < 4; ++jn) asm volatile("" :: "v"(wf[nh][jn][0]), "v"(wf[nh][jn][1]));
    } else
    if constexpr (FP8) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const i32x8 a8 = af8[i];
#pragma unroll
        for (int jn = 0; jn < 4; ++jn) {
          const i32x8 w8 = wf8[nh][jn];
          // formats 0 / 0 = e4m3 x e4m3; block scales 0x7f = 2^0 in every byte (the row scales are applied in the epilogue)
          acc[2 * mh + i][4 * nh + jn] =
              __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(w8, a8, acc[2 * mh + i][4 * nh + jn], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
        }
      }
    } else
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jn = 0; jn < 4; ++jn)
          acc[2 * mh + i][4 * nh + jn] =
              __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nh][jn][kk], af[i][kk], acc[2 * mh + i][4 * nh + jn], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (DIAG == 6) s3 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_barrier();
    if constexpr (DIAG == 6) {
      s4 = __builtin_amdgcn_s_memtime();
      ph_acc[q][0] += s1 - s0; ph_acc[q][1] += s2 - s1; ph_acc[q][2] += s3 - s2; ph_acc[q][3] += s4 - s3;
    }
  };

  // ---- epilogue pieces
  float* const my = (float*)(smem + 2 * KTILE) + wid * 1024;     // this wave's private 16 x 64 fp32 transposition area
  T* const C = (T*)p.C + c_z;
  // Epilogue-local copies of the lane coordinates, re-derived behind an opaque asm at every tile's epilogue: hipcc otherwise hoists the
  // epilogue's per-lane address arithmetic (LDS transposition addresses, row offsets) out of the tile loop and keeps it in registers ACROSS the
  // K loop, which already sits at 256 - the all-tails build carried 71 VGPR spills (scratch stores / loads around every tile hand-over, in the
  // same in-order vmcnt stream as the LDS-DMA units) for it.
  int lane_e = lane, fr_e = fr, fg_e = fg;
  // 16 rows x 64 columns held as v[c][r] = X[row fr][col 16 c + 4 fg + r]  ->  fn(row, c8, y[8]) with
  // y = X[row][8 c8 .. 8 c8 + 7], row = 8 it + lane / 8, c8 = lane % 8.  Same-wave LDS traffic is executed in order.
  auto rows = [&](const f32x4 (&v)[4], auto&& fn) __attribute__((always_inline)) {
#pragma unroll
    for (int c = 0; c < 4; ++c) *(f32x4*)(my + (fr_e * 16 + ((c * 4 + fg_e) ^ fr_e)) * 4) = v[c];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int row = it * 8 + (lane_e >> 3), c8 = lane_e & 7;
      const f32x4 a = *(const f32x4*)(my + (row * 16 + ((2 * c8) ^ row)) * 4);
      const f32x4 b = *(const f32x4*)(my + (row * 16 + ((2 * c8 + 1) ^ row)) * 4);
      float y[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
      fn(row, c8, y);
    }
  };

  int tcount = 0;
  unsigned long long t_start = 0, t_loop = 0, t_epi = 0, n_tiles = 0, rt_start = 0;
  if constexpr (DIAG == 5) { t_start = __builtin_amdgcn_s_memtime(); rt_start = __builtin_amdgcn_s_memrealtime(); }
#pragma unroll 1
  for (int v = blockIdx.x; v < ntiles; v += G) {
    int tile_m, tile_n;
    tile_of(v, tile_m, tile_n);
    // the eight DMA offsets of the tile being staged are RECOMPUTED here (same values): they were last written in the previous tile's hand-over and
    // nothing reads them during the epilogue, so hipcc spilled them there and reloaded them in front of the K loop - with the wait for that
    // reload INSIDE the steady loop (fp8 QKV instantiation: a vmcnt(0) per K-tile, the LDS-DMA ring drained every time).  Redefined at the tile
    // top, they are not live across the epilogue at all.
    if constexpr (FP8 && TAIL == TAIL_QKV) set_stage_tile(v_stage);      // (this instantiation only: in the others the register allocation came out worse with it)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    unsigned long long t_l0 = 0;
    if constexpr (DIAG == 5) t_l0 = __builtin_amdgcn_s_memtime();
    if (wid >= 4) __builtin_amdgcn_s_barrier();   // stagger: G1 runs one barrier behind G0 through the K loop
    // steady part: while K-tile t < nk - 2 is computed, the units staged (K-tiles t + 1, t + 2) all belong to this
    // output tile; the last two K-tiles run the general copy whose look-ahead crosses into the next output tile
    int t = 0;
#pragma unroll 1
    for (; t + 2 < nk; ++t, ++tcount) {
      phase(IC<0>{}, tcount, IC<0>{}); phase(IC<1>{}, tcount, IC<0>{}); phase(IC<2>{}, tcount, IC<0>{}); phase(IC<3>{}, tcount, IC<0>{});
    }
#pragma unroll 1
    for (; t < nk; ++t, ++tcount) {
      phase(IC<0>{}, tcount, IC<1>{}); phase(IC<1>{}, tcount, IC<1>{}); phase(IC<2>{}, tcount, IC<1>{}); phase(IC<3>{}, tcount, IC<1>{});
    }
    if (wid < 4) __builtin_amdgcn_s_barrier();    // level again: both groups run their epilogues side by side
    if constexpr (DIAG == 5) { const unsigned long long n = __builtin_amdgcn_s_memtime(); t_loop += n - t_l0; t_l0 = n; ++n_tiles; }

    lane_e = hw_lane();
    asm volatile("" : "+v"(lane_e));
    fr_e = lane_e & 15; fg_e = lane_e >> 4;
    // the tile coordinates too: everything the epilogue derives from them (64-bit row offsets of C / the residual, ...) is computed HERE,
    // behind the K loop, instead of at the top of the iteration and carried through it
    asm volatile("" : "+s"(tile_m), "+s"(tile_n));
    const int m_base = tile_m * BM + wm * 64;      // + 16 i + row
    const int n_base = tile_n * BN + wn * 128;     // + 64 h + 8 c8 (W-row index of the accumulator columns)
    // fp8 dequantisation: acc * (a_scale[m] * w_scale[n]).  The plain-store / column scale + residual tail (TAIL_FAST: wo, w2) applies it in the
    // ROW layout behind the LDS transposition (8 row scales + 16 column scales per lane instead of 4 + 32, and no 256 multiplies on the
    // accumulators while everything else is still live: this instantiation spilled 49 VGPRs); the tails that compute in the accumulator layout
    // (SwiGLU pairs, head norm / RoPE) and the generic ones dequantise there.
    constexpr bool IS_FAST = TAIL == TAIL_FAST || TAIL == TAIL_FASTR;
    constexpr bool ROWDEQ = FP8 && IS_FAST;
    if constexpr (FP8 && !ROWDEQ) {
      // dequantise in the accumulator layout: lane_e holds C[m_base + 16 i + fr_e][n_base + 16 jn + 4 fg_e + r]
      float sa[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) { const int m = m_base + 16 * i + fr_e; sa[i] = p.a_scale ? p.a_scale[m < p.M ? m : p.M - 1] : p.a_scale_const; }
#pragma unroll
      for (int jn = 0; jn < 8; ++jn) {
        const int n0 = n_base + 16 * jn + 4 * fg_e;
        const f32x4 sw = *(const f32x4*)(p.w_scale + (n0 + 3 < p.Npad ? n0 : p.Npad - 4));
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[i][jn][r] *= sa[i] * sw[r];
      }
    }
    if constexpr (DIAG == 4) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) asm volatile("" :: "v"(acc[i][j]));
    } else if (TAIL == TAIL_ALL && p.ksplit > 1) {
      // raw fp32 partial sums to the split-K workspace [split][Mpad][Npad]; the reduce kernel applies the tail
      float* ws = (float*)p.ws + (long)split * tiles_m * BM * (long)p.Npad;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const f32x4 v4[4] = {acc[i][4 * h], acc[i][4 * h + 1], acc[i][4 * h + 2], acc[i][4 * h + 3]};
          rows(v4, [&](int row, int c8, float (&y)[8]) __attribute__((always_inline)) {
            const int m = m_base + 16 * i + row, n0 = n_base + 64 * h + 8 * c8;
            if (n0 < p.Npad) {
              float* d = ws + (long)m * p.Npad + n0;
              *(f32x4*)d = f32x4{y[0], y[1], y[2], y[3]};
              *(f32x4*)(d + 4) = f32x4{y[4], y[5], y[6], y[7]};
            }
          });
        }
    } else if constexpr (SWIGLU) {
      // W rows come in 32-row groups [16 x w1 | 16 x w3]: fragments jn = 2k, 2k + 1 of one lane_e are the (a, b) pairs
      // of output columns 16 k + 4 fg_e + r (model.py:307); 64 output columns per wave
      const int j_base = tile_n * (BN / 2) + wn * 64;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        f32x4 v4[4];
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float a = Num<T>::rnd(acc[i][2 * k][r]), bb = Num<T>::rnd(acc[i][2 * k + 1][r]);
            v4[k][r] = Num<T>::rnd(Num<T>::rnd(silu_fast(a)) * bb);
          }
        rows(v4, [&](int row, int c8, float (&y)[8]) __attribute__((always_inline)) {
          const int m = m_base + 16 * i + row, j0 = j_base + 8 * c8;
          if (m < p.M && j0 < (p.N >> 1)) {
            if constexpr (FP8) {
              if (p.c8) {      // static activation scale: e4m3 bytes for the w2 GEMM (the values are bf16-rounded already)
                float z[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) z[e] = __builtin_amdgcn_fmed3f(y[e] * p.c8_inv, -448.0f, 448.0f);
                int lo = __builtin_amdgcn_cvt_pk_fp8_f32(z[0], z[1], 0, false);
                lo = __builtin_amdgcn_cvt_pk_fp8_f32(z[2], z[3], lo, true);
                int hi = __builtin_amdgcn_cvt_pk_fp8_f32(z[4], z[5], 0, false);
                hi = __builtin_amdgcn_cvt_pk_fp8_f32(z[6], z[7], hi, true);
                *(int2*)((uint8_t*)p.c8 + (long)m * p.c8_ld + j0) = int2{lo, hi};
                return;
              }
            }
            *(uint4*)(C + (long)m * p.ldc + j0) = pack8_bf16(y);
          }
        });
      }
    } else if ((TAIL == TAIL_ALL || TAIL == TAIL_QKV) && (TAIL == TAIL_QKV || p.qkv_mode)) {
      // Fused QKV(G) tail.  Like the fast tail below it comes in BRANCH-FREE forms, one per (section kind, interior | edge tile), picked by
      // wave-uniform branches OUTSIDE the code that touches memory: a load or store behind a branch - the former `if (sec < 2)` / `if (do_rope)`
      // around the rope / norm-weight requests, the per-lane `if (m < M)` around every store - leaves hipcc's wait-count pass guessing at the
      // joins, and it answered with `s_waitcnt vmcnt(0)` at every piece, at the tile end, at the head of the next tile and inside the K-tiles of
      // the tile hand-over, where it drains the LDS-DMA ring (round 3, read off the ISA).  Edge tiles (ragged last row of tiles, or a sequence
      // length that is not a multiple of 8 in the V section) issue their predicated stores from inline asm (hidden_store*), see the fast tail.
      // the tail's arguments are read from the kernarg segment HERE (scalar loads behind an opaque pointer) instead of living in SGPRs through the
      // K loop: with them the kernel needed more than the 102 SGPRs there are and kept the LDS-DMA cursors in VGPRs (v_readfirstlane before every
      // DMA pair, two VGPR pairs spilled around the K loop)
      kargs_t pk = (kargs_t)__builtin_amdgcn_kernarg_segment_ptr();
      asm volatile("; epilogue arguments" : "+s"(pk));
      const int D = pk->qkv_D;
      const int sec = (tile_n * BN) / D;            // the whole tile lies in one of q | k | v | gate (D % 256 == 0, so n < N everywhere)
      const bool full = m_base + 64 <= pk->M;         // wave-uniform
      if (sec == 2) {
        // V: Vt[b][h * 128 + d][token]; this wave holds 64 tokens x the 128 d of one head.  Per 16 d (fragment jn) the
        // four token fragments go through the private area as [d][token] and leave as 8 tokens per lane_e.
        const int hd_base = n_base - 2 * D;
        auto vsec = [&](auto fullc) __attribute__((always_inline)) {
          constexpr bool FULL = decltype(fullc)::value != 0;      // interior tile and S % 8 == 0: every lane stores 8 whole tokens of one batch row
          form_lane<32 + FULL>(lane_e, fr_e, fg_e);
#pragma unroll
          for (int jn = 0; jn < 8; ++jn) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
              for (int r = 0; r < 4; ++r) my[(4 * fg_e + r) * 64 + ((16 * i + fr_e) ^ (16 * fg_e))] = Num<T>::rnd(acc[i][jn][r]);     // token ^ 16 (d >> 2): the four
#pragma unroll                                                                                                                 // fg groups of a ds_write_b32 hit 4 bank groups, not 1
            for (int it = 0; it < 2; ++it) {
              const int dl = it * 8 + (lane_e >> 3), t8 = lane_e & 7;
              const int tsw = (8 * t8) ^ (16 * (dl >> 2));
              const f32x4 a = *(const f32x4*)(my + dl * 64 + tsw);
              const f32x4 b = *(const f32x4*)(my + dl * 64 + tsw + 4);
              const float y[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
              const int m = tile_m * BM + wm * 64 + 8 * t8;
              const int hd = hd_base + 16 * jn + dl;
              if constexpr (FULL) {
                const int bq = m / pk->qkv_S, sidx = m - bq * pk->qkv_S;
                *(uint4*)((T*)pk->vt + (long)bq * pk->vt_row_stride + (long)hd * pk->vt_ld + sidx) = pack8_bf16(y);
              } else if (m < pk->M) {
                const int bq = m / pk->qkv_S, sidx = m - bq * pk->qkv_S;
                T* dst = (T*)pk->vt + (long)bq * pk->vt_row_stride + (long)hd * pk->vt_ld + sidx;
                if ((pk->qkv_S & 7) == 0 && m + 7 < pk->M) {
                  hidden_store16(dst, pack8_bf16(y));
                } else {
                  for (int e = 0; e < 8 && m + e < pk->M; ++e) {
                    const int bb = (m + e) / pk->qkv_S, ss = (m + e) - bb * pk->qkv_S;
                    hidden_store2((T*)pk->vt + (long)bb * pk->vt_row_stride + (long)hd * pk->vt_ld + ss, Num<T>::st(y[e]));
                  }
                }
              }
            }
          }
        };
        if (full && (pk->qkv_S & 7) == 0) vsec(IC<1>{}); else vsec(IC<0>{});
      } else {
        // q | k | gate sections.  The rope values (cos, sin) and norm weights of a 16-token x 64-column piece are requested
        // as ONE batch (four 16-byte + four 8-byte loads) in front of its arithmetic: one L2 round trip per piece
        // instead of one per 16-column fragment (in-situ: the loads issued one at a time cost 91 us of a 543 us launch at
        // M = 15360).  Larger batches (a whole row, or a row ahead) pushed the kernel over its 256 registers and spilled
        // the accumulators.
        const int nd0 = n_base - sec * D;        // h * 128 (d = 16 jn + 4 fg_e + r)
        auto qkg = [&](auto normc, auto ropec, auto actc, auto fullc) __attribute__((always_inline)) {
          constexpr bool NORM = decltype(normc)::value != 0, ROPE = decltype(ropec)::value != 0, ACT = decltype(actc)::value != 0, FULL = decltype(fullc)::value != 0;
          constexpr int FORM = 16 + NORM * 8 + ROPE * 4 + ACT * 2 + FULL;
          form_lane<FORM>(lane_e, fr_e, fg_e);
          // the norm weights (4 x 8 bytes) and RoPE values (4 x 16 bytes) of piece (i, h) = 16 tokens x 64 columns are requested as one batch ONE
          // PIECE AHEAD: behind the arithmetic of the piece before (its own batch is dead by then: same registers) and IN FRONT OF that piece's
          // stores - the counted wait for the batch then leaves those stores in flight (vmcnt is in order: requested behind them, the batch
          // could only be waited for together with the stores' round trip, once per piece)
          float4 cs[4];
          uint2 w4p[4];
          auto request = [&](int i_, int h_) __attribute__((always_inline)) {
            const T* wp = (const T*)pk->qk_w + (long)sec * D + nd0 + 64 * h_ + 4 * fg_e;
#pragma unroll
            for (int j = 0; j < 4; ++j) w4p[j] = *(const uint2*)(wp + 16 * j);
            if constexpr (ROPE) {
              const int m = m_base + 16 * i_ + fr_e;
              const int pos = pk->pos0 + m % pk->qkv_S;
              const float4* rp = (const float4*)((const float2*)pk->rope + (long)pos * 64) + fg_e;     // pairs 8 jn + 2 fg_e, + 1
#pragma unroll
              for (int j = 0; j < 4; ++j) cs[j] = rp[4 * (4 * h_ + j)];
            }
          };
          if constexpr (NORM) request(0, 0);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            f32x4 yv[8];
#pragma unroll
            for (int jn = 0; jn < 8; ++jn) {
              form_touch<FORM>(acc[i][jn]);      // (the rounding and the sum of squares below are the same in four forms: keep them IN the forms)
#pragma unroll
              for (int r = 0; r < 4; ++r) yv[jn][r] = Num<T>::rnd(acc[i][jn][r]);
            }
            if constexpr (ACT) {      // the attention epilogue's sigmoid, moved here (same operations on the same bf16 values)
#pragma unroll
              for (int jn = 0; jn < 8; ++jn)
#pragma unroll
                for (int r = 0; r < 4; ++r) yv[jn][r] = Num<T>::rnd(sigmoid_fast(yv[jn][r]));
            }
            float rs = 0.f;
            if constexpr (NORM) {
              // per-head RMSNorm (model.py:86-104) on the 128 columns of (token m, this wave's head): 32 values in this
              // lane_e, the other 96 in lanes fr_e + 16, + 32, + 48; then interleaved-pair RoPE on heads < rope_heads
              float ss = 0.f;
#pragma unroll
              for (int jn = 0; jn < 8; ++jn)
#pragma unroll
                for (int r = 0; r < 4; ++r) ss += yv[jn][r] * yv[jn][r];
              ss += __shfl_xor(ss, 16, 64);
              ss += __shfl_xor(ss, 32, 64);
              rs = rsqrtf(ss / 128.0f + pk->qk_eps);
            }
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              if constexpr (NORM) {
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                  const int jn = 4 * h + j;
                  float w4[4];
                  Vec4<T>::unpack(w4p[j], w4);
#pragma unroll
                  for (int r = 0; r < 4; ++r) yv[jn][r] = Num<T>::rnd(__fmul_rn(__fmul_rn(yv[jn][r], rs), w4[r]));
                  if constexpr (ROPE) {
                    const float4 c4 = cs[j];                       // two (cos, sin) pairs
                    const float a0 = yv[jn][0], b0 = yv[jn][1], a1 = yv[jn][2], b1 = yv[jn][3];
                    yv[jn][0] = __fsub_rn(__fmul_rn(a0, c4.x), __fmul_rn(b0, c4.y));
                    yv[jn][1] = __fadd_rn(__fmul_rn(a0, c4.y), __fmul_rn(b0, c4.x));
                    yv[jn][2] = __fsub_rn(__fmul_rn(a1, c4.z), __fmul_rn(b1, c4.w));
                    yv[jn][3] = __fadd_rn(__fmul_rn(a1, c4.w), __fmul_rn(b1, c4.z));
                  }
                }
                __builtin_amdgcn_sched_barrier(0);
                if (2 * i + h + 1 < 8) request(h ? i + 1 : i, h ? 0 : 1);      // the next piece's batch, in front of this piece's stores
                __builtin_amdgcn_sched_barrier(0);
              }
              const f32x4 v4[4] = {yv[4 * h], yv[4 * h + 1], yv[4 * h + 2], yv[4 * h + 3]};
              rows(v4, [&](int row, int c8, float (&y)[8]) __attribute__((always_inline)) {
                const int m = m_base + 16 * i + row, n0 = n_base + 64 * h + 8 * c8;
                if constexpr (FULL) *(uint4*)(C + (long)m * pk->ldc + n0) = pack8_bf16(y);
                else if (m < pk->M) hidden_store16(C + (long)m * pk->ldc + n0, pack8_bf16(y));
              });
            }
            __builtin_amdgcn_sched_barrier(0);      // one 16-row block at a time: without the fence the scheduler starts the next block's rounding / sum of squares under this
          }                                         // block's second half and the form needs ~10 more registers than it has (23 spills per form, each reload a vmcnt(0))
        };
        const bool do_rope = sec < 2 && (nd0 >> 7) < pk->rope_heads;
        if (sec < 2) {
          if (do_rope) { if (full) qkg(IC<1>{}, IC<1>{}, IC<0>{}, IC<1>{}); else qkg(IC<1>{}, IC<1>{}, IC<0>{}, IC<0>{}); }
          else         { if (full) qkg(IC<1>{}, IC<0>{}, IC<0>{}, IC<1>{}); else qkg(IC<1>{}, IC<0>{}, IC<0>{}, IC<0>{}); }
        } else if (pk->qkv_gate_act) {
          if (full) qkg(IC<0>{}, IC<0>{}, IC<1>{}, IC<1>{}); else qkg(IC<0>{}, IC<0>{}, IC<1>{}, IC<0>{});
        } else {
          if (full) qkg(IC<0>{}, IC<0>{}, IC<0>{}, IC<1>{}); else qkg(IC<0>{}, IC<0>{}, IC<0>{}, IC<0>{});
        }
      }
    } else if (TAIL == TAIL_QKV) {
      // (this instantiation is only launched with qkv_mode set)
    } else if (IS_FAST || (p.acc_scale == 1.0f && !p.bias && p.div == 0.0f && p.act == 0 && !p.vec_mod && DIAG != 7)) {
      // fast path of the big EchoDiT linears: y = T(acc) [* colscale] [+ residual].  Fully unrolled (static accumulator
      // reads), addresses hoisted: one 64-bit per-lane_e offset per tile, everything else wave-uniform; the residual rows of
      // a piece are requested before its LDS round trip; interior tiles skip the per-element bounds tests.
      //
      // COMPILE-TIME forms (TAIL_FAST: no column scale, no residual; TAIL_FASTR: both - wo, w2) x (interior tile | edge tile): with the
      // column scale / residual / bounds decisions taken at run time, every load and store of this tail sat behind a wave-uniform branch,
      // hipcc's wait-count pass lost track of how many memory operations were in flight at the joins and put `s_waitcnt vmcnt(0)` in front of
      // every use of a residual chunk, behind every store and at the head of the next tile (round 3, read off the ISA: 10 vmcnt(0) in this
      // tail against none in the SwiGLU one): each of the 16 stores of a wave was waited for before the next piece started, the residual
      // prefetch bought nothing (depth 1 / 3 / 7 measured equal) and the LDS-DMA look-ahead of the next tile was drained at every hand-over.
      // The branch-free forms let the compiler count: stores are fire and forget, residual chunks arrive a piece ahead.  A PREDICATED store is
      // enough to lose it again (measured on the ISA: with per-lane `if (m < M)` stores in the edge form the vmcnt(0)s were back, one of them at
      // the head of the steady K loop, where it drains the LDS-DMA ring at every K-tile).  So the edge form (last, ragged row of tiles; these two
      // instantiations are launched with N % 256 == 0 only) loads its residual chunks unconditionally from clamped rows and issues its
      // predicated stores from inline asm, which the wait-count pass does not see: the hardware counter then runs AHEAD of the compiler's
      // count by the stores issued, so every counted wait of the compiler waits for at least what it meant to wait for (in-order vmcnt).
      // (Storing rows >= M to row M - 1 again - they hold its values bit for bit - is not an option: the engine's residual is in place, so a
      // second read-modify-write of row M - 1 would add the residual twice.)
      const bool full = m_base + 64 <= p.M && n_base + 128 <= p.N;          // wave-uniform
      const T* const resp = p.res ? (const T*)p.res + zo * p.res_bo + zi * p.res_bi : nullptr;
      auto fast_tail = [&](auto csc, auto rsc, auto fullc) __attribute__((always_inline)) {
        constexpr int CSM = decltype(csc)::value, RSM = decltype(rsc)::value, FM = decltype(fullc)::value;   // 0 = no, 1 = yes, 2 = ask the arguments
        form_lane<64 + CSM * 9 + RSM * 3 + FM>(lane_e, fr_e, fg_e);
        const bool has_cs = CSM == 2 ? p.colscale != nullptr : CSM == 1;
        const bool has_res = RSM == 2 ? resp != nullptr : RSM == 1;
        auto inb = [&](int m, int n0) __attribute__((always_inline)) { return FM == 1 ? true : FM == 2 ? (full || (m < p.M && n0 < p.N)) : (m < p.M && n0 < p.N); };
        const int row0 = lane_e >> 3, c8 = lane_e & 7;
        const long off0 = (long)(m_base + row0) * p.ldc + n_base + 8 * c8;
        const long roff0 = (long)(m_base + row0) * p.ldres + n_base + 8 * c8;
        float cs[2][8];
        float dq_w[2][8], dq_a[4][2];      // ROWDEQ: w_scale of this lane's 2 x 8 columns, a_scale of its 8 rows
        if constexpr (ROWDEQ) {
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const int n0 = n_base + 64 * h + 8 * c8;
            const float* wsp = p.w_scale + (n0 + 7 < p.Npad ? n0 : p.Npad - 8);
            const f32x4 w0 = *(const f32x4*)wsp, w1 = *(const f32x4*)(wsp + 4);
            dq_w[h][0] = w0[0]; dq_w[h][1] = w0[1]; dq_w[h][2] = w0[2]; dq_w[h][3] = w0[3];
            dq_w[h][4] = w1[0]; dq_w[h][5] = w1[1]; dq_w[h][6] = w1[2]; dq_w[h][7] = w1[3];
          }
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int it = 0; it < 2; ++it) {
              const int m = m_base + 16 * i + 8 * it + row0;
              const float av = *(p.a_scale ? p.a_scale + (m < p.M ? m : p.M - 1) : p.w_scale);      // always a load (no branch around it), from a valid address
              dq_a[i][it] = p.a_scale ? av : p.a_scale_const;
            }
        }
        if (has_cs) {
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const int n0 = n_base + 64 * h + 8 * c8;
            unpack8_bf16(*(const uint4*)((const T*)p.colscale + (FM != 2 || n0 < p.N ? n0 : p.N - 8)), cs[h]);
          }
        }
        // the residual chunks of piece pc + 1 are requested before piece pc goes through LDS and is stored: a load issued
        // behind a store would wait for that store's round trip too (one in-order vmcnt), once per piece
        constexpr int RES_AHEAD = PP_RES_AHEAD, RQN = RES_AHEAD + 1;
        uint4 rq[RQN][2];
        auto load_res = [&](int pc, uint4 (&r)[2]) __attribute__((always_inline)) {
          const int i = pc >> 1, h = pc & 1;
#pragma unroll
          for (int it = 0; it < 2; ++it) {
            r[it] = uint4{0, 0, 0, 0};
            const int m = m_base + 16 * i + 8 * it + row0, n0 = n_base + 64 * h + 8 * c8;
            if constexpr (FM == 2) {
              if (has_res && inb(m, n0)) r[it] = *(const uint4*)(resp + roff0 + (long)(16 * i + 8 * it) * p.ldres + 64 * h);
            } else if (has_res) {
              if constexpr (FM == 1) r[it] = *(const uint4*)(resp + roff0 + (long)(16 * i + 8 * it) * p.ldres + 64 * h);
              else r[it] = *(const uint4*)(resp + (long)(m < p.M ? m : p.M - 1) * p.ldres + n0);      // edge tile: row M - 1 again, see above
            }
          }
        };
#pragma unroll
        for (int a = 0; a < RES_AHEAD && a < 8; ++a) load_res(a, rq[a % RQN]);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const int pc = 2 * i + h;
            if (pc + RES_AHEAD < 8) load_res(pc + RES_AHEAD, rq[(pc + RES_AHEAD) % RQN]);
            const uint4 (&rr)[2] = rq[pc % RQN];
            if constexpr (DIAG != 9) {
#pragma unroll
              for (int c = 0; c < 4; ++c) *(f32x4*)(my + (fr_e * 16 + ((c * 4 + fg_e) ^ fr_e)) * 4) = acc[i][4 * h + c];
            }
#pragma unroll
            for (int it = 0; it < 2; ++it) {
              const int row = it * 8 + row0;
              f32x4 a, b;
              if constexpr (DIAG == 9) { a = acc[i][4 * h + 2 * it]; b = acc[i][4 * h + 2 * it + 1]; }      // timing experiment: no LDS round trip (wrong layout)
              else {
                a = *(const f32x4*)(my + (row * 16 + ((2 * c8) ^ row)) * 4);
                b = *(const f32x4*)(my + (row * 16 + ((2 * c8 + 1) ^ row)) * 4);
              }
              float y[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
              if constexpr (ROWDEQ) {
#pragma unroll
                for (int e = 0; e < 8; ++e) y[e] *= dq_a[i][it] * dq_w[h][e];      // the same two fp32 products as the accumulator-layout form
              }
#pragma unroll
              for (int e = 0; e < 8; ++e) y[e] = Num<T>::rnd(y[e]);
              if (has_cs) {
#pragma unroll
                for (int e = 0; e < 8; ++e) y[e] = Num<T>::rnd(y[e] * cs[h][e]);
              }
              if (has_res) {
                float r[8];
                unpack8_bf16(rr[it], r);
#pragma unroll
                for (int e = 0; e < 8; ++e) y[e] = Num<T>::rnd(y[e] + r[e]);
              }
              const int m = m_base + 16 * i + row, n0 = n_base + 64 * h + 8 * c8;
              if constexpr (DIAG == 8) {           // timing experiment: the fast-path epilogue without its global stores
                const uint4 pk = pack8_bf16(y);
                asm volatile("" :: "v"(pk.x), "v"(pk.y), "v"(pk.z), "v"(pk.w));
              } else
              if constexpr (FM == 0) {
                if (m < p.M) hidden_store16(C + (long)m * p.ldc + n0, pack8_bf16(y));
              } else if (inb(m, n0)) *(uint4*)(C + off0 + (long)(16 * i + 8 * it) * p.ldc + 64 * h) = pack8_bf16(y);
            }
          }
      };
      if constexpr (TAIL == TAIL_FAST) {
        if (full) fast_tail(IC<0>{}, IC<0>{}, IC<1>{}); else fast_tail(IC<0>{}, IC<0>{}, IC<0>{});
      } else if constexpr (TAIL == TAIL_FASTR) {
        if (full) fast_tail(IC<1>{}, IC<1>{}, IC<1>{}); else fast_tail(IC<1>{}, IC<1>{}, IC<0>{});
      } else {
        fast_tail(IC<2>{}, IC<2>{}, IC<2>{});
      }
    } else if constexpr (!IS_FAST) {
      // the tail is emitted once (runtime loop over the 8 pieces); the accumulators of piece 2 i + h are picked by
      // static register reads pinned with an empty asm (merged stores would turn `acc` into a scratch array)
#pragma unroll 1
      for (int piece = 0; piece < 8; ++piece) {
        f32x4 v4[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int h = 0; h < 2; ++h)
            if (piece == 2 * i + h) {
#pragma unroll
              for (int c = 0; c < 4; ++c) { v4[c] = acc[i][4 * h + c]; asm volatile("" : "+v"(v4[c])); }
            }
        const int i_ = piece >> 1, h_ = piece & 1;
        rows(v4, [&](int row, int c8, float (&y)[8]) __attribute__((always_inline)) {
          const int m = m_base + 16 * i_ + row, n0 = n_base + 64 * h_ + 8 * c8;
          if constexpr (DIAG == 7) {          // timing experiment: the whole epilogue but the global stores
            asm volatile("" :: "v"(y[0]), "v"(y[1]), "v"(y[2]), "v"(y[3]), "v"(y[4]), "v"(y[5]), "v"(y[6]), "v"(y[7]));
          } else {
            if (m < p.M && n0 < p.N) gemm_tail8(p, m, n0, y, zo, zi, C);
          }
        });
      }
    }
    if constexpr (DIAG == 5) t_epi += __builtin_amdgcn_s_memtime() - t_l0;
  }
  if constexpr (DIAG == 6) {
    if (lane == 0 && p.ws) {
      unsigned long long* o = (unsigned long long*)p.ws + ((long)blockIdx.x * 8 + wid) * 16;
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) o[q * 4 + e] = ph_acc[q][e];
    }
  }
  if constexpr (DIAG == 5) {
    if (lane == 0 && p.ws) {
      unsigned long long* o = (unsigned long long*)p.ws + ((long)blockIdx.x * 8 + wid) * 8;
      o[0] = __builtin_amdgcn_s_memtime() - t_start; o[1] = t_loop; o[2] = t_epi; o[3] = __builtin_amdgcn_s_memrealtime() - rt_start;   // o[3]: 100 MHz ticks
      o[4] = 0; o[5] = 0; o[6] = n_tiles; o[7] = 0;
    }
  }
  wait_vmcnt<0>();   // the re-staged look-ahead units must not land after the workgroup has released its LDS
}

int pp_num_cus() {
  static std::atomic<int> cached[64];          // per device ordinal; 0 = not queried yet
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::atomic<int>& slot = cached[dev & 63];
  int n = slot.load(std::memory_order_relaxed);
  if (!n) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
    if (n <= 0) n = 256;
    n &= ~7;                                   // a multiple of the 8 XCDs (tile_of)
    if (const char* e = getenv("ECHO_PP_GRID")) { const int v = atoi(e); if (v >= 8) n = v & ~7; }
    slot.store(n, std::memory_order_relaxed);
  }
  return n;
}

template <bool SW, int DIAG, int LEAD, bool FP8, int TAIL>
hipError_t launch_pp_tail(const GemmArgs& g_in, hipStream_t st);

template <bool SW, int DIAG = 0, int LEAD = PP_LEAD, bool FP8 = false>
hipError_t launch_pp(const GemmArgs& g, hipStream_t st) {
  // the production builds (DIAG 0, default LEAD) come in tail-specialised instantiations (see gemm_pp_kernel); ECHO_PP_TAILS=0 forces the
  // all-tails build (A/B aid)
  if constexpr (DIAG == 0 && LEAD == PP_LEAD && !SW) {
    static const bool split = getenv("ECHO_PP_TAILS") ? atoi(getenv("ECHO_PP_TAILS")) != 0 : true;
    if (split && g.ksplit <= 1) {
      if (g.qkv_mode) return launch_pp_tail<SW, DIAG, LEAD, FP8, TAIL_QKV>(g, st);
      if (g.acc_scale == 1.0f && !g.bias && g.div == 0.0f && g.act == 0 && !g.vec_mod) {
        if (!(g.N & 255)) {      // the branch-free edge form of these two needs whole tile columns
          if (!g.colscale && !g.res) return launch_pp_tail<SW, DIAG, LEAD, FP8, TAIL_FAST>(g, st);
          if (g.colscale && g.res) return launch_pp_tail<SW, DIAG, LEAD, FP8, TAIL_FASTR>(g, st);
        }
      }
      return launch_pp_tail<SW, DIAG, LEAD, FP8, TAIL_ROWS>(g, st);
    }
  }
  return launch_pp_tail<SW, DIAG, LEAD, FP8, TAIL_ALL>(g, st);
}

template <bool SW, int DIAG, int LEAD, bool FP8, int TAIL>
hipError_t launch_pp_tail(const GemmArgs& g_in, hipStream_t st) {
  GemmArgs g = g_in;
  static const int env_gn = getenv("ECHO_PP_GN") ? atoi(getenv("ECHO_PP_GN")) : 0;
  if (g.pp_gn <= 0 && env_gn > 0) g.pp_gn = env_gn;
  static std::atomic<unsigned long long> prepared{0};
  auto kern = gemm_pp_kernel<SW, DIAG, LEAD, FP8, TAIL>;
  constexpr int SMEM = 2 * 4 * 16384 + 32 * 256 * 4;      // ring of two K-tiles + epilogue slab = 160 KiB
  if (hipError_t e = ensure_dyn_lds((const void*)kern, SMEM, prepared); e != hipSuccess) return e;
  const int tiles_m = (g.M + 255) / 256, tiles_n = (g.Npad + 255) / 256;
  const int ntiles = tiles_m * tiles_n;
  const int ks = g.ksplit > 1 ? g.ksplit : 1;
  const int ncu = pp_num_cus();
  dim3 grid(ntiles < ncu ? ntiles : ncu, g.nbatch, ks);   // persistent: one workgroup per CU walks the tile list
  hipLaunchKernelGGL(kern, grid, dim3(512), SMEM, st, g);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess || ks == 1) return e;
  const int Mpad = tiles_m * 256;
  const long items = (long)g.M * (SW ? g.Npad / 8 : g.Npad / 4);
  hipLaunchKernelGGL((splitk_reduce_kernel<bf16_t, SW>), dim3((unsigned)((items + 255) / 256)), dim3(256), 0, st, g, Mpad);
  return hipGetLastError();
}


template <bool SW>
hipError_t launch_pp_sw(const GemmArgs& g, hipStream_t st) {
  if (g.cfg == 5 || g.cfg >= 100) {
    // 16-byte row stores / loads: every row pitch and column count the epilogue touches must be a multiple of 8
    if ((g.N & 7) || (g.ldc & 7) || (g.res && (g.ldres & 7)) || (g.swiglu && (g.N & 15)) || (g.vec_mod & 7) || g.act == 2 ||
        g.snake_alpha || g.C2 || !g.store_main ||
        (g.qkv_mode && ((g.qkv_D & 255) || (g.vt_ld & 7))))
      return hipErrorInvalidValue;
    // the per-lane part of a DMA source address is a 32-bit byte offset (row * pitch + chunk): refuse operands it cannot span
    const long es = g.fp8 ? 1 : 2;
    if ((long)(g.M - 1) * g.lda * es + 128 >= (1L << 32) || (long)(g.Npad - 1) * g.ldw * es + 128 >= (1L << 32)) return hipErrorInvalidValue;
  }
  if (g.fp8 && g.cfg != 5 && g.cfg != 104 && g.cfg != 105 && g.cfg != 108) return hipErrorInvalidValue;   // fp8 operands exist for the ping-pong kernel (and three of its diagnostic builds) only
  if (g.c8 && !(g.fp8 && SW && g.ksplit <= 1 && (g.c8_ld & 7) == 0 && g.c8_inv > 0.0f)) return hipErrorInvalidValue;   // e4m3 output: SwiGLU tail of the fp8 kernel only
  if (g.cfg == 5) {
    {
      if (g.fp8) {
        if ((!g.a_scale && !(g.a_scale_const > 0.0f)) || !g.w_scale || (g.K & 127) || (g.lda & 15) || (g.ldw & 15) || (g.Npad & 3)) return hipErrorInvalidValue;
        return launch_pp<SW, 0, PP_LEAD, true>(g, st);
      }
      return launch_pp<SW>(g, st);
    }
  }
  if (g.cfg >= 100) {   // timing experiments (tools/bench_gemm.py --diag): wrong results by construction
    if constexpr (!SW) {
      if (g.fp8) {      // tools/prof_fp8.py: no epilogue (wrong results) / cycle sums per wave / per-phase cycle sums
        if ((!g.a_scale && !(g.a_scale_const > 0.0f)) || !g.w_scale || (g.K & 127) || (g.lda & 15) || (g.ldw & 15) || (g.Npad & 3)) return hipErrorInvalidValue;
        if (g.cfg == 104) return launch_pp<false, 4, PP_LEAD, true>(g, st);
        if (g.cfg == 105) return launch_pp<false, 5, PP_LEAD, true>(g, st);
        if (g.cfg == 108) return launch_pp<false, 6, PP_LEAD, true>(g, st);
        return hipErrorInvalidValue;
      }
      if (g.cfg == 101) return launch_pp<false, 1>(g, st);
      if (g.cfg == 102) return launch_pp<false, 2>(g, st);
      if (g.cfg == 103) return launch_pp<false, 3>(g, st);
      if (g.cfg == 104) return launch_pp<false, 4>(g, st);
      if (g.cfg == 105) return launch_pp<false, 5>(g, st);
      if (g.cfg == 108) return launch_pp<false, 6>(g, st);
      if (g.cfg == 109) return launch_pp<false, 7>(g, st);
      if (g.cfg == 110) return launch_pp<false, 8>(g, st);      // fast-path epilogue without global stores
      if (g.cfg == 111) return launch_pp<false, 9>(g, st);      // fast-path epilogue without the LDS round trip
      if (g.cfg == 106) return launch_pp<false, 0, 6>(g, st);
      if (g.cfg == 107) return launch_pp<false, 0, 4>(g, st);
    }
    return hipErrorInvalidValue;
  }
  return hipErrorInvalidValue;
}
}  // namespace

hipError_t launch_gemm_pp(const GemmArgs& g, hipStream_t st) {
  return g.swiglu ? launch_pp_sw<true>(g, st) : launch_pp_sw<false>(g, st);
}
