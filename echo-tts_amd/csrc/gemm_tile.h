// Tile kernels (gemm_nt_kernel, plans 0-4 / 6-9) and their launcher template, shared by gemm.hip (bf16 instantiations) and gemm_f32.hip (fp32 /
// split-3 instantiations): two translation units that compile in parallel.
#pragma once
#include "gemm_common.h"

int gemm_tile_m(int cfg);
int gemm_num_cfgs();

namespace {

// Tile configuration: BM x BN output tile, NWM x NWN waves (each wave (BM/NWM) x (BN/NWN)), STAGES LDS buffers.
template <int BM_, int BN_, int NWM_, int NWN_, int STAGES_>
struct TileCfg {
  static constexpr int BM = BM_, BN = BN_, NWM = NWM_, NWN = NWN_, STAGES = STAGES_;
  static constexpr int NW = NWM * NWN, NT = NW * 64;
  static constexpr int WM = BM / NWM, WN = BN / NWN, TM = WM / 32, TN = WN / 32;
  static constexpr int STAGE_BYTES = (BM + BN) * KBYTES;
  static constexpr int SMEM = STAGES * STAGE_BYTES;
  static constexpr int PIECES = (BM + BN) / 8;          // 1 KiB DMA pieces (8 rows x 128 B) per stage
  static constexpr int PPW = PIECES / NW;               // pieces per wave
  static constexpr int WAVES_PER_SIMD = NW / 4 * (SMEM <= 80 * 1024 ? 2 : 1);
  static_assert(PIECES % NW == 0, "pieces must divide evenly over the waves");
  static_assert(WM % 32 == 0 && WN % 32 == 0, "wave tile must be a multiple of 32x32");
  static_assert(32 * BN * 4 <= SMEM, "epilogue slab must fit");
};

// ---- epilogue shared by the tile kernels, 32 output rows per pass through LDS (the staging buffers are free after
// the last barrier): the owning waves write their accumulators as 16-byte chunks into a [32][BN/4] fp32 slab
// (chunk ^= row, conflict spreading; `put(pass, slab)`), then all threads re-read it row-wise so that consecutive lanes
// cover consecutive columns of one row: global stores are whole row segments and the fused tail is emitted once.
// position of 16-byte chunk `chunk` of slab row `ml` (conflict spreading): XOR for power-of-two rows, rotation otherwise
template <int CPR>
__device__ __forceinline__ int slab_pos(int ml, int chunk) {
  if constexpr ((CPR & (CPR - 1)) == 0) return ml * CPR + (chunk ^ ml);
  else return ml * CPR + (chunk + ml) % CPR;
}

// cycle accounting of the diagnostic build (s_memtime sums per wave; compiled out unless PROF)
struct EpiProf { unsigned long long t_bar1 = 0, t_put = 0, t_bar2 = 0, t_rw = 0; };
#define EPI_STAMP(var) do { if constexpr (PROF) { const unsigned long long n__ = __builtin_amdgcn_s_memtime(); prof->var += n__ - last__; last__ = n__; } } while (0)

// NTAIL: 0 = every tail, 1 = the row-layout tails only (generic / SwiGLU): instantiations launched without split-K and without the fused QKV tail
// do not carry those branches (and their registers) at all - see gemm_pp_kernel's TAIL
template <typename T, bool SWIGLU, int BM, int BN, int NT, bool DRAIN, bool PROF = false, int NTAIL = 0, typename PutFn>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& p, float* slab, int tid, int tile_m, int tile_n, int tiles_m, int zo, int zi,
                                              long c_z, int split, PutFn put, EpiProf* prof = nullptr) {
  // tid / tile coordinates behind an opaque asm: the epilogue's per-thread address arithmetic is then computed HERE, after the K loop, instead
  // of being hoisted above it and kept in registers through it (the 256 x 192 split3 instantiation spilled 30 VGPRs for that)
  asm volatile("" : "+v"(tid));
  asm volatile("" : "+s"(tile_m), "+s"(tile_n));
  unsigned long long last__ = 0;
  if constexpr (PROF) last__ = __builtin_amdgcn_s_memtime();
  T* C = (T*)p.C + c_z;
  T* C2 = (T*)p.C2 + c_z;
  constexpr int CPR = BN / 4;               // chunks per slab row
  constexpr int NPASS = BM / 32;
  // DRAIN: the slab reuses the staging buffers, so every LDS-DMA of the main loop must have landed first
  if constexpr (DRAIN) __builtin_amdgcn_s_waitcnt(0);
  TailCols tcols;   // per-column tail operands of this thread's chunk (generic branch), read in the first pass
  // generic branch: every thread keeps ONE 4-column chunk for the whole tile (threads beyond the last full row of chunks idle), so the
  // per-column operands are read once per tile; rows advance by NT / CPR per step.  The residual rows of pass p + 1 are requested
  // while pass p is processed: requested inside their own pass, the first load of every pass sat exposed behind the pass's two
  // barriers (the DAC's 1-tap conv + residual launches: four ~2 us round trips per 128-row tile of a kernel with three K steps).
  constexpr int RPS = NT / CPR, NTA = RPS * CPR, ITER = (32 + RPS - 1) / RPS;
  const int g_chunk = tid % CPR, g_r0 = tid / CPR;
  const int g_n0 = tile_n * BN + g_chunk * 4;
  const bool g_col_ok = tid < NTA && g_n0 < p.N;
  const bool generic = NTAIL >= 1 ? !SWIGLU : (!(p.ksplit > 1) && !SWIGLU && !p.qkv_mode);
  float rs_next[ITER][4];
#pragma unroll
  for (int k = 0; k < ITER; ++k)
#pragma unroll
    for (int i = 0; i < 4; ++i) rs_next[k][i] = 0.f;
  auto fetch_res = [&](int pass_) __attribute__((always_inline)) {
    const int mrow = tile_m * BM + pass_ * 32;
#pragma unroll
    for (int k = 0; k < ITER; ++k) {
      const int ml = g_r0 + k * RPS;
      if (g_col_ok && ml < 32 && mrow + ml < p.M) gemm_tail_res<T>(p, mrow + ml, g_n0, zo, zi, rs_next[k]);
    }
  };
  // conv forms: clamped, unconditional residual loads and the per-column operands read in front of the pass loop
  auto conv_fetch_res = [&](int pass_) __attribute__((always_inline)) {
    const int mrow = tile_m * BM + pass_ * 32;
    const int nc = g_n0 + 3 < p.N ? g_n0 : p.N - 4;
#pragma unroll
    for (int k = 0; k < ITER; ++k) {
      const int ml = g_r0 + k * RPS, mc = mrow + (ml < 32 ? ml : 31);
      Vec4<T>::unpack(*(const typename Vec4<T>::raw*)((const T*)p.res + zo * p.res_bo + zi * p.res_bi + (long)(mc < p.M ? mc : p.M - 1) * p.ldres + nc), rs_next[k]);
    }
  };
  if constexpr (NTAIL >= 2) {
    const int nc = g_n0 + 3 < p.N ? g_n0 : p.N - 4;
    const int nv = p.vec_mod ? nc % p.vec_mod : nc;
    const long bo = zo * p.bias_bo + zi * p.bias_bi;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      tcols.bias[i] = vec_at<T>(p.bias, bo + nv + i);
      tcols.al[i] = vec_at<T>(p.snake_alpha, nv + i);
      tcols.ial[i] = 1.0f / (tcols.al[i] + 1e-9f);
    }
    if constexpr (((NTAIL - 2) & 1) != 0) {
      conv_fetch_res(0);
      // as many (dummy) stores behind the first pass's residual request as every later pass's request has behind it: the wait-count pass merges
      // the loop entry with the back edge, and with nothing behind the request on the entry path it waits for the request as the YOUNGEST
      // operations in flight - on the back edge that includes the previous pass's stores (vmcnt(3..0) instead of vmcnt(7..4), read off the ISA)
      constexpr int NST = 1 + (((NTAIL - 2) & 2) != 0);
#pragma unroll
      for (int k = 0; k < ITER * NST; ++k) *(f32x4*)((char*)p.sink + (k >> 1) * 32768 + tid * 32 + 16 * (k & 1)) = f32x4{0.f, 0.f, 0.f, 0.f};      // distinct addresses: no dead-store elimination
    }
  } else if (generic && p.res) fetch_res(0);
#pragma unroll 1
  for (int pass = 0; pass < NPASS; ++pass) {
    // raw barriers + lgkmcnt only: __syncthreads() would also wait (vmcnt) for the previous pass's global stores
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    EPI_STAMP(t_bar1);
    put(pass, slab);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    EPI_STAMP(t_put);
    __builtin_amdgcn_s_barrier();
    EPI_STAMP(t_bar2);
    const int mrow0 = tile_m * BM + pass * 32;
    if (NTAIL == 0 && p.ksplit > 1) {
      // raw fp32 partial sums to the split-K workspace [split][Mpad][Npad]; the reduce kernel applies the tail
      float* ws = (float*)p.ws + ((long)split * tiles_m * BM + mrow0) * (long)p.Npad + (long)tile_n * BN;
#pragma unroll
      for (int idx = tid; idx < 32 * CPR; idx += NT) {
        const int ml = idx / CPR, chunk = idx % CPR;
        if (tile_n * BN + chunk * 4 < p.Npad)
          *(f32x4*)(ws + (long)ml * p.Npad + chunk * 4) = *(const f32x4*)(slab + slab_pos<CPR>(ml, chunk) * 4);
      }
    } else if constexpr (SWIGLU) {
      // packed rows [16 x w1 | 16 x w3] per 32: chunk pair (b*8 + q, b*8 + 4 + q) -> output columns b*16 + 4q ..
#pragma unroll
      for (int idx = tid; idx < 32 * (CPR / 2); idx += NT) {
        const int ml = idx / (CPR / 2), pc = idx % (CPR / 2);
        const int b = pc >> 2, q = pc & 3;
        const int m = mrow0 + ml;
        const int j0 = tile_n * (BN / 2) + b * 16 + q * 4;
        if (m >= p.M || j0 >= (p.N >> 1)) continue;
        const f32x4 a4 = *(const f32x4*)(slab + slab_pos<CPR>(ml, b * 8 + q) * 4);
        const f32x4 b4 = *(const f32x4*)(slab + slab_pos<CPR>(ml, b * 8 + 4 + q) * 4);
        swiglu_tail<T>(p, m, j0, a4, b4, C);
      }
    } else if (NTAIL == 0 && p.qkv_mode) {
      const int D = p.qkv_D;
      const int sec = (tile_n * BN) / D;           // the whole tile lies in one of q | k | v | gate (D % BN == 0)
      if (sec == 2) {
        // V section: transposed store, 8 consecutive tokens of one d per thread
#pragma unroll
        for (int idx = tid; idx < BN * 4; idx += NT) {
          const int col = idx % BN, sg = idx / BN;
          const int m = mrow0 + 8 * sg;
          if (m >= p.M) continue;
          float v8[8];
#pragma unroll
          for (int r = 0; r < 8; ++r) {
            const int rw = 8 * sg + r;
            v8[r] = Num<T>::rnd(slab[slab_pos<CPR>(rw, col >> 2) * 4 + (col & 3)]);
          }
          const int hd = tile_n * BN + col - 2 * D;        // h * 128 + d
          const int b = m / p.qkv_S, sidx = m - b * p.qkv_S;
          T* dst = (T*)p.vt + (long)b * p.vt_row_stride + (long)hd * p.vt_ld + sidx;
          if ((p.qkv_S & 7) == 0 && m + 7 < p.M) {
            *(typename Vec4<T>::raw*)dst = Vec4<T>::pack(v8);
            *(typename Vec4<T>::raw*)(dst + 4) = Vec4<T>::pack(v8 + 4);
          } else {
            for (int r = 0; r < 8 && m + r < p.M; ++r) {
              const int bb = (m + r) / p.qkv_S, ss = (m + r) - bb * p.qkv_S;
              ((T*)p.vt)[(long)bb * p.vt_row_stride + (long)hd * p.vt_ld + ss] = Num<T>::st(v8[r]);
            }
          }
        }
      } else {
#pragma unroll
        for (int idx = tid; idx < 32 * CPR; idx += NT) {
          const int ml = idx / CPR, chunk = idx % CPR;
          const int m = mrow0 + ml;
          const int n0 = tile_n * BN + chunk * 4;
          const f32x4 a4 = *(const f32x4*)(slab + slab_pos<CPR>(ml, chunk) * 4);
          float y[4] = {Num<T>::rnd(a4[0]), Num<T>::rnd(a4[1]), Num<T>::rnd(a4[2]), Num<T>::rnd(a4[3])};
          if (sec < 2) {
            // 32 consecutive lanes hold the 128 columns of one (token, head): half-wave reduction of the squares
            float ss = y[0] * y[0] + y[1] * y[1] + y[2] * y[2] + y[3] * y[3];
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
            const float rs = rsqrtf(ss / 128.0f + p.qk_eps);
            const int nd = n0 - sec * D;                   // h * 128 + d
#pragma unroll
            for (int i = 0; i < 4; ++i)
              y[i] = Num<T>::rnd(__fmul_rn(__fmul_rn(y[i], rs), vec_at<T>(p.qk_w, (long)sec * D + nd + i)));
            if ((nd >> 7) < p.rope_heads) {
              const int pos = p.pos0 + m % p.qkv_S;
              const float2* rp = (const float2*)p.rope + (long)pos * 64 + ((nd & 127) >> 1);
#pragma unroll
              for (int pr = 0; pr < 2; ++pr) {
                const float2 cs = rp[pr];
                const float a = y[2 * pr], bq = y[2 * pr + 1];
                y[2 * pr] = __fsub_rn(__fmul_rn(a, cs.x), __fmul_rn(bq, cs.y));
                y[2 * pr + 1] = __fadd_rn(__fmul_rn(a, cs.y), __fmul_rn(bq, cs.x));
              }
            }
          }
          if (sec == 3 && p.qkv_gate_act) {
#pragma unroll
            for (int i = 0; i < 4; ++i) y[i] = Num<T>::rnd(sigmoid_fast(y[i]));
          }
          if (m < p.M && n0 < p.N) *(typename Vec4<T>::raw*)(C + (long)m * p.ldc + n0) = Vec4<T>::pack(y);
        }
      }
    } else if constexpr (NTAIL >= 2) {
      // ---- branch-free conv tail (the Fish S1-DAC decoder's Conv1d / ConvTranspose1d launches): y = acc + bias [+ residual],
      // main output [optional], second output snake(y).  NTAIL = 2 + RES + 2 MAIN.  Nothing that touches memory sits behind a branch:
      // the residual rows are loaded from clamped addresses, the stores of threads without a row (beyond the tile's last full row of
      // chunks, beyond M) go to p.sink.  The generic tail below keeps its loads and stores behind the feature tests and per-thread predicates;
      // hipcc's wait-count pass then cannot count what is in flight at the joins and waits for EVERYTHING (vmcnt(0)) in front of the
      // residual of every 32-row pass, i.e. for the round trip of the previous pass's stores, BM / 32 times per tile (read off the ISA).
      constexpr bool F_RES = ((NTAIL - 2) & 1) != 0, F_MAIN = ((NTAIL - 2) & 2) != 0;
      float rs[ITER][4];
#pragma unroll
      for (int k = 0; k < ITER; ++k)
#pragma unroll
        for (int i = 0; i < 4; ++i) rs[k][i] = rs_next[k][i];
      if constexpr (F_RES) { if (pass + 1 < NPASS) conv_fetch_res(pass + 1); }
      char* const sink = (char*)p.sink + tid * 32;
#pragma unroll
      for (int k = 0; k < ITER; ++k) {
        const int ml = g_r0 + k * RPS, m = mrow0 + ml;
        const bool ok = g_col_ok && ml < 32 && m < p.M;
        const f32x4 a4 = *(const f32x4*)(slab + slab_pos<CPR>(ml < 32 ? ml : 31, g_chunk) * 4);
        float y[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) y[i] = Num<T>::rnd(Num<T>::rnd(a4[i] + tcols.bias[i]));
        if constexpr (F_RES) {
#pragma unroll
          for (int i = 0; i < 4; ++i) y[i] = Num<T>::rnd(y[i] + rs[k][i]);
        }
        typedef Vec4<T> V;
        if constexpr (F_MAIN) *(typename V::raw*)(ok ? (char*)(C + (long)m * p.ldc + g_n0) : sink) = V::pack(y);
        float sn4[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float sn = p.snake_fast ? sin_fast(tcols.al[i] * y[i]) : sinf(tcols.al[i] * y[i]);
          sn4[i] = Num<T>::rnd(y[i] + tcols.ial[i] * (sn * sn));
        }
        *(typename V::raw*)(ok ? (char*)(C2 + (long)m * p.ldc + g_n0) : sink + 16) = V::pack(sn4);
      }
    } else {
      const int chunk = g_chunk, r0 = g_r0, n0 = g_n0;
      const bool col_ok = g_col_ok;
      if (pass == 0 && col_ok) gemm_tail_cols<T>(p, n0, zo, zi, tcols);
      float rs[ITER][4];
#pragma unroll
      for (int k = 0; k < ITER; ++k)
#pragma unroll
        for (int i = 0; i < 4; ++i) rs[k][i] = rs_next[k][i];
      if (p.res && pass + 1 < NPASS) fetch_res(pass + 1);
#pragma unroll
      for (int k = 0; k < ITER; ++k) {
        const int ml = r0 + k * RPS, m = mrow0 + ml;
        if (col_ok && ml < 32 && m < p.M) {
          const f32x4 a4 = *(const f32x4*)(slab + slab_pos<CPR>(ml, chunk) * 4);
          float y[4] = {a4[0], a4[1], a4[2], a4[3]};
          gemm_tail_apply<T>(p, m, n0, y, tcols, rs[k], C, C2);
        }
      }
    }
    EPI_STAMP(t_rw);
  }
}

// SPLIT3 (fp32 only): every fp32 operand x is split in registers into bf16 hi = bf16(x), lo = bf16(x - hi) and the
// product is evaluated as hi*hi + hi*lo + lo*hi on the bf16 MFMA with fp32 accumulation (relative error ~2^-16 per
// product, ~1e-5 on sums): 16/3 times the fp32-MFMA rate.  Used for the Fish S1-DAC decoder whose 1e-4 waveform
// tolerance leaves three orders of magnitude of head room; the parity-mode DiT keeps the exact fp32 MFMA.
template <typename T, bool SWIGLU, typename CF, bool SPLIT3, int NTAIL = 0>
__global__ void __launch_bounds__(CF::NT, CF::WAVES_PER_SIMD) gemm_nt_kernel(const GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int KE = KBYTES / (int)sizeof(T);
  constexpr int BM = CF::BM, BN = CF::BN, TM = CF::TM, TN = CF::TN, PPW = CF::PPW, STAGES = CF::STAGES;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  // column tiles cover N (the packed SwiGLU rows: Npad); a tile that lies wholly in the row padding of W is not launched
  const int tiles_m = (p.M + BM - 1) / BM, tiles_n = ((SWIGLU ? p.Npad : p.N) + BN - 1) / BN;
  const int nwg = tiles_m * tiles_n;
  int bid = blockIdx.x;
  {  // bijective XCD remap: workgroups that share an XCD (bid % 8) get a contiguous run of tiles
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tile_m = bid % tiles_m, tile_n = bid / tiles_m;
  const int z = blockIdx.y, zo = z / p.nbi, zi = z - zo * p.nbi;
  const long a_z = zo * p.a_bo + zi * p.a_bi, w_z = zo * p.w_bo + zi * p.w_bi, c_z = zo * p.c_bo + zi * p.c_bi;

  // ---- K range of this workgroup (split-K over blockIdx.z)
  const int kb_per_tap = p.K / KE;
  const int nk_total = kb_per_tap * p.taps;
  const int ks = p.ksplit > 1 ? p.ksplit : 1;
  const int split = blockIdx.z;
  const int it0 = (int)((long)nk_total * split / ks), it1 = (int)((long)nk_total * (split + 1) / ks);
  const int nk = it1 - it0;

  // ---- DMA sources: wave w owns pieces w*PPW .. w*PPW+PPW-1 of the combined [A rows | W rows] stage image
  const char* src[PPW];
  bool is_a[PPW];
#pragma unroll
  for (int i = 0; i < PPW; ++i) {
    const int piece = wid * PPW + i;
    const int r = piece * 8 + (lane >> 3);            // row inside the stage image
    const int chunk = (lane & 7) ^ ((r >> 1) & 7);    // source-side swizzle (BM is a multiple of 16)
    is_a[i] = piece * 8 < BM;
    if (is_a[i]) {
      int gm = tile_m * BM + r;
      gm = gm < p.M ? gm : p.M - 1;
      src[i] = (const char*)p.A + ((long)(gm + p.tap_base) * p.lda + a_z) * (long)sizeof(T) + chunk * 16;
    } else {
      int gn = tile_n * BN + (r - BM);
      gn = gn < p.Npad ? gn : p.Npad - 1;
      src[i] = (const char*)p.W + ((long)gn * p.ldw + w_z) * (long)sizeof(T) + chunk * 16;
    }
  }
  const long a_tap_bytes = ((long)p.tap_shift * p.lda - (long)p.K) * (long)sizeof(T);  // extra A step at a tap boundary
  int kb = it0 % kb_per_tap;
  long a_off = (long)(it0 / kb_per_tap) * p.tap_shift * p.lda * (long)sizeof(T) + (long)kb * KBYTES;
  long w_off = (long)it0 * KBYTES;

  // DMA of one stage: sources are resolved first (next_src), the PPW pieces are then issued one at a time between the
  // MFMAs of the current tile (issue_piece): a burst of 8 LDS-DMA instructions blocks the wave for ~500-1800 cycles.
  const char* nsrc[PPW];
  char* ndst = nullptr;
  auto next_src = [&](int slot) {
    ndst = smem + slot * CF::STAGE_BYTES + wid * (PPW * 1024);
#pragma unroll
    for (int i = 0; i < PPW; ++i) nsrc[i] = src[i] + (is_a[i] ? a_off : w_off);
    a_off += KBYTES; w_off += KBYTES;
    if (++kb == kb_per_tap) { kb = 0; a_off += a_tap_bytes; }
  };
  auto stage = [&](int slot) {
    next_src(slot);
#pragma unroll
    for (int i = 0; i < PPW; ++i) glds16(nsrc[i], ndst + i * 1024);
  };

  // ---- fragment read addresses
  const int wm = wid / CF::NWN, wn = wid % CF::NWN;
  const int fr = lane & 31, fh = lane >> 5;
  const int sw = (lane >> 1) & 7;   // == ((row >> 1) & 7): all row bases are multiples of 16
  const int a_row0 = (wm * CF::WM + fr) * KBYTES;
  const int w_row0 = (BM + wn * CF::WN + fr) * KBYTES;

  f32x16 acc[TN][TM];
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  // ---- pipeline: STAGES-1 tiles in flight
  int issued = 0;
#pragma unroll
  for (int s = 0; s < STAGES - 1; ++s)
    if (issued < nk) { stage(s); ++issued; }

  constexpr int NMF = (Num<T>::is_bf16 ? 4 : 16) * TN * TM;   // MFMAs per K step and wave
  // the K loop exists twice in the SPLIT3 kernels: with W split in registers, and with W pre-split by the host side into [32 hi | 32 lo]
  // bf16 per 32-float block (GemmArgs.w_presplit: static weights; the split of a weight fragment is 24 VALU instructions that every
  // wave of every row tile repeated - the 128x96 tile spent 96 VALU per 9 MFMAs on it)
  auto kloop = [&](auto wpre_) __attribute__((always_inline)) {
  constexpr bool WPRE = decltype(wpre_)::value;
  for (int it = 0; it < nk; ++it) {
    // tile `it` must have landed: allow (tiles still in flight - 1) * PPW younger DMA pieces to stay outstanding
    const int younger = issued - it - 1;
    if (STAGES >= 4 && younger >= 2) wait_vmcnt<2 * PPW>();
    else if (STAGES >= 3 && younger >= 1) wait_vmcnt<PPW>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();   // every wave's pieces of tile `it` are in LDS; everyone finished reading tile it-1
    const bool more = issued < nk;   // wave-uniform
    if (more) { next_src((it + STAGES - 1) % STAGES); ++issued; }

    const char* sa = smem + (it % STAGES) * CF::STAGE_BYTES;
    int mf = 0, piece = 0;
    // after every MFMA: issue the next DMA piece when due, then pin the order (sched_barrier) so that the pieces stay
    // spread over the tile instead of being hoisted into one burst
    auto after_mfma = [&]() {
      ++mf;
      // all pieces go out in the first 1/DMA_SPREAD_DEN of the tile (more than one per call where a wave has more
      // pieces than MFMA groups, e.g. the 128x96 split3 tile: 7 pieces, 6 groups)
#pragma unroll
      for (int due = 0; due < PPW; ++due) {
        if (piece < PPW && mf * PPW * DMA_SPREAD_DEN >= (piece + 1) * NMF) {
          if (more) glds16(nsrc[piece], ndst + piece * 1024);
          ++piece;
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    };
    if constexpr (Num<T>::is_bf16) {
      bf16x8 wf[2][TN], af[2][TM];
      auto load = [&](int kk, int b) {
        const int c = ((2 * kk + fh) ^ sw) << 4;
#pragma unroll
        for (int t = 0; t < TN; ++t) wf[b][t] = *(const bf16x8*)(sa + w_row0 + t * 32 * KBYTES + c);
#pragma unroll
        for (int t = 0; t < TM; ++t) af[b][t] = *(const bf16x8*)(sa + a_row0 + t * 32 * KBYTES + c);
      };
      load(0, 0);
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        if (kk < 3) load(kk + 1, (kk + 1) & 1);
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
          for (int tm = 0; tm < TM; ++tm) {
            acc[tn][tm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[kk & 1][tn], af[kk & 1][tm], acc[tn][tm], 0, 0, 0);
            after_mfma();
          }
      }
    } else if constexpr (SPLIT3) {
      // 32 floats per row and K step = two bf16 MFMA k-steps of 16; lane (fr, fh) needs floats 16kk + 8fh .. +7
      auto split = [&](const char* rowp, int kk, bf16x8& hi, bf16x8& lo) {
        const f32x4 x0 = *(const f32x4*)(rowp + (((4 * kk + 2 * fh) ^ sw) << 4));
        const f32x4 x1 = *(const f32x4*)(rowp + (((4 * kk + 2 * fh + 1) ^ sw) << 4));
        f32x8 x;
#pragma unroll
        for (int i = 0; i < 4; ++i) { x[i] = x0[i]; x[4 + i] = x1[i]; }
        const hbf16x8 h = __builtin_convertvector(x, hbf16x8);
        const f32x8 hf = __builtin_convertvector(h, f32x8);
        const hbf16x8 l = __builtin_convertvector(x - hf, hbf16x8);
        hi = __builtin_bit_cast(bf16x8, h);
        lo = __builtin_bit_cast(bf16x8, l);
      };
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        bf16x8 whi[TN], wlo[TN], ahi[TM], alo[TM];
#pragma unroll
        for (int t = 0; t < TN; ++t) {
          if constexpr (WPRE) {      // the block holds bf16 hi[0..31] | lo[0..31]: chunks 0-3 | 4-7 of 16 bytes, same swizzle
            const char* rowp = sa + w_row0 + t * 32 * KBYTES;
            whi[t] = *(const bf16x8*)(rowp + (((2 * kk + fh) ^ sw) << 4));
            wlo[t] = *(const bf16x8*)(rowp + (((4 + 2 * kk + fh) ^ sw) << 4));
          } else split(sa + w_row0 + t * 32 * KBYTES, kk, whi[t], wlo[t]);
        }
#pragma unroll
        for (int t = 0; t < TM; ++t) split(sa + a_row0 + t * 32 * KBYTES, kk, ahi[t], alo[t]);
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
          for (int tm = 0; tm < TM; ++tm) {
            acc[tn][tm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wlo[tn], ahi[tm], acc[tn][tm], 0, 0, 0);
            acc[tn][tm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(whi[tn], alo[tm], acc[tn][tm], 0, 0, 0);
            acc[tn][tm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(whi[tn], ahi[tm], acc[tn][tm], 0, 0, 0);
            mf += 7; after_mfma();   // 3 bf16 MFMAs stand for 8 fp32 ones in the DMA-piece schedule (NMF counts fp32 MFMAs)
          }
      }
    } else {
      f32x4 wf[2][TN], af[2][TM];
      auto load = [&](int cc, int b) {
        const int c = (cc ^ sw) << 4;
#pragma unroll
        for (int t = 0; t < TN; ++t) wf[b][t] = *(const f32x4*)(sa + w_row0 + t * 32 * KBYTES + c);
#pragma unroll
        for (int t = 0; t < TM; ++t) af[b][t] = *(const f32x4*)(sa + a_row0 + t * 32 * KBYTES + c);
      };
      load(0, 0);
#pragma unroll
      for (int cc = 0; cc < 8; ++cc) {
        if (cc < 7) load(cc + 1, (cc + 1) & 1);
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
          for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) {
              acc[tn][tm] = __builtin_amdgcn_mfma_f32_32x32x2f32(fh ? wf[cc & 1][tn][2 * s + 1] : wf[cc & 1][tn][2 * s],
                                                                 fh ? af[cc & 1][tm][2 * s + 1] : af[cc & 1][tm][2 * s],
                                                                 acc[tn][tm], 0, 0, 0);
              after_mfma();
            }
      }
    }
  }
  };
  if constexpr (SPLIT3) { if (p.w_presplit) kloop(std::true_type{}); else kloop(std::false_type{}); }
  else kloop(std::false_type{});

  // ---- epilogue (gemm_epilogue): this wave's accumulators of one 32-row pass go into the LDS slab
  const int fr_ = fr, fh_ = fh;
  gemm_epilogue<T, SWIGLU, BM, BN, CF::NT, true, false, NTAIL>(p, (float*)smem, tid, tile_m, tile_n, tiles_m, zo, zi, c_z, split, [&](int pass, float* slab) {
    constexpr int CPR = BN / 4;
    if (wm == pass / TM) {
      const int tm_sel = pass % TM;
      auto put = [&](const f32x16& a, int tn) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int chunk = wn * (CF::WN / 4) + tn * 8 + 2 * g + fh_;
          f32x4 v;
          v[0] = a[4 * g]; v[1] = a[4 * g + 1]; v[2] = a[4 * g + 2]; v[3] = a[4 * g + 3];
          *(f32x4*)(slab + slab_pos<CPR>(fr_, chunk) * 4) = v;
        }
      };
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) {
        // static register indices only (a runtime-indexed accumulator array would live in scratch)
        if (tm_sel == 0) put(acc[tn][0], tn);
        if constexpr (TM > 1) { if (tm_sel == 1) put(acc[tn][1], tn); }
        if constexpr (TM > 2) { if (tm_sel == 2) put(acc[tn][2], tn); }
        if constexpr (TM > 3) { if (tm_sel == 3) put(acc[tn][3], tn); }
      }
    }
  });
}

template <typename T, bool SW, typename CF, bool SPLIT3, int NTAIL>
hipError_t launch_cfg_tail(const GemmArgs& g, hipStream_t st);

template <typename T, bool SW, typename CF, bool SPLIT3 = false>
hipError_t launch_cfg(const GemmArgs& g, hipStream_t st) {
  static const bool split = getenv("ECHO_NT_TAILS") ? atoi(getenv("ECHO_NT_TAILS")) != 0 : true;      // 0: the all-tails instantiation everywhere (A/B aid)
  if constexpr (SPLIT3 && !SW && (CF::BN == 96 || CF::BN == 192)) {      // the DAC's conv tiles: branch-free conv tails (gemm_epilogue, NTAIL >= 2)
    static const bool conv = getenv("ECHO_NT_CONV_TAILS") ? atoi(getenv("ECHO_NT_CONV_TAILS")) != 0 : true;
    if (split && conv && g.ksplit <= 1 && !g.qkv_mode && g.sink && g.bias && g.snake_alpha && g.C2 && !g.colscale && g.act == 0 && g.div == 0.0f &&
        g.acc_scale == 1.0f && g.N >= 4) {
      if (g.res) return g.store_main ? launch_cfg_tail<T, SW, CF, SPLIT3, 5>(g, st) : launch_cfg_tail<T, SW, CF, SPLIT3, 3>(g, st);
      return g.store_main ? launch_cfg_tail<T, SW, CF, SPLIT3, 4>(g, st) : launch_cfg_tail<T, SW, CF, SPLIT3, 2>(g, st);
    }
  }
  if (split && g.ksplit <= 1 && !g.qkv_mode) return launch_cfg_tail<T, SW, CF, SPLIT3, 1>(g, st);
  return launch_cfg_tail<T, SW, CF, SPLIT3, 0>(g, st);
}

template <typename T, bool SW, typename CF, bool SPLIT3, int NTAIL>
hipError_t launch_cfg_tail(const GemmArgs& g, hipStream_t st) {
  static std::atomic<unsigned long long> prepared{0};
  auto kern = gemm_nt_kernel<T, SW, CF, SPLIT3, NTAIL>;
  if (hipError_t e = ensure_dyn_lds((const void*)kern, CF::SMEM, prepared); e != hipSuccess) return e;
  const int tiles_m = (g.M + CF::BM - 1) / CF::BM, tiles_n = ((SW ? g.Npad : g.N) + CF::BN - 1) / CF::BN;
  const int ks = g.ksplit > 1 ? g.ksplit : 1;
  dim3 grid(tiles_m * tiles_n, g.nbatch, ks);
  hipLaunchKernelGGL(kern, grid, dim3(CF::NT), CF::SMEM, st, g);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess || ks == 1) return e;
  const int Mpad = tiles_m * CF::BM;
  const long items = (long)g.M * (SW ? g.Npad / 8 : g.Npad / 4);
  hipLaunchKernelGGL((splitk_reduce_kernel<T, SW>), dim3((unsigned)((items + 255) / 256)), dim3(256), 0, st, g, Mpad);
  return hipGetLastError();
}

typedef TileCfg<128, 128, 2, 2, 2> Cfg0;   // 4 waves, 64 KiB: two workgroups per CU
typedef TileCfg<128, 128, 2, 2, 4> Cfg1;   // 4 waves, 128 KiB, 3 tiles in flight: latency-bound small grids
typedef TileCfg<256, 256, 2, 4, 2> Cfg2;   // 8 waves (128x64 each), 128 KiB: lowest L2 traffic per FLOP
typedef TileCfg<256, 128, 4, 2, 3> Cfg3;   // 8 waves (64x64 each), 144 KiB, 2 tiles in flight
typedef TileCfg<128, 256, 2, 4, 3> Cfg4;   // 8 waves (64x64 each), 144 KiB, 2 tiles in flight
typedef TileCfg<128, 192, 2, 2, 2> Cfg6;   // 4 waves (64x96 each), 80 KiB: N = 192 / 384 without padding waste (DAC 192- and 384-channel convs)
typedef TileCfg<128, 96, 4, 1, 2> Cfg7;    // 4 waves (32x96 each), 56 KiB: N = 96 (DAC 96-channel convs)
typedef TileCfg<256, 192, 4, 2, 2> Cfg8;   // 8 waves (64x96 each), 112 KiB: less L2->LDS traffic per output for the long-M convs
typedef TileCfg<384, 96, 6, 1, 2> Cfg9;    // 6 waves (64x96 each), 120 KiB: N = 96, the weight tile amortised over 384 rows

template <typename T, bool SW>
hipError_t launch_sw(const GemmArgs& g, hipStream_t st) {
  if constexpr (!Num<T>::is_bf16) {
    if (g.cfg == 5 || g.cfg >= 100) return hipErrorInvalidValue;   // the ping-pong kernel (and its diagnostic builds) is bf16 only
    if (g.split3) {
      switch (g.cfg) {
        case 6: return launch_cfg<T, SW, Cfg6, true>(g, st);
        case 7: return launch_cfg<T, SW, Cfg7, true>(g, st);
        case 8: return launch_cfg<T, SW, Cfg8, true>(g, st);
        case 9: return launch_cfg<T, SW, Cfg9, true>(g, st);
        case 1: return launch_cfg<T, SW, Cfg1, true>(g, st);
        case 2: return launch_cfg<T, SW, Cfg2, true>(g, st);
        case 3: return launch_cfg<T, SW, Cfg3, true>(g, st);
        case 4: return launch_cfg<T, SW, Cfg4, true>(g, st);
        default: return launch_cfg<T, SW, Cfg0, true>(g, st);
      }
    }
  }
  if (g.cfg == 5 || g.cfg >= 100) {          // the persistent ping-pong kernel and its diagnostic builds live in gemm_pp.hip
    if constexpr (Num<T>::is_bf16) return launch_gemm_pp(g, st);
    else return hipErrorInvalidValue;
  }
  if (g.fp8 || g.c8) return hipErrorInvalidValue;   // fp8 operands / e4m3 output exist for the ping-pong kernel only
  if (g.cfg >= 6 && g.qkv_mode) return hipErrorInvalidValue;   // the fused QKV tail needs tiles that divide a section (BN | D)
  switch (g.cfg) {
    case 1: return launch_cfg<T, SW, Cfg1>(g, st);
    case 2: return launch_cfg<T, SW, Cfg2>(g, st);
    case 3: return launch_cfg<T, SW, Cfg3>(g, st);
    case 4: return launch_cfg<T, SW, Cfg4>(g, st);
    case 6: return launch_cfg<T, SW, Cfg6>(g, st);
    case 7: return launch_cfg<T, SW, Cfg7>(g, st);
    case 8: return launch_cfg<T, SW, Cfg8>(g, st);
    case 9: return launch_cfg<T, SW, Cfg9>(g, st);
    default: return launch_cfg<T, SW, Cfg0>(g, st);
  }
}

}  // namespace


template <typename T>
hipError_t launch_gemm_nt(const GemmArgs& g, hipStream_t st) {
  constexpr int KE = KBYTES / (int)sizeof(T);
  if (g.M <= 0 || g.N <= 0 || g.K <= 0 || g.K % KE != 0 || g.Npad % 128 != 0 || g.Npad < g.N || (g.N & 3) ||
      g.taps < 1 || g.nbatch < 1 || g.nbi < 1 || (g.lda % (16 / (int)sizeof(T))) || (g.ldw % (16 / (int)sizeof(T))) ||
      (g.ldc & 3) || g.cfg < 0 || (g.cfg >= gemm_num_cfgs() && (g.cfg < 101 || g.cfg > 111)))
    return hipErrorInvalidValue;
  if (g.qkv_mode && (g.ksplit > 1 || g.swiglu || g.nbatch != 1 || g.qkv_D % 256 || !g.vt || !g.qk_w || !g.rope || g.qkv_S < 1))
    return hipErrorInvalidValue;
  if (g.ksplit > 1) {
    if (g.nbatch != 1 || !g.ws || g.ksplit > (g.K / KE) * g.taps) return hipErrorInvalidValue;
    const long need = (long)g.ksplit * ((g.M + gemm_tile_m(g.cfg) - 1) / gemm_tile_m(g.cfg)) * gemm_tile_m(g.cfg) * g.Npad * 4;
    if (g.ws_bytes < need) return hipErrorInvalidValue;
  }
  return g.swiglu ? launch_sw<T, true>(g, st) : launch_sw<T, false>(g, st);
}
