// gemm_nt: C = epilogue(A · Wᵀ) on MFMA, bf16 (v_mfma_f32_32x32x16_bf16) or exact fp32
// (v_mfma_f32_32x32x2_f32).  One kernel serves every Linear on the hot path (reference: every
// nn.Linear in model.py / autoencoder.py) and, through the `taps` loop, every causal Conv1d /
// ConvTranspose1d of the DAC decoder in channels-last layout (autoencoder.py:264-331).
//
// Tile: 128 (m) x 128 (n) per 256-thread workgroup, K-step = 128 bytes per row (64 bf16 / 32 fp32).
// Both operands are K-contiguous, so A and W tiles are staged the same way: direct global->LDS
// DMA (global_load_lds_dwordx4), lane-linear LDS image, XOR swizzle applied on the SOURCE address
// (chunk ^= (row>>1)&7) and again on the ds_read_b128 address, which makes every fragment read
// conflict-free (cdna_hip_programming.md §5.4 rule 21, §5.5 T2).  Double-buffered, one barrier per
// K-step.  Operands are swapped into the MFMA (W rows -> MFMA rows, activation rows -> MFMA
// columns) so that each lane ends up with 4 consecutive output columns of one output row: the
// epilogue stores 8 B (bf16) / 16 B (fp32) vectors and the SwiGLU pair (w1, w3) lives in one lane.
#include "common.h"

void gemm_args_init(GemmArgs* g) {
  memset(g, 0, sizeof(*g));
  g->taps = 1; g->nbatch = 1; g->nbi = 1; g->acc_scale = 1.0f; g->store_main = 1;
}

namespace {

constexpr int BM = 128, BN = 128, KBYTES = 128;
constexpr int TILE_BYTES = BM * KBYTES;           // 16 KiB per operand per stage
constexpr int STAGE_BYTES = 2 * TILE_BYTES;       // A + W
constexpr int SMEM_BYTES = 2 * STAGE_BYTES;       // double buffered: 64 KiB

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __forceinline__ void glds16(const char* src, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)lds_wave_base, 16, 0, 0);
}

template <typename T>
__device__ __forceinline__ float vec_at(const void* p, long i) { return Num<T>::ld(((const T*)p)[i]); }

template <typename T, bool SWIGLU>
__global__ void __launch_bounds__(256, 2) gemm_nt_kernel(const GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int KE = KBYTES / (int)sizeof(T);
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int tiles_m = (p.M + BM - 1) / BM, tiles_n = p.Npad / BN;
  const int nwg = tiles_m * tiles_n;
  int bid = blockIdx.x;
  {  // bijective XCD remap: workgroups that share an XCD (bid % 8) get a contiguous run of tiles
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tile_m = bid % tiles_m, tile_n = bid / tiles_m;
  const int z = blockIdx.y, zo = z / p.nbi, zi = z - zo * p.nbi;
  const long a_z = zo * p.a_bo + zi * p.a_bi, w_z = zo * p.w_bo + zi * p.w_bi, c_z = zo * p.c_bo + zi * p.c_bi;

  // ---- staging addresses: wave w issues DMA pieces 4w..4w+3, each 8 rows x 128 B
  const char* asrc[4];
  const char* wsrc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = (wid * 4 + i) * 8 + (lane >> 3);
    const int chunk = (lane & 7) ^ ((r >> 1) & 7);
    int gm = tile_m * BM + r;
    gm = gm < p.M ? gm : p.M - 1;
    asrc[i] = (const char*)p.A + ((long)(gm + p.tap_base) * p.lda + a_z) * (long)sizeof(T) + chunk * 16;
    const int gn = tile_n * BN + r;
    wsrc[i] = (const char*)p.W + ((long)gn * p.ldw + w_z) * (long)sizeof(T) + chunk * 16;
  }
  const int kb_per_tap = p.K / KE;
  const int nk = kb_per_tap * p.taps;
  const long a_tap_bytes = ((long)p.tap_shift * p.lda - (long)p.K) * (long)sizeof(T);  // extra step at a tap boundary

  auto stage = [&](int buf, long a_off, long w_off) {
    char* base = smem + buf * STAGE_BYTES + wid * 4096;
#pragma unroll
    for (int i = 0; i < 4; ++i) glds16(asrc[i] + a_off, base + i * 1024);
#pragma unroll
    for (int i = 0; i < 4; ++i) glds16(wsrc[i] + w_off, base + TILE_BYTES + i * 1024);
  };

  // ---- fragment read addresses (bytes inside a tile)
  const int wn = wid & 1, wm = wid >> 1;
  const int fr = lane & 31, fh = lane >> 5;
  const int sw = (lane >> 1) & 7;  // == ((row >> 1) & 7) because the row bases are multiples of 16
  const int a_row0 = (wm * 64 + fr) * KBYTES, w_row0 = (wn * 64 + fr) * KBYTES;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  long a_off = 0, w_off = 0;
  int kb = 0;
  stage(0, a_off, w_off);
  for (int it = 0; it < nk; ++it) {
    __syncthreads();  // (vmcnt(0) + barrier): stage `it` has landed, everyone is done with the other buffer
    if (it + 1 < nk) {
      a_off += KBYTES; w_off += KBYTES;
      if (++kb == kb_per_tap) { kb = 0; a_off += a_tap_bytes; }
      stage((it + 1) & 1, a_off, w_off);
    }
    const char* sa = smem + (it & 1) * STAGE_BYTES;
    const char* swt = sa + TILE_BYTES;
    if constexpr (Num<T>::is_bf16) {
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const int c = ((2 * kk + fh) ^ sw) << 4;
        bf16x8 wf[2], af[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          wf[t] = *(const bf16x8*)(swt + w_row0 + t * 32 * KBYTES + c);
          af[t] = *(const bf16x8*)(sa + a_row0 + t * 32 * KBYTES + c);
        }
#pragma unroll
        for (int tn = 0; tn < 2; ++tn)
#pragma unroll
          for (int tm = 0; tm < 2; ++tm)
            acc[tn][tm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[tn], af[tm], acc[tn][tm], 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int cc = 0; cc < 8; ++cc) {
        const int c = (cc ^ sw) << 4;
        f32x4 wf[2], af[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          wf[t] = *(const f32x4*)(swt + w_row0 + t * 32 * KBYTES + c);
          af[t] = *(const f32x4*)(sa + a_row0 + t * 32 * KBYTES + c);
        }
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
          for (int tn = 0; tn < 2; ++tn)
#pragma unroll
            for (int tm = 0; tm < 2; ++tm)
              acc[tn][tm] = __builtin_amdgcn_mfma_f32_32x32x2f32(fh ? wf[tn][2 * s + 1] : wf[tn][2 * s],
                                                                 fh ? af[tm][2 * s + 1] : af[tm][2 * s],
                                                                 acc[tn][tm], 0, 0, 0);
      }
    }
  }

  // ---- epilogue.  The accumulators go through LDS (the staging buffers are free now): each lane
  // writes its 4-column runs as 16-byte chunks into a [128][32 chunks] fp32 image (chunk ^= row & 31
  // against bank conflicts); the tile is then re-read row-wise so that 32 consecutive lanes cover one
  // output row: global stores are whole 256/512-byte row segments and the fused tail below is
  // emitted once (a runtime loop) instead of 16 times per lane.
  __syncthreads();
  {
    float* sc = (float*)smem;
#pragma unroll
    for (int tm = 0; tm < 2; ++tm) {
      const int m = wm * 64 + tm * 32 + fr;
#pragma unroll
      for (int tn = 0; tn < 2; ++tn)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int chunk = wn * 16 + tn * 8 + 2 * g + fh;
          f32x4 v;
          v[0] = acc[tn][tm][4 * g]; v[1] = acc[tn][tm][4 * g + 1]; v[2] = acc[tn][tm][4 * g + 2]; v[3] = acc[tn][tm][4 * g + 3];
          *(f32x4*)(sc + (m * 32 + (chunk ^ (m & 31))) * 4) = v;
        }
    }
  }
  __syncthreads();
  typedef Vec4<T> V;
  const float* sc = (const float*)smem;
  T* C = (T*)p.C + c_z;
  T* C2 = (T*)p.C2 + c_z;
  const int vm = p.vec_mod;
  if constexpr (SWIGLU) {
    // packed rows [16 x w1 | 16 x w3] per 32: chunk pair (b*8 + q, b*8 + 4 + q) -> output columns b*16 + 4q ..
#pragma unroll 1
    for (int j = 0; j < 8; ++j) {
      const int idx = j * 256 + tid;
      const int ml = idx >> 4, pc = idx & 15;
      const int b = pc >> 2, q = pc & 3;
      const int m = tile_m * BM + ml;
      const int j0 = tile_n * (BN / 2) + b * 16 + q * 4;
      const f32x4 a4 = *(const f32x4*)(sc + (ml * 32 + ((b * 8 + q) ^ (ml & 31))) * 4);
      const f32x4 b4 = *(const f32x4*)(sc + (ml * 32 + ((b * 8 + 4 + q) ^ (ml & 31))) * 4);
      float o[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float a = Num<T>::rnd(a4[i]);
        const float bb = Num<T>::rnd(b4[i]);
        o[i] = Num<T>::rnd(Num<T>::rnd(silu_f(a)) * bb);
      }
      if (m < p.M && j0 < (p.N >> 1)) *(typename V::raw*)(C + (long)m * p.ldc + j0) = V::pack(o);
    }
  } else {
#pragma unroll 1
    for (int j = 0; j < 16; ++j) {
      const int idx = j * 256 + tid;
      const int ml = idx >> 5, chunk = idx & 31;
      const int m = tile_m * BM + ml;
      const int n0 = tile_n * BN + chunk * 4;
      if (m >= p.M || n0 >= p.N) continue;
      const f32x4 a4 = *(const f32x4*)(sc + (ml * 32 + (chunk ^ (ml & 31))) * 4);
      float y[4] = {a4[0], a4[1], a4[2], a4[3]};
      if (p.acc_scale != 1.0f) {
#pragma unroll
        for (int i = 0; i < 4; ++i) y[i] *= p.acc_scale;
      }
      const int nv = vm ? n0 % vm : n0;
      if (p.bias) {
        const long bo = zo * p.bias_bo + zi * p.bias_bi;
#pragma unroll
        for (int i = 0; i < 4; ++i) y[i] += vec_at<T>(p.bias, bo + nv + i);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) y[i] = Num<T>::rnd(y[i]);
      if (p.div != 0.0f) {
#pragma unroll
        for (int i = 0; i < 4; ++i) y[i] = Num<T>::rnd(y[i] / p.div);
      }
      if (p.act == 1) {
#pragma unroll
        for (int i = 0; i < 4; ++i) y[i] = Num<T>::rnd(silu_f(y[i]));
      } else if (p.act == 2) {
#pragma unroll
        for (int i = 0; i < 4; ++i) y[i] = Num<T>::rnd(gelu_erf_f(y[i]));
      }
      if (p.colscale) {
#pragma unroll
        for (int i = 0; i < 4; ++i) y[i] = Num<T>::rnd(y[i] * vec_at<T>(p.colscale, nv + i));
      }
      if (p.res) {
        float r[4];
        V::unpack(*(const typename V::raw*)((const T*)p.res + zo * p.res_bo + zi * p.res_bi + (long)m * p.ldres + n0), r);
#pragma unroll
        for (int i = 0; i < 4; ++i) y[i] = Num<T>::rnd(y[i] + r[i]);
      }
      if (p.store_main) *(typename V::raw*)(C + (long)m * p.ldc + n0) = V::pack(y);
      if (p.snake_alpha) {
        float sn4[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float al = vec_at<T>(p.snake_alpha, nv + i);
          const float sn = sinf(al * y[i]);
          sn4[i] = Num<T>::rnd(y[i] + (1.0f / (al + 1e-9f)) * (sn * sn));
        }
        *(typename V::raw*)(C2 + (long)m * p.ldc + n0) = V::pack(sn4);
      }
    }
  }
}

template <typename T, bool SW>
hipError_t launch_impl(const GemmArgs& g, hipStream_t st) {
  static bool attr_set = false;
  auto kern = gemm_nt_kernel<T, SW>;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  const int tiles_m = (g.M + BM - 1) / BM, tiles_n = g.Npad / BN;
  dim3 grid(tiles_m * tiles_n, g.nbatch, 1);
  hipLaunchKernelGGL(kern, grid, dim3(256), SMEM_BYTES, st, g);
  return hipGetLastError();
}

}  // namespace

template <typename T>
hipError_t launch_gemm_nt(const GemmArgs& g, hipStream_t st) {
  constexpr int KE = KBYTES / (int)sizeof(T);
  if (g.M <= 0 || g.N <= 0 || g.K <= 0 || g.K % KE != 0 || g.Npad % BN != 0 || g.Npad < g.N || (g.N & 3) ||
      g.taps < 1 || g.nbatch < 1 || g.nbi < 1 || (g.lda % (16 / (int)sizeof(T))) || (g.ldw % (16 / (int)sizeof(T))) ||
      (g.ldc & 3))
    return hipErrorInvalidValue;
  return g.swiglu ? launch_impl<T, true>(g, st) : launch_impl<T, false>(g, st);
}
template hipError_t launch_gemm_nt<bf16_t>(const GemmArgs&, hipStream_t);
template hipError_t launch_gemm_nt<float>(const GemmArgs&, hipStream_t);
