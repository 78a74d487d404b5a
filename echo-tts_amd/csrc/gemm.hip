// GEMM kernels of libechohip: C = epilogue(A · Wᵀ), both operands K-contiguous.
//
// * gemm_pp_kernel (plan cfg 5): bf16, persistent 256x256 "ping-pong" kernel on v_mfma_f32_16x16x32_bf16 — every large
//   EchoDiT linear (QKVG with its fused head-norm / RoPE / Vᵀ tail, wo, SwiGLU w1‖w3, w2).  Described at its definition.
// * gemm_nt_kernel (plans 0-4): bf16 (v_mfma_f32_32x32x16_bf16), exact fp32 (v_mfma_f32_32x32x2_f32) and fp32 "split3"
//   (three bf16 MFMAs per product).  One kernel serves the small / odd-shaped linears (reference: every nn.Linear in
//   model.py / autoencoder.py) and, through the `taps` loop, every causal Conv1d / ConvTranspose1d of the DAC in
//   channels-last layout (autoencoder.py:264-331).  Tile configurations (TileCfg): 128x128 / 256x128 / 128x256 / 256x256
//   output tiles, 4 or 8 waves, 2-4 LDS stages with a counted-vmcnt DMA pipeline; K-step = 128 bytes per row (64 bf16 /
//   32 fp32); optional split-K with fp32 partial slabs and a deterministic reduce + tail kernel.
//
// Common to both: A and W tiles are staged by direct global->LDS DMA (global_load_lds_dwordx4) into a lane-linear LDS
// image; the XOR swizzle is applied on the SOURCE address (chunk ^= (row>>1)&7) and again on the ds_read_b128 address,
// which makes every fragment read conflict-free (cdna_hip_programming.md §5.4 rule 21, §5.5 T2).  Operands are swapped
// into the MFMA (W rows -> MFMA rows, activation rows -> MFMA columns) so that each lane ends up with 4 consecutive
// output columns of one output row: the SwiGLU pair (w1, w3) and the RoPE pairs live in one lane.

#include "gemm_tile.h"

void gemm_args_init(GemmArgs* g) {
  memset(g, 0, sizeof(*g));
  g->taps = 1; g->nbatch = 1; g->nbi = 1; g->acc_scale = 1.0f; g->store_main = 1;
}

int gemm_tile_m(int cfg) { return cfg == 9 ? 384 : cfg == 2 || cfg == 3 || cfg == 5 || cfg == 8 || cfg >= 100 ? 256 : 128; }
int gemm_num_cfgs() { return 10; }

template hipError_t launch_gemm_nt<bf16_t>(const GemmArgs&, hipStream_t);


// In-place reformat of an fp32 weight matrix [rows][ld] (ld % 32 == 0) for GemmArgs.w_presplit: every aligned block of 32 floats
// becomes 32 bf16 hi = bf16(x) followed by 32 bf16 lo = bf16(x - hi) - the values the SPLIT3 kernels compute in registers.
__global__ void presplit_w_kernel(float* __restrict__ w, long nblocks) {
  const long b = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= nblocks) return;
  float* blk = w + b * 32;
  f32x8 x[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) x[i] = *(const f32x8*)(blk + 8 * i);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const hbf16x8 h = __builtin_convertvector(x[i], hbf16x8);
    const f32x8 hf = __builtin_convertvector(h, f32x8);
    const hbf16x8 l = __builtin_convertvector(x[i] - hf, hbf16x8);
    *(bf16x8*)((char*)blk + 16 * i) = __builtin_bit_cast(bf16x8, h);
    *(bf16x8*)((char*)blk + 64 + 16 * i) = __builtin_bit_cast(bf16x8, l);
  }
}
hipError_t launch_presplit_w(float* w, long rows, long ld, hipStream_t st) {
  if (!w || rows < 1 || ld < 32 || (ld & 31)) return hipErrorInvalidValue;
  const long nb = rows * (ld / 32);
  hipLaunchKernelGGL(presplit_w_kernel, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, st, w, nb);
  return hipGetLastError();
}

