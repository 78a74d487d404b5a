// What the power envelope allows: every SIMD of the chip issues back-to-back v_mfma_f32_32x32x16_bf16 from registers (no LDS, no memory)
// for about a second per operand pattern; prints the rate.  Run next to `rocm-smi -P -c` (tools/power_probe.py samples the same way).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 mfma_power.hip -o mfma_power && ./mfma_power
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

template <int WAVES>
__global__ void __launch_bounds__(WAVES * 64) spin(const u32x4* __restrict__ src, float* out, int iters) {
  const int lane = threadIdx.x;
  u32x4 r[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) r[i] = src[(blockIdx.x * 8 + i) * (WAVES * 64) + lane];
  f32x16 acc[4] = {};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, r[2 * i]), __builtin_bit_cast(bf16x8, r[2 * i + 1]), acc[i], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, r[(2 * i + 3) & 7]), __builtin_bit_cast(bf16x8, r[(2 * i + 6) & 7]), acc[i], 0, 0, 0);
    }
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) s += acc[i][j];
  if (s == 12345.678f) out[0] = s;
}

static unsigned short bf(float x) { unsigned u; __builtin_memcpy(&u, &x, 4); return (unsigned short)((u + 0x7fff + ((u >> 16) & 1)) >> 16); }

template <int WAVES>
static void run(const char* label, int mode) {
  const int G = 256 * (WAVES > 4 ? 1 : 1), NT = WAVES * 64;
  const size_t n16 = (size_t)G * 8 * NT * 8;
  std::vector<unsigned short> h(n16);
  srand(1);
  for (size_t i = 0; i < n16; ++i) {
    float v = mode == 0 ? 0.f : mode == 1 ? 0.5f : ((rand() & 0xffff) / 32768.0f - 1.0f) * 0.02f;
    h[i] = bf(v);
  }
  u32x4* d; float* o;
  hipMalloc(&d, n16 * 2); hipMalloc(&o, 4);
  hipMemcpy(d, h.data(), n16 * 2, hipMemcpyHostToDevice);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const int iters = getenv("ITERS") ? atoi(getenv("ITERS")) : 4000000;
  hipLaunchKernelGGL(spin<WAVES>, dim3(G), dim3(NT), 0, 0, d, o, 2000);
  hipDeviceSynchronize();
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(a);
    hipLaunchKernelGGL(spin<WAVES>, dim3(G), dim3(NT), 0, 0, d, o, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double flops = (double)G * WAVES * iters * 8 * 32768.0;
    printf("%-10s %d waves/CU rep %d: %8.1f ms  %7.0f TFLOP/s  (%.0f MHz-equivalent at 1024 flop/clk/SIMD)\n", label, WAVES, rep, ms, flops / ms / 1e9,
           flops / ms / 1e3 / (256.0 * 4 * 1024));
    fflush(stdout);
  }
  hipFree(d); hipFree(o);
}

int main() {
  run<4>("zeros", 0);
  run<4>("constant", 1);
  run<4>("random", 2);
  run<8>("random", 2);
  return 0;
}
