// Micro-benchmark (gfx950): cycles per pair of v_mfma_f32_32x32x16_bf16 with vector-ALU fillers between them, one wave per SIMD.
// Which MFMA operand placements let v_exp_f32 / v_add_f32 issue under a running MFMA?  Build: hipcc --offload-arch=gfx950 -O3 mfma_fill.hip
// -o mfma_fill; run on the GPU box.  Prints cycles per pair (two MFMAs + the fillers of the variant) measured with s_memtime by wave 0.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <type_traits>
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) short bf16x8;
template <int OFF> __device__ __forceinline__ bf16x8 rd(unsigned a) { bf16x8 v; asm volatile("ds_read_b128 %0, %1 offset:%2" : "=a"(v) : "v"(a), "n"(OFF)); return v; }
__device__ __forceinline__ void pair_pv_fill(f32x16& o0, f32x16& o1, float& x0, float& x1, float& r0, float& r1, const bf16x8& a, const bf16x8& p0, const bf16x8& p1) {
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %6, %7, %0\n\tv_exp_f32 %2, %2\n\tv_add_f32 %4, %4, %3\n\t"
               "v_mfma_f32_32x32x16_bf16 %1, %6, %8, %1\n\tv_exp_f32 %3, %3\n\tv_add_f32 %5, %5, %2"
               : "+a"(o0), "+a"(o1), "+v"(x0), "+v"(x1), "+v"(r0), "+v"(r1) : "a"(a), "v"(p0), "v"(p1));
}
__device__ __forceinline__ void pair_pv(f32x16& o0, f32x16& o1, const bf16x8& a, const bf16x8& p0, const bf16x8& p1) {
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n\tv_mfma_f32_32x32x16_bf16 %1, %2, %4, %1" : "+a"(o0), "+a"(o1) : "a"(a), "v"(p0), "v"(p1));
}
__device__ __forceinline__ void pair_s_fill(f32x16& s0, f32x16& s1, float& x0, float& x1, float& r0, float& r1, const bf16x8& a, const bf16x8& qa, const bf16x8& qb) {
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %6, %7, %0\n\tv_exp_f32 %2, %2\n\tv_add_f32 %4, %4, %3\n\t"
               "v_mfma_f32_32x32x16_bf16 %1, %6, %8, %1\n\tv_exp_f32 %3, %3\n\tv_add_f32 %5, %5, %2"
               : "+v"(s0), "+v"(s1), "+v"(x0), "+v"(x1), "+v"(r0), "+v"(r1) : "a"(a), "a"(qa), "a"(qb));
}
__device__ __forceinline__ void one_pv(f32x16& o, const bf16x8& a, const bf16x8& p) { asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(o) : "a"(a), "v"(p)); }
__device__ __forceinline__ void fill2(float& x0, float& x1, float& r0, float& r1) { asm volatile("v_exp_f32 %0, %0\n\tv_add_f32 %2, %2, %1\n\tv_exp_f32 %1, %1\n\tv_add_f32 %3, %3, %0" : "+v"(x0), "+v"(x1), "+v"(r0), "+v"(r1)); }
__device__ __forceinline__ void add4(float& x0, float& x1, float& r0, float& r1) { asm volatile("v_add_f32 %0, %0, %1\n\tv_add_f32 %1, %1, %2\n\tv_add_f32 %2, %2, %3\n\tv_add_f32 %3, %3, %0" : "+v"(x0), "+v"(x1), "+v"(r0), "+v"(r1)); }
__device__ __forceinline__ void use_a(const bf16x8& a) { asm volatile("" ::"a"(a)); }
__device__ __forceinline__ void wait8() { asm volatile("s_waitcnt lgkmcnt(8)"); }
template <int I = 0, typename F> __device__ __forceinline__ void sfor16(F&& f) { if constexpr (I < 16) { f(std::integral_constant<int, I>{}); sfor16<I + 1>(f); } }

#define PAIR_PV(FILL)                                                                                     \
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n\t" FILL                                        \
               "v_mfma_f32_32x32x16_bf16 %1, %2, %4, %1\n\t" FILL                                        \
               : "+a"(o0), "+a"(o1), "+v"(x0), "+v"(x1), "+v"(r0), "+v"(r1) : "a"(va), "v"(p0), "v"(p1));
// operand numbering differs per variant: write each variant out explicitly instead

template <int V>
__global__ void __launch_bounds__(256, 1) bench(unsigned long long* out, int iters) {
  f32x16 o0 = {0}, o1 = {0}, s0 = {0}, s1 = {0};
  bf16x8 va = {1, 2, 3, 4, 5, 6, 7, 8}, qa = {1, 1, 1, 1, 1, 1, 1, 1}, qb = {2, 2, 2, 2, 2, 2, 2, 2}, p0 = {3, 3, 3, 3, 3, 3, 3, 3}, p1 = {4, 4, 4, 4, 4, 4, 4, 4};
  float x0 = -1.0f + threadIdx.x * 1e-3f, x1 = -2.0f, r0 = 0.f, r1 = 0.f;
  asm volatile("" : "+a"(va), "+a"(qa), "+a"(qb));
  unsigned long long t0, t1;
  __shared__ __attribute__((aligned(16))) char lds[65536];
  const unsigned la = (unsigned)(unsigned long)(__attribute__((address_space(3))) char*)lds + threadIdx.x % 64 * 16;
  bf16x8 w[16];
  if constexpr (V >= 8) {
    sfor16([&](auto uc) __attribute__((always_inline)) {
      constexpr int u = decltype(uc)::value;
      if constexpr (u < 8) w[u] = rd<1024 * u>(la);
    });
    asm volatile("s_waitcnt lgkmcnt(0)");
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int i = 0; i < iters; ++i) {
    if constexpr (V >= 8) {
      sfor16([&](auto uc) __attribute__((always_inline)) {
        constexpr int u = decltype(uc)::value;
        w[(u + 8) % 16] = rd<1024 * ((u + 8) % 16)>(la);
        wait8();
        if constexpr (V == 8) pair_pv_fill(o0, o1, x0, x1, r0, r1, w[u], p0, p1);
        else if constexpr (V == 9) pair_pv(o0, o1, w[u], p0, p1);
        else if constexpr (V == 10) pair_s_fill(s0, s1, x0, x1, r0, r1, w[u], qa, qb);
        else if constexpr (V == 12) { fill2(x0, x1, r0, r1); use_a(w[u]); }
        else if constexpr (V == 13) use_a(w[u]);
        else if constexpr (V == 14) { pair_pv(o0, o1, w[u], p0, p1); add4(x0, x1, r0, r1); }
        else if constexpr (V == 11) {     // fillers split into separate statements with sched barriers, as the kernel has them
          one_pv(o0, w[u], p0);
          __builtin_amdgcn_sched_barrier(0);
          x0 = __builtin_amdgcn_exp2f(x0); r0 += x1;
          __builtin_amdgcn_sched_barrier(0);
          one_pv(o1, w[u], p1);
          __builtin_amdgcn_sched_barrier(0);
          x1 = __builtin_amdgcn_exp2f(x1); r1 += x0;
          __builtin_amdgcn_sched_barrier(0);
        }
      });
      continue;
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      if constexpr (V == 0) {        // PV type (C/D AGPR, A AGPR, B VGPR), no fillers
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n\tv_mfma_f32_32x32x16_bf16 %1, %2, %4, %1" : "+a"(o0), "+a"(o1) : "a"(va), "v"(p0), "v"(p1));
      } else if constexpr (V == 1) { // PV type + exp, add after each MFMA
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %6, %7, %0\n\tv_exp_f32 %2, %2\n\tv_add_f32 %4, %4, %3\n\t"
                     "v_mfma_f32_32x32x16_bf16 %1, %6, %8, %1\n\tv_exp_f32 %3, %3\n\tv_add_f32 %5, %5, %2"
                     : "+a"(o0), "+a"(o1), "+v"(x0), "+v"(x1), "+v"(r0), "+v"(r1) : "a"(va), "v"(p0), "v"(p1));
      } else if constexpr (V == 2) { // S type (C/D VGPR, A AGPR, B AGPR), no fillers
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n\tv_mfma_f32_32x32x16_bf16 %1, %2, %4, %1" : "+v"(s0), "+v"(s1) : "a"(va), "a"(qa), "a"(qb));
      } else if constexpr (V == 3) { // S type + exp, add
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %6, %7, %0\n\tv_exp_f32 %2, %2\n\tv_add_f32 %4, %4, %3\n\t"
                     "v_mfma_f32_32x32x16_bf16 %1, %6, %8, %1\n\tv_exp_f32 %3, %3\n\tv_add_f32 %5, %5, %2"
                     : "+v"(s0), "+v"(s1), "+v"(x0), "+v"(x1), "+v"(r0), "+v"(r1) : "a"(va), "a"(qa), "a"(qb));
      } else if constexpr (V == 4) { // fillers only
        asm volatile("v_exp_f32 %0, %0\n\tv_add_f32 %2, %2, %1\n\tv_exp_f32 %1, %1\n\tv_add_f32 %3, %3, %0" : "+v"(x0), "+v"(x1), "+v"(r0), "+v"(r1));
      } else if constexpr (V == 5) { // PV type + 2 x (exp, add) after each MFMA
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %6, %7, %0\n\tv_exp_f32 %2, %2\n\tv_add_f32 %4, %4, %3\n\tv_exp_f32 %3, %3\n\tv_add_f32 %5, %5, %2\n\t"
                     "v_mfma_f32_32x32x16_bf16 %1, %6, %8, %1\n\tv_exp_f32 %2, %2\n\tv_add_f32 %4, %4, %3\n\tv_exp_f32 %3, %3\n\tv_add_f32 %5, %5, %2"
                     : "+a"(o0), "+a"(o1), "+v"(x0), "+v"(x1), "+v"(r0), "+v"(r1) : "a"(va), "v"(p0), "v"(p1));
      } else if constexpr (V == 6) { // PV type + 4 v_add after each MFMA (no transcendental)
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %6, %7, %0\n\tv_add_f32 %2, %2, %3\n\tv_add_f32 %4, %4, %3\n\tv_add_f32 %3, %3, %5\n\tv_add_f32 %5, %5, %2\n\t"
                     "v_mfma_f32_32x32x16_bf16 %1, %6, %8, %1\n\tv_add_f32 %2, %2, %3\n\tv_add_f32 %4, %4, %3\n\tv_add_f32 %3, %3, %5\n\tv_add_f32 %5, %5, %2"
                     : "+a"(o0), "+a"(o1), "+v"(x0), "+v"(x1), "+v"(r0), "+v"(r1) : "a"(va), "v"(p0), "v"(p1));
      } else if constexpr (V == 7) { // PV type with B from the accumulator file too + exp, add
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %6, %7, %0\n\tv_exp_f32 %2, %2\n\tv_add_f32 %4, %4, %3\n\t"
                     "v_mfma_f32_32x32x16_bf16 %1, %6, %8, %1\n\tv_exp_f32 %3, %3\n\tv_add_f32 %5, %5, %2"
                     : "+a"(o0), "+a"(o1), "+v"(x0), "+v"(x1), "+v"(r0), "+v"(r1) : "a"(va), "a"(qa), "a"(qb));
      }
    }
  }
  asm volatile("s_nop 15\n\ts_nop 15\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  if constexpr (V >= 8) asm volatile("" ::"a"(w[0]), "a"(w[1]), "a"(w[2]), "a"(w[3]), "a"(w[4]), "a"(w[5]), "a"(w[6]), "a"(w[7]));
  float acc = x0 + x1 + r0 + r1;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc += o0[i] + o1[i] + s0[i] + s1[i];
  if (threadIdx.x == 0) { out[blockIdx.x * 2] = t1 - t0; out[blockIdx.x * 2 + 1] = (unsigned long long)acc; }
}

template <int V> void run(const char* name, unsigned long long* d) {
  const int iters = 2000;
  hipLaunchKernelGGL(bench<V>, dim3(256), dim3(256), 0, 0, d, iters);
  hipLaunchKernelGGL(bench<V>, dim3(256), dim3(256), 0, 0, d, iters);
  hipDeviceSynchronize();
  unsigned long long h[2];
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  printf("%-58s %7.1f cycles per pair\n", name, (double)h[0] / (iters * 16.0));
}
int main() {
  unsigned long long* d;
  hipMalloc(&d, 4096 * 2 * 8);
  run<0>("PV type (C/D acc file, A acc, B vgpr), bare", d);
  run<1>("PV type + (v_exp, v_add) behind each MFMA", d);
  run<5>("PV type + 2 x (v_exp, v_add) behind each MFMA", d);
  run<6>("PV type + 4 x v_add behind each MFMA", d);
  run<7>("PV type with B from the acc file + (v_exp, v_add)", d);
  run<2>("S type (C/D vgpr, A acc, B acc), bare", d);
  run<3>("S type + (v_exp, v_add) behind each MFMA", d);
  run<4>("fillers only: 2 x (v_exp, v_add)", d);
  run<9>("PV type, A from a 16-slot LDS-read window (8 ahead), bare", d);
  run<8>("PV type, LDS-read window + (v_exp, v_add) behind each MFMA", d);
  run<10>("S type, LDS-read window + (v_exp, v_add) behind each MFMA", d);
  run<11>("PV type, window, fillers as compiler code between sched barriers", d);
  run<12>("window reads + fillers 2 x (v_exp, v_add), no MFMA", d);
  run<13>("window reads only (ds_read_b128 + lgkmcnt(8) per pair)", d);
  run<14>("PV pair, then 4 v_add, window reads", d);
  return 0;
}
