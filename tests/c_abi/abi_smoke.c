/* A caller of libechohip written in plain C: proves that include/echo_hip.h is a C header (no C++, no torch types)
 * and that the library works without Python.  Built by tests (gcc, linked against libechohip.so + libamdhip64):
 *   not-gpu: compile + link only;  gpu: run -> one fp32 GEMM on small integers must be exact, a bad descriptor must
 *   return a status instead of launching.  Exit code 0 = pass. */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "echo_hip.h"

#define CHECK_HIP(x) do { hipError_t e__ = (x); if (e__ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e__)); return 2; } } while (0)

int main(void) {
  if (echo_abi_version() != ECHO_ABI_VERSION) { fprintf(stderr, "ABI version mismatch\n"); return 1; }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) { fprintf(stderr, "no HIP device\n"); return 3; }
  CHECK_HIP(hipSetDevice(0));

  enum { M = 200, N = 128, K = 96 };               /* ragged M, one 128-column tile, three 32-float K steps */
  float *hA = malloc(sizeof(float) * M * K), *hW = malloc(sizeof(float) * N * K), *hC = malloc(sizeof(float) * M * N);
  unsigned s = 12345u;
  for (int i = 0; i < M * K; ++i) { s = s * 1664525u + 1013904223u; hA[i] = (float)((int)(s >> 28) - 8); }
  for (int i = 0; i < N * K; ++i) { s = s * 1664525u + 1013904223u; hW[i] = (float)((int)(s >> 29) - 4); }
  float *dA, *dW, *dC;
  CHECK_HIP(hipMalloc((void**)&dA, sizeof(float) * M * K));
  CHECK_HIP(hipMalloc((void**)&dW, sizeof(float) * N * K));
  CHECK_HIP(hipMalloc((void**)&dC, sizeof(float) * M * N));
  CHECK_HIP(hipMemcpy(dA, hA, sizeof(float) * M * K, hipMemcpyHostToDevice));
  CHECK_HIP(hipMemcpy(dW, hW, sizeof(float) * N * K, hipMemcpyHostToDevice));
  CHECK_HIP(hipMemset(dC, 0, sizeof(float) * M * N));

  echo_gemm_desc d;
  memset(&d, 0, sizeof d);
  d.A = dA; d.W = dW; d.C = dC;
  d.M = M; d.N = N; d.K = K; d.Npad = N;
  d.lda = K; d.ldw = K; d.ldc = N;
  d.taps = 1; d.nbatch = 1; d.nbi = 1;
  d.acc_scale = 1.0f; d.store_main = 1; d.ksplit = 1;
  if (echo_op_gemm(ECHO_F32, &d, NULL) != 0) { fprintf(stderr, "echo_op_gemm: %s\n", echo_last_error(NULL)); return 4; }
  CHECK_HIP(hipDeviceSynchronize());
  CHECK_HIP(hipMemcpy(hC, dC, sizeof(float) * M * N, hipMemcpyDeviceToHost));
  for (int m = 0; m < M; ++m)
    for (int n = 0; n < N; ++n) {
      float ref = 0.0f;                            /* small integers: every product and partial sum is exact in fp32 */
      for (int k = 0; k < K; ++k) ref += hA[m * K + k] * hW[n * K + k];
      if (hC[m * N + n] != ref) { fprintf(stderr, "C[%d][%d] = %g, expected %g\n", m, n, hC[m * N + n], ref); return 5; }
    }

  d.K = 7;                                         /* not a multiple of the K step: must be refused, not launched */
  if (echo_op_gemm(ECHO_F32, &d, NULL) == 0) { fprintf(stderr, "invalid descriptor was accepted\n"); return 6; }
  echo_ctx* ctx = NULL;
  if (echo_ctx_create(NULL, 0, &ctx) == 0 || ctx != NULL) { fprintf(stderr, "NULL config was accepted\n"); return 7; }
  if (!echo_last_error(NULL) || !echo_last_error(NULL)[0]) { fprintf(stderr, "no error text\n"); return 8; }

  hipFree(dA); hipFree(dW); hipFree(dC);
  free(hA); free(hW); free(hC);
  printf("abi_smoke: ok\n");
  return 0;
}
