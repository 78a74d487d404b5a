// fp32 (exact v_mfma_f32_32x32x2_f32) and split-3 (three bf16 MFMAs per product) instantiations of the tile kernels: the fp32 parity engine
// and the Fish S1-DAC.  See gemm_tile.h / gemm.hip.
#include "gemm_tile.h"

template hipError_t launch_gemm_nt<float>(const GemmArgs&, hipStream_t);
