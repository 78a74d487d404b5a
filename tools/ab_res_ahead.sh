#!/bin/bash
# A/B of the residual prefetch depth of gemm_pp's column scale + residual tail on one box: rebuilds gemm_pp.o with -DPP_RES_AHEAD=<n> and relinks
set -e
cd "$(dirname "$0")/.."
D=echo-tts_amd
ls $D/build/*.o > /dev/null
for a in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DPP_RES_AHEAD=$a -c $D/csrc/gemm_pp.hip -o $D/build/gemm_pp.o 2> /dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $D/libechohip.so $D/build/gemm_pp.o $D/build/gemm.o $D/build/gemm_f32.o $D/build/attention.o $D/build/elementwise.o $D/build/dac.o $D/build/postproc.o $D/build/engine.o
  echo "== PP_RES_AHEAD=$a"
  python tools/prof_fp8.py 15360 2>&1 | grep "^M="
  python tools/prof_fp8.py 46080 2>&1 | grep "^M="
done
