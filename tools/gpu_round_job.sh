# Round-end measurement job (run on the GPU box through gpurun): PMC passes -> tests -> smoke -> bench line -> kernel trace.
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
V=${1:-v1}
O=gpurun_out/round_$V
mkdir -p $O
# 1. HBM-side traffic of the GEMM kernels: one counter per pass (MI355X_MICROARCH.md, HBM / rocprofv3), no tracing domains
PMC_CMD="python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-eager-baseline --no-roofline"
timeout -k 10 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o run -- $PMC_CMD > $O/pmc_fetch.log 2>&1
timeout -k 10 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o run -- $PMC_CMD > $O/pmc_write.log 2>&1
python3 tools/summarize_pmc.py $O/pmc_fetch $O/pmc_write profiles/r02_pmc_gemm.json "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace --output-format csv -- $PMC_CMD (two separate passes; default batch 24, 2 streams)"
cp profiles/r02_pmc_gemm.json $O/r02_pmc_gemm.json
rm -rf $O/pmc_fetch $O/pmc_write
# 1b. MFMA-pipe utilisation and effective clock per kernel family (one more counter pass, no tracing domains besides the kernel trace)
timeout -k 10 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_mfma -o run -- $PMC_CMD > $O/pmc_mfma.log 2>&1
python3 tools/summarize_pmc_mfma.py $O/pmc_mfma profiles/r02_pmc_mfma.json "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -- $PMC_CMD (default batch 24, 2 streams)"
cp profiles/r02_pmc_mfma.json $O/r02_pmc_mfma.json
rm -rf $O/pmc_mfma
# 2. tests + smoke
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1 || { tail -30 $O/gpu_tests.log; exit 1; }
tail -2 $O/gpu_tests.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 || { tail -20 $O/smoke.log; exit 1; }
tail -3 $O/smoke.log
# 3. the bench line (default flags, as the driver runs it)
ECHO_PROFILE_SHAPES=1 timeout -k 10 900 python bench.py > $O/bench.log 2> $O/bench.err
tail -n 1 $O/bench.log > $O/r02_bench_line_$V.json
grep "^\[echo\] shape" $O/bench.err > $O/r02_gemm_shapes_$V.txt || true
tail -c 300 $O/bench.log
# 3b. BASELINE config C5 (fp8 operands, 100 steps) on the same box
timeout -k 10 900 python bench.py --c5 --steps 2 --warmup 1 --no-cpu-baseline --no-eager-baseline > $O/bench_c5.log 2> $O/bench_c5.err
tail -n 1 $O/bench_c5.log > $O/r02_bench_line_c5_$V.json
head -c 200 $O/r02_bench_line_c5_$V.json; echo
# 4. kernel trace of the same command (shorter run, no CPU leg)
timeout -k 10 700 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o run -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-eager-baseline > $O/prof.log 2>&1
rm -f $O/prof/*kernel_trace.csv $O/prof/*/*kernel_trace.csv
python3 tools/summarize_prof.py $(ls $O/prof/*kernel_stats.csv $O/prof/*/*kernel_stats.csv 2>/dev/null | head -1) $O/r02_bench_kernel_stats_$V.csv "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-eager-baseline (batch 24, 2 streams)"
tail -n 1 $O/prof.log | head -c 600
