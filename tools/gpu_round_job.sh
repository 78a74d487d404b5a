# Round-end measurement job (run on the GPU box through gpurun): tests -> smoke -> bench line -> kernel trace -> PMC passes.
# Usage: bash tools/gpu_round_job.sh <tag> [round]   (outputs under gpurun_out/round_<tag>/, summaries copied to profiles/ by hand)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
V=${1:-v1}
R=${2:-r03}
O=gpurun_out/round_$V
mkdir -p $O
# 1. tests (as the driver runs them: -x) + smoke
timeout -k 10 1000 python -m pytest tests -x -q -m gpu -p no:cacheprovider -rP --durations=8 > $O/gpu_tests.log 2>&1
echo "pytest rc=$?" >> $O/gpu_tests.log
grep -h "^FAILED\|^ERROR\| passed\| failed\|pytest rc" $O/gpu_tests.log | cut -c1-300
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
echo "smoke rc=$?" >> $O/smoke.log; tail -4 $O/smoke.log
# 2. the bench line (default flags, as the driver runs it) with the per-shape table
ECHO_PROFILE_SHAPES=1 timeout -k 10 1100 python bench.py > $O/bench.log 2> $O/bench.err
tail -n 1 $O/bench.log > $O/${R}_bench_line_$V.json
grep "^\[echo\] shape" $O/bench.err > $O/${R}_gemm_shapes_$V.txt || true
head -c 400 $O/${R}_bench_line_$V.json; echo
# 3. kernel trace of the same command (shorter run, no baselines / legs)
timeout -k 10 700 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o run -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-eager-baseline --no-c5 --no-legs > $O/prof.log 2>&1
rm -f $O/prof/*kernel_trace.csv $O/prof/*/*kernel_trace.csv
python3 tools/summarize_prof.py $(ls $O/prof/*kernel_stats.csv $O/prof/*/*kernel_stats.csv 2>/dev/null | head -1) $O/${R}_bench_kernel_stats_$V.csv "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-eager-baseline --no-c5 --no-legs (batch 24, 2 streams)"
head -8 $O/${R}_bench_kernel_stats_$V.csv
# 4. HBM-side traffic of the GEMM kernels: one counter per pass (MI355X_MICROARCH.md, HBM / rocprofv3), no tracing domains
PMC_CMD="python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-eager-baseline --no-roofline"
timeout -k 10 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o run -- $PMC_CMD > $O/pmc_fetch.log 2>&1
timeout -k 10 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o run -- $PMC_CMD > $O/pmc_write.log 2>&1
python3 tools/summarize_pmc.py $O/pmc_fetch $O/pmc_write $O/${R}_pmc_gemm.json "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace --output-format csv -- $PMC_CMD (two separate passes; default batch 24, 2 streams)"
rm -rf $O/pmc_fetch $O/pmc_write
# 4b. MFMA-pipe utilisation and effective clock per kernel family (one more counter pass)
timeout -k 10 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_mfma -o run -- $PMC_CMD > $O/pmc_mfma.log 2>&1
python3 tools/summarize_pmc_mfma.py $O/pmc_mfma $O/${R}_pmc_mfma.json "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -- $PMC_CMD (default batch 24, 2 streams)"
rm -rf $O/pmc_mfma
ls $O
