# Round 3, first GPU job: the whole -m gpu suite (no -x: every failure is wanted), then the single-request profile.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r3a
mkdir -p $O
timeout -k 10 1000 python -m pytest tests -q -m gpu -p no:cacheprovider --durations=15 > $O/gpu_tests.log 2>&1
echo "pytest rc=$?" >> $O/gpu_tests.log
tail -5 $O/gpu_tests.log
# single request (C2 proper): per-shape table + phases, then a kernel trace for the gap analysis
ECHO_PROFILE_SHAPES=1 timeout -k 10 600 python bench.py --batch 1 --concurrency 1 --steps 3 --warmup 1 --no-cpu-baseline --no-eager-baseline --no-c5 --no-legs > $O/single.log 2> $O/single.err
tail -c 1500 $O/single.log
grep "^\[echo\] shape" $O/single.err > $O/single_shapes.txt
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof1 -o run -- python3 bench.py --batch 1 --concurrency 1 --steps 2 --warmup 1 --no-cpu-baseline --no-eager-baseline --no-roofline > $O/prof1.log 2>&1
T=$(ls $O/prof1/*kernel_trace.csv $O/prof1/*/*kernel_trace.csv 2>/dev/null | head -1)
python3 tools/trace_gaps.py $T 0.6 > $O/single_gaps.txt 2>&1
cat $O/single_gaps.txt
python3 tools/summarize_prof.py $(ls $O/prof1/*kernel_stats.csv $O/prof1/*/*kernel_stats.csv 2>/dev/null | head -1) $O/single_kernel_stats.csv "rocprofv3 --kernel-trace --stats -- python3 bench.py --batch 1 --concurrency 1 --steps 2 --warmup 1 (single request)"
rm -f $O/prof1/*kernel_trace.csv $O/prof1/*/*kernel_trace.csv
