cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/small
mkdir -p $O
timeout -k 10 600 python -m pytest tests -q -m gpu -p no:cacheprovider -rP -k "teeth or load_audio or c4_blockwise" > $O/tests.log 2>&1
echo "pytest rc=$?" >> $O/tests.log
grep -h "^FAILED\|^ERROR\| passed\| failed\|pytest rc\|negated\|full depth, t\|^E  " $O/tests.log | cut -c1-600
timeout -k 10 600 rocprofv3 --kernel-trace --output-format csv -d $O/prof1 -o run -- python3 bench.py --batch 1 --concurrency 1 --steps 3 --warmup 1 --no-cpu-baseline --no-eager-baseline --no-roofline > $O/prof1.log 2>&1
T=$(ls $O/prof1/*kernel_trace.csv $O/prof1/*/*kernel_trace.csv 2>/dev/null | head -1)
python3 tools/trace_gaps.py $T 0.5 1000 > $O/r03_single_request_trace_gaps.txt 2>&1
cat $O/r03_single_request_trace_gaps.txt
rm -rf $O/prof1
