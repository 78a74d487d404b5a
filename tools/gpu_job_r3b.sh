cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r3b
mkdir -p $O
timeout -k 10 1000 python -m pytest tests -q -m gpu -p no:cacheprovider -rP --durations=8 > $O/gpu_tests.log 2>&1
echo "pytest rc=$?" >> $O/gpu_tests.log
grep -h "^FAILED\|^ERROR\| passed\| failed\|pytest rc" $O/gpu_tests.log | cut -c1-300
grep -h "full depth\|negated\|C4 chunk\|teacher-forced\|codes differing\|out vs restatement\|fp8 engine vs\|static activation\|C2 shapes\|batch .* x full depth" $O/gpu_tests.log | cut -c1-600
