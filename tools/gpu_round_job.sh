set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/gpu_tests5.log 2>&1 || { tail -30 gpurun_out/gpu_tests5.log; exit 1; }
tail -2 gpurun_out/gpu_tests5.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke2.log 2>&1 || { tail -20 gpurun_out/smoke2.log; exit 1; }
tail -3 gpurun_out/smoke2.log
timeout -k 10 900 python bench.py > gpurun_out/bench16.log 2> gpurun_out/bench16.err
tail -c 400 gpurun_out/bench16.log
timeout -k 10 700 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r1g -o run -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/prof_r1g.log 2>&1
rm -f gpurun_out/prof_r1g/*kernel_trace.csv
