set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/gpu_tests3.log 2>&1 || { tail -30 gpurun_out/gpu_tests3.log; exit 1; }
tail -2 gpurun_out/gpu_tests3.log
timeout -k 10 600 python bench.py > gpurun_out/bench15.log 2> gpurun_out/bench15.err
tail -c 600 gpurun_out/bench15.log
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r1f -o run -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/prof_r1f.log 2>&1
rm -f gpurun_out/prof_r1f/*kernel_trace.csv
timeout -k 10 500 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch2 -o run -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline > gpurun_out/pmc_fetch2.log 2>&1
timeout -k 10 500 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write2 -o run -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline > gpurun_out/pmc_write2.log 2>&1
python3 - <<'PY'
import csv, glob, json, collections
out = {}
for name, d in (("FETCH_SIZE", "gpurun_out/pmc_fetch2"), ("WRITE_SIZE", "gpurun_out/pmc_write2")):
    f = glob.glob(d + "/*counter_collection.csv")[0]
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != name: continue
        k = r["Kernel_Name"]
        k = "gemm_pp_kernel" if "gemm_pp_kernel" in k else ("gemm_nt_kernel<bf16>" if ("gemm_nt_kernel" in k and "unsigned short" in k) else None)
        if k is None: continue
        agg[k][0] += 1; agg[k][1] += float(r["Counter_Value"])
    out[name] = {k: {"launches": v[0], "sum": v[1]} for k, v in agg.items()}
json.dump(out, open("gpurun_out/pmc_gemm_raw.json", "w"), indent=1)
print(json.dumps(out))
PY
rm -rf gpurun_out/pmc_fetch2 gpurun_out/pmc_write2
