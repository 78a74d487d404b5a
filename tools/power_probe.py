"""Is the bf16 GEMM bound by the power envelope?  Runs the production ping-pong kernel (cfg 5) at the sampler's QKVG shape for a few seconds
per operand pattern (zeros, one constant, random) and samples the board power and the shader clock from rocm-smi while it runs.
    python tools/power_probe.py            (on the GPU box)"""
import json, os, subprocess, sys, threading, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import gpu_util as U


def sample(stop, out):
    while not stop.is_set():
        try:
            r = subprocess.run(["rocm-smi", "-P", "-c", "--json"], capture_output=True, text=True, timeout=20)
            d = json.loads(r.stdout)
            card = d[sorted(d)[0]]
            rec = {}
            for k, v in card.items():
                kl = k.lower()
                if "power" in kl and "(w)" in kl:
                    rec["power_w"] = float(v)
                if kl.startswith("sclk clock speed"):
                    rec["sclk"] = v
            out.append(rec)
        except Exception as e:  # noqa: BLE001 - a probe: report and go on
            out.append({"error": repr(e)})
        time.sleep(0.2)


def run(label, A, W, M, N, K, secs=4.0):
    C = torch.zeros((M, N), dtype=torch.bfloat16, device="cuda")
    kw = dict(M=M, N=N, K=K, lda=K, ldw=K, ldc=N, Npad=N, cfg=5)
    for _ in range(20):
        U.gemm(A, W, C, **kw)
    torch.cuda.synchronize()
    stop, out = threading.Event(), []
    th = threading.Thread(target=sample, args=(stop, out))
    th.start()
    n = 0
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.time()
    s.record()
    while time.time() - t0 < secs:
        for _ in range(200):
            U.gemm(A, W, C, **kw)
        n += 200
        torch.cuda.synchronize()
    e.record(); torch.cuda.synchronize()
    stop.set(); th.join()
    us = s.elapsed_time(e) / n * 1e3
    pw = [r["power_w"] for r in out if "power_w" in r]
    ck = [r["sclk"] for r in out if "sclk" in r]
    print(f"{label:28s}: {us:7.1f} us = {2.0 * M * N * K / us / 1e6:6.0f} TFLOP/s | power W {pw} | sclk {ck} | {[r for r in out if 'error' in r][:1]}", flush=True)


M, N, K = 15360, 8192, 2048
z = torch.zeros((M + 256, K), dtype=torch.bfloat16, device="cuda")
wz = torch.zeros((N, K), dtype=torch.bfloat16, device="cuda")
run("zeros", z, wz, M, N, K)
run("constant 1.0 x 0.5", z + 1, wz + 0.5, M, N, K)
A = (torch.rand((M + 256, K), device="cuda") * 2 - 1).to(torch.bfloat16)
W = ((torch.rand((N, K), device="cuda") * 2 - 1) * 0.05).to(torch.bfloat16)
run("random uniform", A, W, M, N, K)
A = torch.randn((M + 256, K), device="cuda").to(torch.bfloat16)
W = (torch.randn((N, K), device="cuda") * 0.02).to(torch.bfloat16)
run("random normal", A, W, M, N, K)
run("random A, zero W", A, wz, M, N, K)

# joint attention (attn5_kernel, 24 rows: the batch-8 CFG step's launch) under the same sampling
import importlib.util
spec = importlib.util.spec_from_file_location("bench_attn4", os.path.join(os.path.dirname(os.path.abspath(__file__)), "bench_attn4.py"))
ba = importlib.util.module_from_spec(spec); spec.loader.exec_module(ba)
stop, out = threading.Event(), []
th = threading.Thread(target=sample, args=(stop, out))
th.start()
ba.run(24, iters=5000)
stop.set(); th.join()
print("attention 24 rows: power W", [r["power_w"] for r in out if "power_w" in r], "| sclk", [r["sclk"] for r in out if "sclk" in r], flush=True)
