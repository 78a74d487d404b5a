#!/usr/bin/env python3
"""Condense two `rocprofv3 --pmc` passes (FETCH_SIZE, WRITE_SIZE; one counter per pass, as MI355X_MICROARCH.md's HBM
section prescribes) into profiles/r01_pmc_gemm.json: HBM-side bytes per launch of the GEMM kernels.

    python tools/summarize_pmc.py <dir with *counter_collection.csv of pass 1> <dir of pass 2> <out.json> "<command note>"

Corrections applied (same guide): both counters are in KiB; on gfx950 FETCH_SIZE counts half of a wide coalesced read
stream, so read bytes = 2 x FETCH_SIZE x 1024.  Infinity-Cache hits are included (L2-miss traffic = an upper bound on
HBM traffic)."""
import csv
import glob
import json
import os
import sys


def family(name: str) -> str:
    if "gemm_pp_kernel" in name:
        return "gemm_pp_kernel"
    if "gemm_nt_kernel" in name:
        return "gemm_nt_kernel<bf16>" if ("unsigned short" in name or "<t" in name) else "gemm_nt_kernel<f32>"
    return ""


def collect(d: str) -> dict:
    out = {}
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        raise SystemExit(f"no counter_collection.csv under {d}")
    for f in files:
        for r in csv.DictReader(open(f)):
            fam = family(r["Kernel_Name"])
            if not fam:
                continue
            c = out.setdefault(r["Counter_Name"], {}).setdefault(fam, {"launches": 0, "sum": 0.0})
            c["launches"] += 1
            c["sum"] += float(r["Counter_Value"])
    return out


def main() -> None:
    d1, d2, dst = sys.argv[1], sys.argv[2], sys.argv[3]
    note = sys.argv[4] if len(sys.argv) > 4 else ""
    raw = {}
    for d in (d1, d2):
        for k, v in collect(d).items():
            raw[k] = v
    f, w = raw["FETCH_SIZE"]["gemm_pp_kernel"], raw["WRITE_SIZE"]["gemm_pp_kernel"]
    fetch = 2.0 * 1024.0 * f["sum"] / f["launches"]
    write = 1024.0 * w["sum"] / w["launches"]
    out = {
        "command": note,
        "kernel": "gemm_pp_kernel (all instantiations; the launches of the sampler calls plus the plan-tuning launches on the same shapes)",
        "raw": raw,
        "corrections": "FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts half of a wide coalesced read stream "
                       "(MI355X_MICROARCH.md, HBM), so read bytes = 2 x FETCH_SIZE x 1024; Infinity-Cache hits are included "
                       "(L2-miss traffic, an upper bound on HBM traffic)",
        "fetch_bytes_per_launch": fetch, "write_bytes_per_launch": write, "traffic_bytes_per_launch": fetch + write,
    }
    json.dump(out, open(dst, "w"), indent=1)
    print(json.dumps({k: out[k] for k in ("fetch_bytes_per_launch", "write_bytes_per_launch", "traffic_bytes_per_launch")}))


if __name__ == "__main__":
    main()
