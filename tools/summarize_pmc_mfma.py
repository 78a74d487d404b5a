#!/usr/bin/env python3
"""Condense a `rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace` pass into profiles/r02_pmc_mfma.json: MFMA-pipe
utilisation and effective clock per kernel family (MI355X_MICROARCH.md: SQ_VALU_MFMA_BUSY_CYCLES counts cycles, 32 per
v_mfma_f32_32x32x16_bf16, summed over the SIMDs; GRBM_GUI_ACTIVE is the sum over the 8 XCDs, effective clock = GRBM_GUI_ACTIVE / 8 / wall time).

    python tools/summarize_pmc_mfma.py <dir with *counter_collection.csv (+ *kernel_trace.csv)> <out.json> "<command note>"
"""
import csv, glob, json, os, sys

FAMILIES = ("gemm_pp_kernel", "attn5_kernel", "gemm_nt_kernel", "norm_kernel")
NSIMD = 256 * 4


def main():
    d, dst = sys.argv[1], sys.argv[2]
    note = sys.argv[3] if len(sys.argv) > 3 else ""
    per = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            fam = next((x for x in FAMILIES if x in r["Kernel_Name"]), None)
            if not fam:
                continue
            key = (fam, r["Dispatch_Id"])
            e = per.setdefault(key, {})
            e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
            if "Start_Timestamp" in r and "End_Timestamp" in r:
                e["ns"] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    out = {"command": note, "method": "per dispatch: utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8); "
           "effective clock = GRBM_GUI_ACTIVE / 8 / dispatch wall time; dispatches shorter than 0.2 ms are left out of the clock", "kernels": {}}
    for fam in FAMILIES:
        rows = [v for (f, _), v in per.items() if f == fam and v.get("GRBM_GUI_ACTIVE", 0) > 0]
        if not rows:
            continue
        busy = sum(v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) for v in rows)
        act = sum(v["GRBM_GUI_ACTIVE"] for v in rows) / 8.0
        long_rows = [v for v in rows if v.get("ns", 0) > 2e5]
        clk = (sum(v["GRBM_GUI_ACTIVE"] for v in long_rows) / 8.0) / sum(v["ns"] for v in long_rows) if long_rows else None
        out["kernels"][fam] = {"dispatches": len(rows), "mfma_busy_cycles": busy, "active_cycles": act,
                               "mfma_utilisation": busy / (NSIMD * act) if act else None, "effective_clock_ghz": clk}
    json.dump(out, open(dst, "w"), indent=1)
    print(json.dumps(out["kernels"]))


if __name__ == "__main__":
    main()
