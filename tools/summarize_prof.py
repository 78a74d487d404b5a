#!/usr/bin/env python3
"""Condense a rocprofv3 `--kernel-trace --stats` kernel_stats.csv into a short table (long template names trimmed)."""
import csv
import re
import sys


def short(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = name.replace("unsigned short", "bf16")
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([A-Za-z0-9_:]+(?:<[^()]*?>)?)", name)
    s = m.group(1) if m else name
    return s[:90]


def main(src: str, dst: str, note: str = "") -> None:
    rows = list(csv.DictReader(open(src)))
    tot = sum(int(r["TotalDurationNs"]) for r in rows)
    with open(dst, "w") as f:
        if note:
            f.write(f"# {note}\n")
        f.write(f"# source: rocprofv3 --kernel-trace --stats; total kernel time {tot/1e6:.2f} ms\n")
        f.write("kernel,calls,total_ms,avg_us,pct,min_us,max_us\n")
        for r in rows:
            t = int(r["TotalDurationNs"])
            if t / tot < 0.0005:
                continue
            f.write(f"{short(r['Name'])},{r['Calls']},{t/1e6:.3f},{float(r['AverageNs'])/1e3:.2f},{100*t/tot:.2f},"
                    f"{int(r['MinNs'])/1e3:.2f},{int(r['MaxNs'])/1e3:.2f}\n")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else "")
