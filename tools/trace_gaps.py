"""GPU idle time between kernels of a rocprofv3 --kernel-trace CSV: per stream (queue) the gaps between one kernel's end and the next one's start,
and overall the fraction of the traced wall interval in which no kernel ran.  Usage: trace_gaps.py <kernel_trace.csv> [skip_first_fraction]"""
import csv, sys
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "0"), r["Kernel_Name"]))
rows.sort()
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
t0 = rows[0][0] + (rows[-1][1] - rows[0][0]) * skip          # drop warm-up / tuning at the start
rows = [r for r in rows if r[0] >= t0]
wall = rows[-1][1] - rows[0][0]
busy, cur_end, gaps, big, prev = 0, rows[0][0], [], [], ""
for s, e, q, n in rows:
    if s > cur_end:
        gaps.append(s - cur_end)
        if s - cur_end > 20000: big.append((s - cur_end, prev[:40], n[:40]))
        busy += e - s
        cur_end = e
    elif e > cur_end:
        busy += e - cur_end
        cur_end = e
    prev = n
print(f"kernels {len(rows)}, wall {wall/1e6:.2f} ms, some kernel running {busy/1e6:.2f} ms ({100*busy/wall:.1f} %), idle {100*(1-busy/wall):.1f} % in {len(gaps)} gaps, "
      f"median gap {sorted(gaps)[len(gaps)//2]/1e3:.2f} us, mean {sum(gaps)/max(len(gaps),1)/1e3:.2f} us")
from collections import Counter
c = Counter()
tot = Counter()
for g, a, b in big:
    c[(a, b)] += 1; tot[(a, b)] += g
print(f"gaps > 20 us: {len(big)}, {sum(g for g, _, _ in big)/1e6:.2f} ms in total; by (kernel before -> kernel after):")
for k, v in tot.most_common(12):
    print(f"  {v/1e6:8.2f} ms in {c[k]:4d} gaps: {k[0]} -> {k[1]}")
