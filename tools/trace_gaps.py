"""GPU idle time between kernels of a rocprofv3 --kernel-trace CSV.

Usage: trace_gaps.py <kernel_trace.csv> [skip_first_fraction] [host_gap_us]

Kernels are merged over all queues into busy intervals; a gap is the time between the end of one busy interval and the start of the
next.  Gaps longer than `host_gap_us` (default 1000) are host phases (weight creation, the bench's python between legs) and are
reported separately; the rest are what a hipGraph capture of the sampler's step loop could remove.  The first `skip_first_fraction`
of the trace (default 0.5: model build, warm-up) is dropped."""
import csv, sys
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
host_gap = (float(sys.argv[3]) if len(sys.argv) > 3 else 1000.0) * 1e3
t0 = rows[0][0] + (rows[-1][1] - rows[0][0]) * skip
rows = [r for r in rows if r[0] >= t0]
busy, cur_end, gaps, host, prev = 0, rows[0][0], [], [], ""
pairs = {}
for s, e, n in rows:
    if s > cur_end:
        g = s - cur_end
        if g > host_gap:
            host.append(g)
        else:
            gaps.append(g)
            k = (prev[:48], n[:48])
            c = pairs.setdefault(k, [0, 0])
            c[0] += 1; c[1] += g
        busy += e - s
        cur_end = e
    elif e > cur_end:
        busy += e - cur_end
        cur_end = e
    prev = n
gaps.sort()
tot = sum(gaps)
q = lambda p: gaps[min(len(gaps) - 1, int(p * len(gaps)))] / 1e3 if gaps else 0.0
print(f"kernels {len(rows)}; some kernel running {busy/1e6:.2f} ms; {len(host)} host phases > {host_gap/1e3:.0f} us ({sum(host)/1e6:.2f} ms, excluded)")
print(f"device-side gaps: {len(gaps)} ({100.0*len(gaps)/max(len(rows),1):.1f} % of the launches are preceded by one), total {tot/1e6:.3f} ms = "
      f"{100.0*tot/max(busy+tot,1):.2f} % of busy + gaps; median {q(0.5):.2f} us, p90 {q(0.9):.2f} us, max {gaps[-1]/1e3 if gaps else 0:.1f} us")
print("largest contributors (kernel before -> kernel after): count, total ms, mean us")
for k, (c, g) in sorted(pairs.items(), key=lambda kv: -kv[1][1])[:10]:
    print(f"  {c:6d} {g/1e6:9.3f} {g/c/1e3:8.2f}  {k[0]} -> {k[1]}")
