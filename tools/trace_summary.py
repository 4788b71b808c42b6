"""Summarise a rocprofv3 kernel_trace.csv inside a time window given as fractions [lo, hi] of the
whole trace (or, with lo, hi >= 1, between the lo-th and hi-th optimiser update): per-kernel totals, GPU busy
vs idle.  usage: trace_summary.py kernel_trace.csv [lo hi [top]]"""
import collections
import csv
import sys

path = sys.argv[1]
lo = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
hi = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
top = int(sys.argv[4]) if len(sys.argv) > 4 else 40
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
t0, t1 = rows[0][0], max(r[1] for r in rows)
a, b = t0 + (t1 - t0) * lo, t0 + (t1 - t0) * hi
if lo >= 1.0:       # lo, hi >= 1: window = from the lo-th to the hi-th optimiser update (update_kernel launches)
    ups = [r[0] for r in rows if "update_kernel" in r[2] or "update_noclip_kernel" in r[2]]
    a, b = ups[int(lo)], ups[int(hi)]
rows = [r for r in rows if r[0] >= a and r[1] <= b]
span = rows[-1][1] - rows[0][0]
busy, last_end = 0, rows[0][0]
agg = collections.defaultdict(lambda: [0, 0])
gap = collections.defaultdict(lambda: [0, 0])       # idle time in front of a kernel's launches (nothing else running)
for s, e, n in rows:
    agg[n][0] += e - s
    agg[n][1] += 1
    if s > last_end:
        gap[n][0] += s - last_end
        gap[n][1] += 1
    if e > last_end:
        busy += e - max(s, last_end)
        last_end = e
print("window %.1f ms, busy %.1f ms (%.1f%%), %d launches" % (span / 1e6, busy / 1e6, 100. * busy / span, len(rows)))
tot = sum(v[0] for v in agg.values())
for n, (d, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:top]:
    print("%6.2f%% %9.3f ms %7d calls %9.1f us/call  %s" % (100. * d / tot, d / 1e6, c, d / c / 1e3, n[:120]))
print("idle in front of (top 8):")
for n, (d, c) in sorted(gap.items(), key=lambda kv: -kv[1][0])[:8]:
    print("        %9.3f ms over %5d gaps = %6.1f us each, %d launches  %s" % (d / 1e6, c, d / c / 1e3, agg[n][1], n[:100]))
