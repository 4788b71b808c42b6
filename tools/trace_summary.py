"""Summarise a rocprofv3 kernel_trace.csv: steady-state window only (last `frac` of
the trace by time), per-kernel totals, GPU busy vs idle."""
import csv, sys, collections
path, frac = sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
t0, t1 = rows[0][0], max(r[1] for r in rows)
cut = t1 - (t1 - t0) * frac
rows = [r for r in rows if r[0] >= cut]
span = rows[-1][1] - rows[0][0]
busy, last_end = 0, rows[0][0]
agg = collections.defaultdict(lambda: [0, 0])
for s, e, n in rows:
    agg[n][0] += e - s; agg[n][1] += 1
    if e > last_end:
        busy += e - max(s, last_end); last_end = e
print("window %.1f ms, busy %.1f ms (%.1f%%), %d launches" % (span / 1e6, busy / 1e6, 100. * busy / span, len(rows)))
tot = sum(v[0] for v in agg.values())
for n, (d, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:40]:
    print("%6.2f%% %9.3f ms %7d calls %9.1f us/call  %s" % (100. * d / tot, d / 1e6, c, d / c / 1e3, n[:110]))
