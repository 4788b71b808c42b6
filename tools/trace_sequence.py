"""One bench step as a timeline: every kernel between two consecutive GAE scans of a rocprofv3
kernel_trace.csv, with start offset, duration and the idle gap before it.
usage: trace_sequence.py kernel_trace.csv [marker-substring [which-window, default -1 = last]]"""
import csv
import sys

path = sys.argv[1]
marker = sys.argv[2] if len(sys.argv) > 2 else "scan_lds_kernel"
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
marks = [i for i, r in enumerate(rows) if marker in r[2]]
which = int(sys.argv[3]) if len(sys.argv) > 3 else -1
a, b = marks[which - 1], marks[which]
t0, last = rows[a][0], rows[a][0]
busy = 0
for s, e, n in rows[a:b]:
    short = n.replace("(anonymous namespace)::", "").replace("void ", "")
    short = short[:short.index("(")] if "(" in short and not short.startswith("(") else short
    print("%9.1f us  %7.1f us  gap %5.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - last) / 1e3, short[:90]))
    busy += e - max(s, last) if e > last else 0
    last = max(last, e)
print("step %.1f us, busy %.1f us, %d launches" % ((rows[b][0] - t0) / 1e3, busy / 1e3, b - a))
