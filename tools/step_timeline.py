"""One steady-state step of bench.py as the device ran it: every launch between two consecutive rollout starts
(the N-th and N+1-th launch of the marker kernel: the return scan, once per step), with its start offset, duration and the idle gap in
front of it.  usage: step_timeline.py kernel_trace.csv [which_step [marker_substring]]"""
import csv
import sys

path = sys.argv[1]
which = int(sys.argv[2]) if len(sys.argv) > 2 else 60
marker = sys.argv[3] if len(sys.argv) > 3 else "scan_lds_kernel"
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
marks = [i for i, r in enumerate(rows) if marker in r[2]]
a, b = marks[which], marks[which + 1]
# the step's first launch is the copy in front of the marker, if any
while a > 0 and "copyBuffer" in rows[a - 1][2] and rows[a][0] - rows[a - 1][1] < 3000:
    a -= 1
    b -= 1
t0 = rows[a][0]
last = rows[a - 1][1] if a else t0
busy = 0
for s, e, n in rows[a:b]:
    gap = s - last
    print("%9.1f us  %7.1f us  gap %6.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap / 1e3, n[:110]))
    busy += e - max(s, last) if e > last else 0
    last = max(last, e)
span = rows[b][0] - t0
print("step: %.1f us, busy %.1f us, %d launches" % (span / 1e3, busy / 1e3, b - a))
