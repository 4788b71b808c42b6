"""Idle gaps (> 2 us) between consecutive kernels of ONE steady-state bench step of a rocprofv3 kernel_trace.csv, each
with the kernel in front of it: where the GPU waits (graph boundaries, eager launches between graphs).
usage: python tools/gap_list.py kernel_trace.csv"""
import csv, sys
rows=[]
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
marks=[i for i,r in enumerate(rows) if "scan_lds_kernel" in r[2]]
a,b=marks[-6],marks[-5]
last=rows[a-1][1]
def short(n):
    n=n.replace("(anonymous namespace)::","").replace("void ","")
    return n[:70]
for i in range(a-3,b+3):
    s,e,n=rows[i]
    gap=(s-rows[i-1][1])/1e3
    if gap>2.0 or abs(i-a)<3:
        for j in (i-1,i):
            ss,ee,nn=rows[j]
            print("%s %8.1f us dur %6.1f gap %6.1f %s" % ("   " if j<i else ">>>", (ss-rows[a][0])/1e3,(ee-ss)/1e3,(ss-rows[j-1][1])/1e3, short(nn)))
        print()
