"""Summarise rocprofv3 --pmc counter_collection.csv for one kernel name substring."""
import csv, sys, collections
path, needle = sys.argv[1], sys.argv[2]
vals = collections.defaultdict(list)
with open(path) as f:
    for r in csv.DictReader(f):
        if needle in r["Kernel_Name"]:
            vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in vals.items():
    v = sorted(v)
    print("%s n=%d mean=%.1f median=%.1f min=%.1f max=%.1f" % (k, len(v), sum(v) / len(v), v[len(v) // 2], v[0], v[-1]))
