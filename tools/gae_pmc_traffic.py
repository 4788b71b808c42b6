"""HBM bytes per launch of the GAE scan from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate
runs of `python bench.py --roofline-only`), corrected as /opt/skills/guides/MI355X_MICROARCH.md prescribes
(gfx950 reports half of the wide coalesced read traffic).
usage: gae_pmc_traffic.py fetch_counter_collection.csv write_counter_collection.csv [log2_elems [kernel label]] > gae_pmc_traffic.json
(the same passes over `bench.py --workload a2c1024 --roofline-only` give the n-step scan's record: tools/nstep_pmc.sh)"""
import csv
import hashlib
import json
import os
import sys

SCAN_SRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "accel_rl_amd", "csrc", "scan.hip")


def median_of(path, counter):
    vals = []
    with open(path) as f:
        for r in csv.DictReader(f):
            if "scan_lds_kernel" in r["Kernel_Name"] and r["Counter_Name"] == counter:
                vals.append(float(r["Counter_Value"]))
    vals.sort()
    return vals[len(vals) // 2], len(vals)


fetch_kb, n = median_of(sys.argv[1], "FETCH_SIZE")
write_kb, _ = median_of(sys.argv[2], "WRITE_SIZE")
log2 = int(sys.argv[3]) if len(sys.argv) > 3 else 26
t = 5
n_env = (1 << log2) // t
fetch_b, write_b = int(fetch_kb * 1024 * 2), int(write_kb * 1024)
print(json.dumps(dict(
    log2_elems=log2, kernel=sys.argv[4] if len(sys.argv) > 4 else "scan_lds_kernel<false,0,256>", FETCH_SIZE_KB=fetch_kb, WRITE_SIZE_KB=write_kb,
    fetch_bytes_corrected=fetch_b, write_bytes=write_b, hbm_bytes_per_launch=fetch_b + write_b,
    algorithmic_bytes_per_launch=17 * n_env * t + 4 * n_env,
    scan_hip_sha1=hashlib.sha1(open(SCAN_SRC, "rb").read()).hexdigest(),      # bench.py drops the figure when scan.hip changes
    note="rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `python bench.py "
         "--roofline-only`; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports 1/2 of wide coalesced "
         "reads); median over %d launches; raw CSVs next to this file's per-round copy (profiles/rNN/)" % n), indent=1))
