"""HBM bytes per launch of the GAE scan from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate
runs of `python bench.py --roofline-only`), corrected as /opt/skills/guides/MI355X_MICROARCH.md prescribes
(gfx950 reports half of the wide coalesced read traffic).
usage: gae_pmc_traffic.py fetch_counter_collection.csv write_counter_collection.csv [log2_elems] > gae_pmc_traffic.json"""
import csv
import json
import sys


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
    log2_elems=log2, kernel="scan_lds_kernel<false,0,256>", FETCH_SIZE_KB=fetch_kb, WRITE_SIZE_KB=write_kb,
    fetch_bytes_corrected=fetch_b, write_bytes=write_b, hbm_bytes_per_launch=fetch_b + write_b,
    algorithmic_bytes_per_launch=17 * n_env * t + 4 * n_env,
    note="rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `python bench.py "
         "--roofline-only`; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports 1/2 of wide coalesced "
         "reads); median over %d launches; raw CSVs in profiles/r01/" % n), indent=1))
