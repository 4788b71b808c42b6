# PMC passes over the served step's launch (arl_env_step_served): bash tools/serve_pmc.sh [n_env]   (repo root, GPU box)
N=${1:-256}; R=$(pwd); cd /tmp; export TMPDIR=/tmp
for set in "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU"; do
  rm -rf /tmp/pp; timeout 120 rocprofv3 --pmc $set --output-format csv -d /tmp/pp -o p -- python $R/tools/serve_step_probe.py --pmc $N > /tmp/pp.log 2>&1
  python - <<PY
import csv, glob, collections
f = glob.glob("/tmp/pp/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])) if f else []:
    n = r["Kernel_Name"]
    if not any(k in n for k in ("serve_step", "igemm_split_kernel<4, 1, 1, 2")):
        continue
    acc[n[:n.index("(")].replace("(anonymous namespace)::", "").replace("void ", "")[:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in acc.items():
    print(k, {n: round(sum(v) / len(v)) for n, v in c.items()})
if not f: print(open("/tmp/pp.log").read()[-1500:])
PY
done
