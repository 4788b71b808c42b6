# PMC passes over the prototype binary: bash tools/proto/pmc_proto.sh <binary> <kernel-name-substring> [args...]
BIN=$(pwd)/$1; NEEDLE=$2; shift; shift; cd /tmp; export TMPDIR=/tmp
for set in "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU" \
           "SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr"; do
  rm -rf /tmp/pp; timeout 120 rocprofv3 --pmc $set --output-format csv -d /tmp/pp -o p -- $BIN "$@" > /tmp/pp.log 2>&1
  python - <<PY
import csv, glob, collections
f = glob.glob("/tmp/pp/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])) if f else []:
    n = r["Kernel_Name"]
    if "$NEEDLE" not in n:
        continue
    acc[n[:90]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in acc.items():
    print(k, {n: round(sum(v) / len(v)) for n, v in c.items()})
if not f: print(open("/tmp/pp.log").read()[-1500:])
PY
done
