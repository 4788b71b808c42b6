# PMC passes over conv 1's two kernels at the PPO minibatch: bash tools/conv1_pmc.sh [out file]   (repo root, GPU box)
R=$(pwd); O=$R/${1:-gpurun_out/r04/conv1_pmc.txt}; mkdir -p $(dirname $O); cd /tmp; export TMPDIR=/tmp
{
for set in "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU" \
           "SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr"; do
  rm -rf /tmp/pp; timeout 120 rocprofv3 --pmc $set --output-format csv -d /tmp/pp -o p -- python $R/tools/conv1_u8_probe.py 5 > /tmp/pp.log 2>&1
  python - <<PY
import csv, glob, collections
f = glob.glob("/tmp/pp/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])) if f else []:
    n = r["Kernel_Name"]
    if not any(k in n for k in ("conv1_img", "wgrad", "igemm")):
        continue
    acc[n[:n.index("(")].replace("(anonymous namespace)::", "").replace("void ", "")[:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in acc.items():
    print(k, {n: round(sum(v) / len(v)) for n, v in c.items()})
if not f: print(open("/tmp/pp.log").read()[-1500:])
PY
done
} 2>&1 | grep -v amdgpu.ids | tee $O
