# What bounds arl_env_step at a bandwidth-bound size (16 384 envs, rollout rows written once)?
#  (1) timing knock-outs (arl_dev_env_variant; results wrong): every env reads bank frame 0 / the older planes of the
#      stack are not stored / nor loaded;
#  (2) rocprofv3 --pmc passes over the product kernel: L1 (TCP) and L2 (TCC) request / hit / miss counts, the fabric
#      request counters behind FETCH_SIZE / WRITE_SIZE, wave-cycle accounting (SQ_WAIT_ANY = parked at s_waitcnt /
#      barrier, SQ_WAIT_INST_ANY = stalled at issue, SQ_ACTIVE_INST_*).
# usage: bash tools/env_step_bound.sh [n_envs] [out dir]      (from the repo root, on the GPU box)
N=${1:-16384}; R=$(pwd); O=$R/${2:-gpurun_out/r04}; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
{
for v in 0 1 2 3; do ARL_ENV_VARIANT=$v python $R/tools/env_step_probe.py $N 1 2>&1 | grep dbg; done
for set in "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INST_CYCLES_VMEM SQ_WAVES SQ_ACTIVE_INST_SCA" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" \
           "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_WRITE_sum TCC_EA0_RDREQ_32B_sum"; do
  rm -rf /tmp/pb; timeout 200 rocprofv3 --pmc $set --output-format csv -d /tmp/pb -o p -- python $R/tools/env_step_probe.py $N 1 > /tmp/pb.log 2>&1
  python - <<PY
import csv, glob, collections
f = glob.glob("/tmp/pb/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f[0])) if f else []:
    if "env_step_kernel" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("pmc (median per launch over %d launches):" % (max(map(len, acc.values())) if acc else 0),
      {n: sorted(v)[len(v) // 2] for n, v in acc.items()})
if not acc: print(open("/tmp/pb.log").read()[-800:])
PY
done
} 2>&1 | grep -v amdgpu.ids | tee $O/env_step_bound.txt
