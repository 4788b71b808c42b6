# The N > 1 learner's structure at world size 1 (collective forced, backend nccl = RCCL), next to the single-GPU line:
# captured in one hipGraph (default), eager minibatches, and captured with ONE blocking all-reduce per minibatch.
# usage: bash tools/sync_ab.sh <out dir>
O=${1:-gpurun_out/r04}; mkdir -p $O
run() { tag=$1; shift; env "$@" timeout 600 python bench.py --no-cpu-baseline --no-roofline --steps 300 --warmup 30 2>/dev/null | tail -n 1 > $O/bench_$tag.json
  python - <<PY
import json
d = json.load(open("$O/bench_$tag.json"))
print("%-28s %9.1f env-steps/s  %.3f ms/step  phases %s" % ("$tag", d["value"], d["ms_per_step"], d.get("phases")))
PY
}
for round in 1 2; do
run single_r$round A=1
run force_sync_captured_r$round ARL_FORCE_SYNC=1
run force_sync_eager_r$round ARL_FORCE_SYNC=1 ARL_SYNC_GRAPH=0
run force_sync_captured_blocking_r$round ARL_FORCE_SYNC=1 ARL_SYNC_OVERLAP=0
done 2>&1 | tee $O/sync_ab.txt
