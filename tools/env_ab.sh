# Same-box A/B of HIP runtime environment settings on the bench line: bash tools/env_ab.sh <out dir> "VAR=val" ["VAR=val VAR2=val" ...]
# ("-" = the default environment); two alternating rounds
O=$1; shift; mkdir -p $O
for round in 1 2; do
  i=0
  for E in "$@"; do
    i=$((i+1))
    if [ "$E" = "-" ]; then EV=""; else EV="$E"; fi
    timeout 120 env $EV python bench.py --no-cpu-baseline --steps 100 2>/dev/null | tail -n 1 > $O/env_${i}_$round.json
    python - <<PY
import json
try:
    d = json.load(open("$O/env_${i}_$round.json"))
    print("%-44s" % "$E", $round, d["value"], d["phases"]["rollout_ms"], d["phases"]["learner_ms"], d["mfma"]["sustained_clock_ghz"],
          " ".join("%.1f" % r["avg_launch_us"] for r in d["mfma"]["kernels"]))
except Exception as e:
    print("%-44s" % "$E", $round, "FAILED", e)
PY
  done
done
