# Same-box A/B of library builds: bash tools/ab_bench.sh <out dir> <lib.so> [<lib.so> ...]  (two rounds, alternating)
O=$1; shift; mkdir -p $O
for round in 1 2; do
  for L in "$@"; do
    n=$(basename $L .so)
    python tools/ab_lib.py $L bench.py --no-cpu-baseline --steps 100 2>/dev/null | tail -n 1 > $O/ab_${n}_$round.json
    python - <<PY
import json
d = json.load(open("$O/ab_${n}_$round.json"))
print("$n", $round, d["value"], d["phases"], d["mfma"]["frac"], d["mfma"]["sustained_clock_ghz"],
      " ".join("%.1f" % r["avg_launch_us"] for r in d["mfma"]["kernels"]))
PY
  done
done
