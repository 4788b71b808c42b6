# Is the PPO step power-limited?  Sample socket power and the shader clock (rocm-smi) while bench.py runs a long timed
# region, for the three contraction routes (ARL_CONV_PRECISION 9 = default, 6, 0 = fp32 MFMA chain).
# usage: bash tools/power_bench.sh <out dir>
O=${1:-gpurun_out/power}; mkdir -p $O
rocm-smi --showmaxpower > $O/maxpower.txt 2>&1
for mode in 9 6 0; do
  (for i in $(seq 1 400); do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr -s ' \t' ' ' | tr '\n' ';'; echo; sleep 0.1; done) > $O/smi_$mode.log &
  SPID=$!
  ARL_CONV_PRECISION=$mode python bench.py --no-cpu-baseline --no-roofline --steps 6000 --warmup 50 2>/dev/null | tail -n 1 > $O/bench_$mode.json
  kill $SPID 2>/dev/null; wait $SPID 2>/dev/null
  python - <<PY
import json, re
d = json.load(open("$O/bench_$mode.json"))
rows = [l for l in open("$O/smi_$mode.log") if "Power" in l]
pw = sorted(float(m.group(1)) for l in rows for m in [re.search(r"Power \(W\): ([0-9.]+)", l)] if m)
ck = sorted(int(m.group(1)) for l in rows for m in [re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", l)] if m)
print("mode $mode: %.0f env-steps/s, ms/step %.3f, sustained clock %s; power W p50 %s p90 %s max %s; sclk MHz p10 %s p50 %s (n=%d)" % (
    d["value"], d["ms_per_step"], d.get("mfma", {}).get("sustained_clock_ghz"),
    pw[len(pw) // 2] if pw else None, pw[len(pw) * 9 // 10] if pw else None, pw[-1] if pw else None,
    ck[len(ck) // 10] if ck else None, ck[len(ck) // 2] if ck else None, len(rows)))
PY
done 2>&1 | tee $O/summary.txt
head -n 3 $O/smi_9.log >> $O/summary.txt
