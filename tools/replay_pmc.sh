# HBM traffic of arl_replay_extract at bench.py's roofline batch: two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE;
# counters only, no tracing) over tools/replay_pmc.py, summarised into <out>/replay_extract_pmc.json.
# usage: bash tools/replay_pmc.sh [batch] [out dir]     (from the repo root, on the GPU box)
B=${1:-4096}; R=$(pwd); O=$R/${2:-gpurun_out/r05}; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pr_$c; timeout 300 rocprofv3 --pmc $c --output-format csv -d /tmp/pr_$c -o p -- python $R/tools/replay_pmc.py probe $B > /tmp/pr_$c.log 2>&1
  cp $(find /tmp/pr_$c -name "*counter_collection.csv" | head -n 1) $O/replay_extract_pmc_$c.csv
done
python $R/tools/replay_pmc.py summarise $O/replay_extract_pmc_FETCH_SIZE.csv $O/replay_extract_pmc_WRITE_SIZE.csv $B | tee $O/replay_extract_pmc.json
