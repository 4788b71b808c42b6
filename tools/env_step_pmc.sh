# HBM traffic of arl_env_step at a bandwidth-bound size: two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over
# tools/env_step_probe.py (rollout rows written once), summarised per env-step into profiles/env_step_pmc.json.
# usage: bash tools/env_step_pmc.sh [n_envs] [out dir]     (from the repo root, on the GPU box)
N=${1:-32768}; R=$(pwd); O=$R/${2:-gpurun_out/r04}; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pe_$c; timeout 300 rocprofv3 --pmc $c --output-format csv -d /tmp/pe_$c -o p -- python $R/tools/env_step_probe.py $N 1 > /tmp/pe_$c.log 2>&1
  cp $(find /tmp/pe_$c -name "*counter_collection.csv" | head -n 1) $O/env_step_pmc_$c.csv
done
python $R/tools/env_step_pmc.py $O/env_step_pmc_FETCH_SIZE.csv $O/env_step_pmc_WRITE_SIZE.csv $N | tee $O/env_step_pmc.json
