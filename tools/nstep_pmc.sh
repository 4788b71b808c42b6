# HBM traffic of the n-step return scan (config 3's roofline kernel) at 2^26 elements: two rocprofv3 --pmc passes over
# `python bench.py --workload a2c1024 --roofline-only`, summarised into <out>/nstep_pmc_traffic.json (the record
# bench.py reads from profiles/; keyed by scan.hip's sha1).
# usage: bash tools/nstep_pmc.sh [out dir]     (from the repo root, on the GPU box)
R=$(pwd); O=$R/${1:-gpurun_out/r05}; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pn_$c; timeout 600 rocprofv3 --pmc $c --output-format csv -d /tmp/pn_$c -o p -- python $R/bench.py --workload a2c1024 --roofline-only > /tmp/pn_$c.log 2>&1
  cp $(find /tmp/pn_$c -name "*counter_collection.csv" | head -n 1) $O/nstep_pmc_$c.csv
done
python $R/tools/gae_pmc_traffic.py $O/nstep_pmc_FETCH_SIZE.csv $O/nstep_pmc_WRITE_SIZE.csv 26 "scan_lds_kernel<NSTEP,NEP50,256>" | tee $O/nstep_pmc_traffic.json
