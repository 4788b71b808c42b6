set -x
R=/root/repo; O=$R/gpurun_out/fin; mkdir -p $O
cd $R && timeout 900 python bench.py > $O/bench.log 2>$O/bench.err; tail -1 $O/bench.log > $O/bench.json
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pA -o b -- python $R/bench.py --steps 60 --warmup 20 --no-cpu-baseline > $O/prof.log 2>&1
tail -200 $O/prof.log | grep '^{"metric' | tail -1 > $O/bench_profiled.json
cp $(find /tmp/pA -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats.csv
python $R/tools/trace_summary.py $(find /tmp/pA -name "*kernel_trace.csv" | head -1) 200 560 40 > $O/bench_steady_state_summary.txt
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pB -o g -- python $R/bench.py --roofline-only > $O/gae.log 2>&1
cp $(find /tmp/pB -name "*kernel_stats.csv" | head -1) $O/gae_kernel_stats.csv
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pC -o f -- python $R/bench.py --roofline-only > $O/gaef.log 2>&1
cp $(find /tmp/pC -name "*counter_collection.csv" | head -1) $O/gae_pmc_FETCH_SIZE.csv
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pD -o w -- python $R/bench.py --roofline-only > $O/gaew.log 2>&1
cp $(find /tmp/pD -name "*counter_collection.csv" | head -1) $O/gae_pmc_WRITE_SIZE.csv
python $R/tools/gae_pmc_traffic.py $O/gae_pmc_FETCH_SIZE.csv $O/gae_pmc_WRITE_SIZE.csv > $O/gae_pmc_traffic.json
cd $R; timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --output-format csv -d /tmp/pE -o m -- python tools/conv_bench.py 512 o > $O/mfma.log 2>&1
cp $(find /tmp/pE -name "*counter_collection.csv" | head -1) $O/mfma_pmc_counter_collection.csv
python tools/mfma_pmc_summary.py $O/mfma_pmc_counter_collection.csv > $O/mfma_pmc_summary.json
ls -la $O
