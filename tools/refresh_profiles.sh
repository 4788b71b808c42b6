# Regenerates the one-box part of profiles/r06 on one MI355X box: bash tools/refresh_profiles.sh  (from the repo root;
# ~14 GPU-minutes); output in gpurun_out/fin, to be copied into profiles/r06 (and profiles/*.json: the PMC records bench.py reads)
set -x
R=$(pwd); O=$R/gpurun_out/fin; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pA -o b -- python $R/bench.py --steps 60 --warmup 20 --no-cpu-baseline > $O/prof.log 2>&1
tail -n 200 $O/prof.log | grep '^{"metric' | tail -n 1 > $O/bench_profiled.json
cp $(find /tmp/pA -name "*kernel_stats.csv" | head -n 1) $O/bench_kernel_stats.csv
python $R/tools/trace_summary.py $(find /tmp/pA -name "*kernel_trace.csv" | head -n 1) 200 560 40 > $O/bench_steady_state_summary.txt
python $R/tools/step_timeline.py $(find /tmp/pA -name "*kernel_trace.csv" | head -n 1) 50 > $O/step_timeline.txt
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pB -o g -- python $R/bench.py --roofline-only > $O/gae.log 2>&1
cp $(find /tmp/pB -name "*kernel_stats.csv" | head -n 1) $O/gae_kernel_stats.csv
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pC -o f -- python $R/bench.py --roofline-only > $O/gaef.log 2>&1
cp $(find /tmp/pC -name "*counter_collection.csv" | head -n 1) $O/gae_pmc_FETCH_SIZE.csv
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pD -o w -- python $R/bench.py --roofline-only > $O/gaew.log 2>&1
cp $(find /tmp/pD -name "*counter_collection.csv" | head -n 1) $O/gae_pmc_WRITE_SIZE.csv
python $R/tools/gae_pmc_traffic.py $O/gae_pmc_FETCH_SIZE.csv $O/gae_pmc_WRITE_SIZE.csv > $O/gae_pmc_traffic.json
cp $O/gae_pmc_traffic.json $R/profiles/gae_pmc_traffic.json     # bench.py reads it (roofline.traffic; keyed by scan.hip's sha1)
cd $R; bash tools/nstep_pmc.sh gpurun_out/fin > /dev/null 2>&1; cp $O/nstep_pmc_traffic.json $R/profiles/nstep_pmc_traffic.json    # config 3's roofline.traffic
bash tools/replay_pmc.sh 4096 gpurun_out/fin > /dev/null 2>&1; cp $O/replay_extract_pmc.json $R/profiles/replay_extract_pmc.json   # config 5's
cd $R; bash tools/env_step_pmc.sh 32768 gpurun_out/fin > /dev/null 2>&1
cp $O/env_step_pmc.json $R/profiles/env_step_pmc.json     # bench.py reads it (env_step@sweep traffic; keyed by env.hip's sha1)
cd $R; timeout 900 python bench.py > $O/bench.log 2>$O/bench.err; tail -n 1 $O/bench.log > $O/bench.json
cd /tmp
cd $R; timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --output-format csv -d /tmp/pE -o m -- python tools/conv_bench.py 512 o > $O/mfma.log 2>&1
python tools/mfma_pmc_summary.py $(find /tmp/pE -name "*counter_collection.csv" | head -n 1) > $O/mfma_pmc_summary.json
python tools/gae_sweep.py > $O/gae_sweep.txt 2>&1
# the bf16-split routes: accuracy of every kernel against float64 and launch times per route; the bench line per route
python tools/split_check.py 512 48 2>&1 | grep -v amdgpu.ids > $O/split_check.txt
# (the bench line per route is in bench.json itself since round 6: alt_routes / accuracy)
# tile / persistence / timeline probes of the fp32 MFMA chain (ARL_CONV_PRECISION=0: arl_conv_geom.route = ARL_CONV_ROUTE_FP32), whose tile choices they label
export ARL_CONV_PRECISION=0
(for w in c1f c2f c3f df c3d c2d; do python tools/context_trace.py $w 2>&1 | grep -v amdgpu.ids; done) > $O/context_trace.txt
python tools/conv_trace.py 512 2>&1 | grep -v amdgpu.ids > $O/conv_trace.txt
python tools/learner_probe.py 2>&1 | grep minibatch > $O/learner_probe.txt
unset ARL_CONV_PRECISION
(for n in 256 1024 4096 16384 32768; do python tools/env_step_probe.py $n 2>&1 | grep dbg; done) > $O/env_step_probe.txt
python tools/batch_sweep.py 2>&1 | grep "^spec" > $O/batch_sweep_passes.txt
timeout 300 python bench.py --workload a2c1024 --steps 200 --warmup 20 2>/dev/null | tail -n 1 > $O/bench_a2c1024.json
timeout 300 python bench.py --scaling strong --total-envs 2048 --steps 20 --warmup 5 2>/dev/null | tail -n 1 > $O/bench_strong_2048_n1.json
timeout 600 python bench.py --workload catdqn --steps 30 --warmup 5 2>/dev/null | tail -n 1 > $O/bench_catdqn.json    # the reference's minibatch 32; alt: 512
python tools/fwd_tile_probe.py 2>&1 | grep -v amdgpu > $O/fwd_tile_probe.txt
ARL_BENCH_ONE_GPU=1 ARL_BENCH_BACKEND=gloo timeout 300 python bench.py --gpus 2 --steps 3 --warmup 2 --no-graph --no-cpu-baseline --no-roofline 2>/dev/null | tail -n 1 > $O/bench_spawn_2ranks_devmode.json
ARL_BENCH_ONE_GPU=1 ARL_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 8 --steps 3 --warmup 2 --no-graph --no-cpu-baseline --no-roofline 2>/dev/null | tail -n 1 > $O/bench_spawn_8ranks_devmode.json
ARL_FORCE_SYNC=1 timeout 300 python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | tail -n 1 > $O/bench_force_sync_n1.json
bash tools/env_step_bound.sh 16384 gpurun_out/fin > /dev/null 2>&1
bash tools/conv1_pmc.sh gpurun_out/fin/conv1_pmc.txt > /dev/null 2>&1
python tools/replay_bench.py 2>&1 | grep -v amdgpu > $O/replay_bench.txt
(for k in "conv2 fwd" "conv2 wgrad" "conv2 dgrad" "dense pair"; do echo "== $k"; bash tools/pmc_one.sh $k 2>&1 | grep -v amdgpu; done) > $O/pmc_one_split_kernels.txt
python tools/wave_scan_probe.py 26 2>&1 | grep -v amdgpu > $O/wave_scan_probe.txt
python tools/route_kernel_times.py 2>&1 | grep -v amdgpu > $O/route_kernel_times.txt
python tools/dense_fwd_tile_probe.py 2>&1 | grep -v amdgpu > $O/dense_fwd_tile_probe.txt
# configs 3 and 5 launch by launch (rocprofv3 kernel trace + tools/trace_sequence.py)
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pF -o a -- python $R/bench.py --workload a2c1024 --steps 60 --warmup 10 --no-cpu-baseline --no-roofline > /dev/null 2>&1
cp $(find /tmp/pF -name "*kernel_stats.csv" | head -n 1) $O/a2c1024_kernel_stats.csv
python $R/tools/trace_sequence.py $(find /tmp/pF -name "*kernel_trace.csv" | head -n 1) scan_lds_kernel -3 > $O/a2c1024_step_sequence.txt
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pG -o c -- python $R/bench.py --workload catdqn --steps 8 --warmup 3 --no-cpu-baseline --no-roofline --no-alt-routes > /dev/null 2>&1
cp $(find /tmp/pG -name "*kernel_stats.csv" | head -n 1) $O/catdqn_kernel_stats.csv
python $R/tools/trace_sequence.py $(find /tmp/pG -name "*kernel_trace.csv" | head -n 1) sumtree_sample_kernel -5 > $O/catdqn_update_sequence.txt
cd $R
ls -la $O
