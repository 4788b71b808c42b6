#!/usr/bin/env python
"""bench.py -- the rollout + learner hot path on N MI355X of one node.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch of synthetic input: one
`sampler.obtain_samples` (horizon x (policy forward, act_step, frame_step)) plus
one `algo.optimize_policy` (bootstrap forward, GAE scan, PPO epochs x minibatches
with the HIP flat-bucket optimiser; sync all-reduce when N > 1).
Workload = BASELINE.json configs[1]: PPO "breakout", 256 envs per GPU, horizon 5,
spec-1 CNN, minibatch 512 x 4 epochs  ->  1280 agent steps per GPU per step.
Metric = the reference runner's SamplesPerSecond (accel_rl/runners/accel_rl.py:93-98).

Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  "roofline":     the GAE scan kernel at a bandwidth-bound sweep size (HIP events
                  on the launch stream) + the at-config point, and
  "kernels":      event-timed averages of the hand-written HBM-bound kernels of the step,
  "mfma":         the policy's fp32-MFMA conv / dense kernels at the PPO minibatch vs the MFMA peak,
  "cpu_baseline": the oracle's CPU sampler port timed on this box's host cores.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_F32_PEAK_TFS = 157.3      # MI355X_MICROARCH.md: fp32-input MFMA = the fp32 vector rate
MFMA_BF16_PEAK_TFS = 2500.0    # MI355X_MICROARCH.md: dense bf16 MFMA peak (no sparsity)
N_ENVS, HORIZON, GAME, CNN_SPEC, MINIBATCH = 256, 5, "breakout", 1, 512
GAMES_8 = ["pong", "breakout", "seaquest", "space_invaders", "qbert", "beam_rider", "enduro", "ms_pacman"]


def build_workload(device, seed, rank, world, game, use_graph, quiet=True, kind="ppo", pad_actions_to=None):
    from accel_rl_amd.algos.pg.a2c import A2C, mA2C
    from accel_rl_amd.algos.pg.ppo import PPO, mPPO
    from accel_rl_amd.envs.synthetic_atari import SynthAtariEnv
    from accel_rl_amd.policies.atari_cnn_policy import AtariCnnPolicy
    from accel_rl_amd.policies.atari_cnn_specs import cnn_specs
    from accel_rl_amd.runners.accel_rl import AccelRL
    from accel_rl_amd.runners.sync import AccelRLSync
    from accel_rl_amd.sampler.gpu_sampler import GpuVecSampler
    from accel_rl_amd.util import logger
    logger.set_quiet(quiet)
    env_args = dict(game=game) if pad_actions_to is None else dict(game=game, pad_actions_to=pad_actions_to)
    sampler = GpuVecSampler(EnvCls=SynthAtariEnv, env_args=env_args, horizon=HORIZON,
                            n_parallel=16, envs_per=N_ENVS // 32, max_path_length=int(27e3),
                            mid_batch_reset=True, max_decorrelation_steps=2000, device=device,
                            use_graph=use_graph)
    policy = AtariCnnPolicy(**cnn_specs[CNN_SPEC])
    multi = world > 1 or os.environ.get("ARL_FORCE_SYNC") == "1"
    if kind == "a2c":
        algo = (mA2C if multi else A2C)(discount=0.99, gae_lambda=1)
    elif multi:
        algo = mPPO(discount=0.99, gae_lambda=0.95, optimizer_args=dict(minibatch_size=MINIBATCH))
    else:
        algo = PPO(discount=0.99, gae_lambda=0.95, optimizer_args=dict(minibatch_size=MINIBATCH))
    affinities = dict(gpu=device.index)
    if world > 1:
        # accel_rl_base.py:71-72 pins each runner to its affinities["gpu_cpus"] (the launcher's table,
        # scripts/launching/affinities.py): here the process's CPUs dealt out evenly to the node's ranks, so that eight
        # ranks replaying their graphs do not share cores
        cpus = sorted(os.sched_getaffinity(0))
        per = len(cpus) // world
        if per >= 4:        # (fewer: a rank's main thread would share its core with RCCL's polling proxy thread -- leave it)
            affinities["gpu_cpus"] = tuple(cpus[(rank % world) * per:(rank % world + 1) * per])
    if multi:
        algo.optimizer._force_collective = True
        if os.environ.get("ARL_SYNC_GRAPH") == "0":     # A/B switch: eager minibatches instead of one captured hipGraph
            algo.optimizer.graph_collectives = False
        if os.environ.get("ARL_SYNC_OVERLAP") == "0":   # A/B switch: ONE blocking all-reduce per minibatch, no tail / head split
            algo.optimizer._overlap_allreduce = False
        runner = AccelRLSync(algo=algo, policy=policy, sampler=sampler, n_steps=1e9, seed=seed,
                             affinities=affinities, log_interval_steps=1e8)
    else:
        runner = AccelRL(algo=algo, policy=policy, sampler=sampler, n_steps=1e9, seed=seed,
                         affinities=affinities, log_interval_steps=1e8)
    runner.startup()
    return runner, sampler, algo, policy


def one_step(itr, sampler, algo):
    samples, _ = sampler.obtain_samples(itr)
    algo.optimize_policy(itr, samples)


def graph_time_ms(fn, per_graph=20, replays=5):
    """Average device time of fn() as the step runs it: per_graph launches back to back inside ONE hipGraph (like
    the learner's graph), HIP events on the current stream around `replays` replays.  An isolated launch timed by
    its own pair of events also pays the event records and an idle-clock ramp (measured: +10...15 % on 40 us
    kernels); rocprofv3's per-kernel durations of the steady-state step agree with THIS figure."""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    from accel_rl_amd.util.misc import graph_capture_mode
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode=graph_capture_mode()):
        for _ in range(per_graph):
            fn()
    g.replay()
    torch.cuda.synchronize()
    start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    start.record()
    for _ in range(replays):
        g.replay()
    stop.record()
    torch.cuda.synchronize()
    return start.elapsed_time(stop) / (per_graph * replays)


def event_time_ms(fn, reps, warm=3):
    """Average device time of fn() over reps launches, HIP events on the current stream."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    start = [torch.cuda.Event(enable_timing=True) for _ in range(reps)]
    stop = [torch.cuda.Event(enable_timing=True) for _ in range(reps)]
    for i in range(reps):
        start[i].record()
        fn()
        stop[i].record()
    torch.cuda.synchronize()
    ts = sorted(s.elapsed_time(e) for s, e in zip(start, stop))
    return float(np.mean(ts)), float(ts[len(ts) // 2])


def roofline_gae(device, log2_elems, reps):
    """GAE scan at a bandwidth-bound size: algorithmic bytes = 17*N*T + 4*N."""
    from accel_rl_amd import _lib
    t = HORIZON
    n = (1 << log2_elems) // t
    gen = torch.Generator(device=device).manual_seed(1)
    r = torch.randn(n * t, device=device, generator=gen)
    v = torch.randn(n * t, device=device, generator=gen)
    d = (torch.rand(n * t, device=device, generator=gen) < 0.05).to(torch.uint8)
    lv = torch.randn(n, device=device, generator=gen)
    adv, ret = torch.empty_like(r), torch.empty_like(r)
    mean_ms, med_ms = event_time_ms(
        lambda: _lib.gae_scan(r, v, d, lv, 0.99, 0.95, n, t, adv, ret), reps)
    nbytes = 17 * n * t + 4 * n
    achieved = nbytes / (mean_ms * 1e-3) / 1e9
    return dict(bound="hbm", kernel="scan_lds_kernel<GAE,NEP50,256>", achieved=round(achieved, 1),
                peak=HBM_PEAK_GBS, unit="GB/s", frac=round(achieved / HBM_PEAK_GBS, 4),
                traffic=pmc_traffic(log2_elems),
                bytes_per_launch=nbytes, n_env=n, horizon=t, launches=reps,
                avg_launch_us=round(mean_ms * 1e3, 2), median_launch_us=round(med_ms * 1e3, 2))


def roofline_nstep(device, log2_elems, reps):
    """The n-step return scan (discount_returns + adv = ret - v, algos/pg/util.py:26-37, aac_base.py:121 -- config 3's
    process_samples) at the same bandwidth-bound size: 17 B per (env, t) + 4 B per env, as the GAE scan."""
    from accel_rl_amd import _lib
    t = HORIZON
    n = (1 << log2_elems) // t
    gen = torch.Generator(device=device).manual_seed(1)
    r = torch.randn(n * t, device=device, generator=gen)
    v = torch.randn(n * t, device=device, generator=gen)
    d = (torch.rand(n * t, device=device, generator=gen) < 0.05).to(torch.uint8)
    lv = torch.randn(n, device=device, generator=gen)
    adv, ret = torch.empty_like(r), torch.empty_like(r)
    mean_ms, med_ms = event_time_ms(lambda: _lib.nstep_return(r, d, v, lv, 0.99, n, t, ret, adv), reps)
    nbytes = 17 * n * t + 4 * n
    achieved = nbytes / (mean_ms * 1e-3) / 1e9
    return dict(bound="hbm", kernel="scan_lds_kernel<NSTEP,NEP50,256>", achieved=round(achieved, 1),
                peak=HBM_PEAK_GBS, unit="GB/s", frac=round(achieved / HBM_PEAK_GBS, 4),
                traffic=pmc_traffic(log2_elems, "nstep_pmc_traffic.json"),
                bytes_per_launch=nbytes, n_env=n, horizon=t, launches=reps,
                avg_launch_us=round(mean_ms * 1e3, 2), median_launch_us=round(med_ms * 1e3, 2))


def pmc_traffic(log2_elems, record="gae_pmc_traffic.json"):
    """HBM bytes per launch from the committed rocprofv3 --pmc passes (bench.py cannot
    read PMC counters itself); None if no profile for this size is committed -- or if it was taken from another
    version of the scan kernels (the record carries the sha1 of scan.hip; tools/refresh_profiles.sh renews it)."""
    import hashlib
    path = os.path.join(ROOT, "profiles", record)
    try:
        with open(path) as f:
            rec = json.load(f)
        with open(os.path.join(ROOT, "accel_rl_amd", "csrc", "scan.hip"), "rb") as f:
            sha = hashlib.sha1(f.read()).hexdigest()
        if rec.get("log2_elems") != log2_elems or rec.get("scan_hip_sha1") != sha:
            return None
        return rec["hbm_bytes_per_launch"]
    except (OSError, ValueError, KeyError):
        return None


ENV_SWEEP_ENVS = 16384          # environments of the bandwidth-bound env_step measurement (2.7 GB of rollout rows)


def env_step_sweep(device, n_env=ENV_SWEEP_ENVS):
    """arl_env_step (the whole env side of a rollout step: sample, emulate, max / crop / 2x2 box, stack, store) at a
    bandwidth-bound size.  Algorithmic bytes per env-step: preprocess + stack 75 520 (two raw frames in, one
    preprocessed frame out) + rollout store 33 295 + 4 A; `achieved` is computed from that figure, not from what the
    kernel moves (it also reads the three older frames of the stack, 24 960 B: the reference's stacked-observation layout
    -- the PMC traffic, tools/env_step_pmc.sh, is in `traffic`)."""
    from accel_rl_amd import _lib
    from accel_rl_amd.envs.synthetic_atari import SynthAtariEnv
    from accel_rl_amd.sampler.gpu_sampler import GpuVecSampler
    smp = GpuVecSampler(EnvCls=SynthAtariEnv, env_args=dict(game=GAME), horizon=HORIZON, n_parallel=16,
                        envs_per=n_env // 32, max_path_length=int(27e3), max_decorrelation_steps=0, device=device)
    state = np.random.get_state()
    smp.initialize(seed=1, discount=0.99, need_extra_obs=True)

    class Served(object):
        recurrent = False
        serves_rows = True
        def reset(self, n_batch): pass                                   # noqa: E301,E704
        def get_action(self, ob): return None, None                      # noqa: E301,E704
    pol = Served()
    smp.policy_init(pol)
    np.random.set_state(state)
    a = smp.env_spec.action_space.n
    prob = torch.full((n_env, a), 1.0 / a, device=device)
    val = torch.zeros(n_env, device=device)
    u = torch.rand(n_env, dtype=torch.float64, device=device)
    per_graph = 20
    ms = graph_time_ms(lambda: _lib.env_step(smp._game, smp._state, smp._rollout, prob, val, u, 1, True, 27000, 0.99,
                                             smp.env.max_start_noops, single_write=True), per_graph=per_graph, replays=5)
    nbytes = n_env * (75520 + 33295 + 4 * a)
    achieved = nbytes / (ms * 1e-3) / 1e9
    traffic = None
    try:
        with open(os.path.join(ROOT, "profiles", "env_step_pmc.json")) as f:
            rec = json.load(f)
        import hashlib
        h = hashlib.sha1()
        for name in ("env.hip", "env_dev.h"):                      # (tools/env_step_pmc.py: env_sources_sha1)
            with open(os.path.join(ROOT, "accel_rl_amd", "csrc", name), "rb") as f:
                h.update(f.read())
        if rec.get("env_hip_sha1") == h.hexdigest():
            traffic = int(rec["hbm_bytes_per_env_step"] * n_env)
    except (OSError, ValueError, KeyError):
        pass
    smp.shutdown()
    return dict(kernel="env_step@sweep (env_step_kernel, %d envs, rollout rows written once)" % n_env, bound="hbm",
                avg_launch_us=round(ms * 1e3, 2), bytes_per_launch=nbytes, achieved=round(achieved, 1), peak=HBM_PEAK_GBS,
                unit="GB/s", frac=round(achieved / HBM_PEAK_GBS, 4), traffic=traffic,
                # what the DRAM really moves (PMC bytes / launch time): the 2 MB raw-frame bank is served from L2 / MALL,
                # so this is BELOW the algorithmic fraction -- the kernel is not HBM-bound (LABNOTES.md, rounds 1-5 section 6, round 4)
                frac_dram=(round(traffic / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if traffic else None),
                timing="%d launches per hipGraph, 5 replays" % per_graph)


def kernel_table(device, sampler, algo, policy, reps=20):
    """Event-timed averages of the step's hand-written kernels at the CONFIG sizes."""
    from accel_rl_amd import _lib
    n, t, a = N_ENVS, HORIZON, sampler.env_spec.action_space.n
    buf = sampler.samples_buf
    rows = []

    def add(name, fn, nbytes, launches_per_step):
        mean_ms, med_ms = event_time_ms(fn, reps)
        gbs = nbytes / (mean_ms * 1e-3) / 1e9
        rows.append(dict(kernel=name, avg_launch_us=round(mean_ms * 1e3, 2),
                         median_launch_us=round(med_ms * 1e3, 2), bytes_per_launch=int(nbytes),
                         achieved_GBs=round(gbs, 1), frac_hbm=round(gbs / HBM_PEAK_GBS, 4),
                         launches_per_step=launches_per_step))

    opt = algo._opt_buf
    lv = torch.zeros(n, device=device)
    add("gae_scan@config", lambda: _lib.gae_scan(buf.rewards, buf.agent_infos["value"], buf.dones, lv,
                                                 0.99, 0.95, n, t, opt["advantages"], opt["returns"]),
        17 * n * t + 4 * n, 1)
    prob = torch.full((n, a), 1. / a, device=device)
    val = torch.zeros(n, device=device)
    u = torch.rand(n, dtype=torch.float64, device=device)
    snap = {k: v.clone() for k, v in sampler._st.items()}
    obs_snap = sampler.step_obs.clone()
    ro, env = sampler._rollout, sampler.env
    # the env side of a step is ONE launch; bytes = SURVEY 8d's algorithmic figure (2 raw frames in, 1 preprocessed
    # frame out = 75 520 B per env-step) -- what the launch really moves is more: + the previous stack read
    # (24 960 B) + the whole stacked observation written (33 280 B, once; twice for policies that do not serve
    # rows of the rollout buffer)
    add("env_step (act + frame + epoch bump, one launch)",
        lambda: _lib.env_step(sampler._game, sampler._state, ro, prob, val, u, 0, True, 27e3, 0.99,
                              env.max_start_noops, single_write=sampler._single_write), n * 75520, t)
    for k, v in snap.items():
        sampler._st[k].copy_(v)
    sampler.step_obs.copy_(obs_snap)
    idx = torch.randperm(n * t, device=device)[:512].to(torch.int32)
    out = torch.empty((512, 4, 104, 80), device=device).contiguous(memory_format=torch.channels_last)
    # (conv 1 reads the u8 observations in place: the gather + scale pass is no longer on the step's path)
    add("gather_scale_obs_nhwc", lambda: _lib.gather_scale_obs_nhwc(buf.observations, idx, out, 1. / 255),
        512 * 33280 * 5, 0 if getattr(policy, "_u8", False) else 8)
    optim = algo.optimizer
    state = [policy.flat_params, optim._slot0, optim._slot1, optim._step_count] + \
        ([optim._step_pp] if optim._step_pp is not None else [])
    saved = [x.clone() for x in state]
    policy.flat_grads.normal_()
    n_upd = optim._n_updates
    host_state = {k: getattr(optim, k, None) for k in ("_hole", "_call_hole", "_hole_count", "_pending_avg")}
    # adam, no clipping: ONE pass over p, g, m, v (read 16 B + written 12 B per parameter; the logged norm's sum
    # of squares rides along)
    add("opt_step (adam + norm partials, one launch)", lambda: optim._apply_update(1.0),
        policy.flat_params.numel() * 28, 8)
    optim._n_updates = n_upd
    for k, v in host_state.items():             # (the learner's captured graph has the call's split layout baked in)
        setattr(optim, k, v)
    for x, s in zip(state, saved):
        x.copy_(s)
    return rows


def mfma_table(device, policy, batch=512, reps=20):
    """The policy's dense contractions (fp32 MFMA implicit GEMM, csrc/mfma_conv.hip) at the PPO
    minibatch: event-timed per call, 2*MACs/time against the fp32 MFMA peak."""
    from accel_rl_amd import _lib
    conv_g, dense_g = policy._layer_geoms(batch)
    ws = policy._conv_ws
    rows = []
    gen = torch.Generator(device=device).manual_seed(3)
    k = 0
    # route of the fp32 contractions (arl_conv_geom.route): 0 = fp32 MFMA chain; 9 / 6 = every fp32 operand split exactly
    # into three bf16 pieces, nine / six piece products per multiply on the bf16 matrix pipe (u8 pixels: one piece,
    # three products) -- the matrix pipe then does `products` MFMA flops per fp32 flop, priced against the bf16 peak
    mode = _lib.conv_precision()
    for name, g in [("conv%d" % (i + 1), g) for i, g in enumerate(conv_g)] + \
                   [("dense%d" % (i + 1), g) for i, g in enumerate(dense_g)]:
        ho, wo = _lib.conv_out_hw(g)
        x = torch.randn(batch, g.in_h, g.in_w, g.in_c, device=device, generator=gen)
        dy = torch.randn(batch, ho, wo, g.out_c, device=device, generator=gen)
        y, dx, dw = torch.empty_like(dy), torch.empty_like(x), torch.empty_like(policy._w[k])
        w, b = policy._w[k], policy._w[k + 1]
        flops = 2.0 * batch * ho * wo * g.out_c * g.kh * g.kw * g.in_c
        calls = [("fwd", lambda: _lib.conv2d_fwd(x, w, b, y, g, True, ws)),
                 ("wgrad", lambda: _lib.conv2d_bwd_weight(dy, x, dw, g, ws))]
        if k == 0 and getattr(policy, "_u8", False):    # conv 1 as the step runs it: from 1280 u8 observations, rows by index
            obs8 = torch.randint(0, 256, (N_ENVS * HORIZON, g.in_c, g.in_h, g.in_w), device=device,
                                 dtype=torch.int32).to(torch.uint8)
            idx8 = torch.randperm(N_ENVS * HORIZON, device=device)[:batch].to(torch.int32)
            folds = _lib.FoldList()

            def wgrad_u8():
                folds.conv2d_u8_bwd_weight(dy, obs8, idx8, 1. / 255, dw, g, ws)
                folds.run()
            name = "conv1(u8 in place)"
            calls = [("fwd", lambda: _lib.conv2d_u8_fwd(obs8, idx8, 1. / 255, w, b, y, g, True)),
                     ("wgrad+fold", wgrad_u8)]
        if k > 0:                                   # the first layer's input needs no gradient
            # as the learner runs it: on the layer's k-contiguous weight copy where the policy keeps one (round 6; the
            # copy itself is one 5 us launch per backward pass for all layers, not counted in this row)
            wt = getattr(policy, "_wt", {}).get(k // 2) if name.startswith("conv") else None
            if wt is not None:
                _lib.conv2d_dgrad_weights([(w, wt, g)])
            calls.append(("dgrad", lambda: _lib.conv2d_bwd_data(dy, w, None, dx, g, wt=wt)))
        for tag, fn in calls:
            iso_ms, _ = event_time_ms(fn, reps)
            mean_ms = graph_time_ms(fn)
            tfs = flops / (mean_ms * 1e-3) / 1e12
            # What bounds the launch is the pipe its MFMAs are issued on: the fp32 MFMA chain (157.3 TF/s) for route 0
            # and for the <= 16-column layers, the bf16 pipe for the split routes -- where one fp32 multiply is
            # `products` bf16 MFMA products, i.e. a ceiling of 2 500 / products TF/s of fp32 flops (278 at nine, 833
            # for u8 pixels).  `frac` is against THAT ceiling; the fp32-MFMA fraction stays as a secondary key.
            products = (1 if mode == 1 else 3 if "u8" in name else mode) if (mode and g.out_c > 16) else 0
            route_peak = MFMA_BF16_PEAK_TFS / products if products else MFMA_F32_PEAK_TFS
            row = dict(kernel="%s %s" % (name, tag), avg_launch_us=round(mean_ms * 1e3, 2),
                       isolated_launch_us=round(iso_ms * 1e3, 2),
                       flops_per_launch=int(flops), achieved_TFs=round(tfs, 1), route_peak_TFs=round(route_peak, 1),
                       frac=round(tfs / route_peak, 4),
                       frac_mfma_f32=round(tfs / MFMA_F32_PEAK_TFS, 4), launches_per_step=8)
            if products:
                row.update(products_per_multiply=products, matrix_pipe_TFs=round(tfs * products, 1),
                           frac_mfma_bf16=round(tfs * products / MFMA_BF16_PEAK_TFS, 4))
            row["_pipe_us_at_peak"] = flops / route_peak / 1e6
            rows.append(row)
        k += 2
    total_us = sum(r["avg_launch_us"] for r in rows)
    total_fl = sum(r["flops_per_launch"] for r in rows)
    # the ceiling of the whole set on the route it runs: total flops / the time its MFMAs take at their pipes' peaks
    route_peak = total_fl / sum(r.pop("_pipe_us_at_peak") for r in rows) / 1e6
    out = dict(bound="mfma", dtype="f32", peak=round(route_peak, 1), unit="TFLOP/s", batch=batch,
               route_peak=round(route_peak, 1), fp32_mfma_peak=MFMA_F32_PEAK_TFS, bf16_mfma_peak=MFMA_BF16_PEAK_TFS,
               route=("fp32 MFMA chain (v_mfma_f32_32x32x2_f32)" if not mode else
                      "fp32 operands split exactly into three bf16 pieces, %d piece products per multiply accumulated in fp32 "
                      "(v_mfma_f32_32x32x16_bf16; u8 pixels are one piece: three products); achieved counts fp32 flops (2 x MACs); "
                      "peak = route_peak = those flops / the time the issued bf16 MFMA products take at the dense bf16 peak "
                      "(2 500 TF/s / products per multiply, flop-weighted over the kernels), frac = achieved / route_peak; "
                      "frac_vs_fp32_mfma = the same flops against the fp32 MFMA peak this route does not use" % mode),
               timing="avg_launch_us: 20 launches back to back in one hipGraph (as the learner runs them); "
                      "isolated_launch_us: one launch between its own pair of events",
               achieved=round(total_fl / total_us / 1e6, 1),
               frac=round(total_fl / total_us / 1e6 / route_peak, 4),
               frac_vs_fp32_mfma=round(total_fl / total_us / 1e6 / MFMA_F32_PEAK_TFS, 4), kernels=rows)
    # The peak above assumes 2.4 GHz.  The clock these kernels actually sustain: per workgroup, shader-cycle
    # counter against the 100 MHz wall clock over a traced conv-2 forward launch that follows 40 untraced ones
    # (arl_dev_conv_trace_buffer, as tools/conv_trace.py).
    try:
        g = conv_g[1] if len(conv_g) > 1 else conv_g[0]
        ho, wo = _lib.conv_out_hw(g)
        x = torch.randn(batch, g.in_h, g.in_w, g.in_c, device=device, generator=gen)
        y = torch.empty(batch, ho, wo, g.out_c, device=device)
        w = torch.randn(g.out_c * g.kh * g.kw * g.in_c, device=device, generator=gen)
        clocks = []
        tr = torch.zeros(8192 * 8, dtype=torch.int64, device=device)
        for _ in range(5):
            tr.zero_()
            for _ in range(40):                     # the traced launch runs at the END of a busy stretch
                _lib.conv2d_fwd(x, w, None, y, g, True, ws)
            _lib.load().arl_dev_conv_trace_buffer(tr.data_ptr())
            _lib.conv2d_fwd(x, w, None, y, g, True, ws)
            _lib.load().arl_dev_conv_trace_buffer(None)
            torch.cuda.synchronize()
            t = tr.cpu().numpy().reshape(-1, 8)
            t = t[t[:, 0] != 0]
            if len(t):          # per workgroup: shader cycles / (100 MHz ticks * 10 ns)
                clocks.append(float(np.median((t[:, 3] - t[:, 0]) / np.maximum(t[:, 5] - t[:, 4], 1) / 10.0)))
        clk = float(np.median(clocks))
        peak_clk = route_peak * clk / 2.4
        out.update(sustained_clock_ghz=round(clk, 3), peak_at_sustained_clock=round(peak_clk, 1),
                   frac_at_sustained_clock=round(out["achieved"] / peak_clk, 4))
    finally:
        _lib.load().arl_dev_conv_trace_buffer(None)
    return out


ROUTE_NAMES = {9: "split9", 6: "split6", 0: "fp32_mfma", 1: "bf16_operands"}
SPEC1_LAYERS = [("conv1", 104, 80, 4, 32, 8, 4, 0), ("conv2", 25, 19, 32, 64, 4, 2, 1),
                ("conv3", 12, 9, 64, 64, 3, 1, 1), ("dense1", 1, 1, 6912, 512, 1, 1, 0)]


def route_accuracy(device, batch=48, modes=(9, 6, 0, 1), seed=3):
    """Error of every contraction kernel of the spec-1 network (config 2's layer shapes) on each arithmetic route
    (arl_conv_geom.route: nine / six exact bf16-split products, fp32 MFMA chain, and the labelled reduced-precision
    option -- operands rounded to bf16, one product) against a FLOAT64 contraction of the
    same inputs: {layer: {pass: {"rms_err_vs_f64": {route: rms(err) / rms(ref)}, "max_err_vs_f64": {route: max|err| /
    max|ref|}}}}.  The float64 reference is torch's (ATen) convolution -- a checker, not the product path.  The batch
    is small because the float64 reference is slow; the errors do not depend on it (every output is its own dot
    product)."""
    import torch.nn.functional as F
    from accel_rl_amd import _lib
    ws = _lib.conv_workspace(device)
    gen = torch.Generator(device=device).manual_seed(seed)
    was, cudnn_was = _lib.conv_precision(), torch.backends.cudnn.enabled
    torch.backends.cudnn.enabled = False          # (ATen's own convolution: no MIOpen search on a fresh box)
    table = {}

    def err(got, want):
        d = (got.double() - want).abs()
        return (d.max().item() / max(want.abs().max().item(), 1e-30),
                (d.pow(2).mean().sqrt() / want.pow(2).mean().sqrt()).item())
    try:
        for name, h, w, c, k, ks, st, p in SPEC1_LAYERS:
            geom0 = _lib.conv_geom(batch, h, w, c, k, ks, ks, st, p, p)
            ho, wo = _lib.conv_out_hw(geom0)
            x = torch.randn(batch, h, w, c, device=device, generator=gen).relu()
            wt = torch.randn(k, ks, ks, c, device=device, generator=gen) / np.sqrt(ks * ks * c)
            bias = torch.randn(k, device=device, generator=gen)
            dy = torch.randn(batch, ho, wo, k, device=device, generator=gen)
            y, dx, dw = torch.empty(batch, ho, wo, k, device=device), torch.empty_like(x), torch.empty_like(wt)
            u8 = name == "conv1"
            obs = torch.randint(0, 256, (batch, c, h, w), device=device, dtype=torch.int32, generator=gen).to(torch.uint8) if u8 else None
            w8 = wt.permute(0, 3, 1, 2).contiguous() if u8 else None
            dw8 = torch.empty_like(w8) if u8 else None
            dyd = dy.double().permute(0, 3, 1, 2)
            ref = {}
            if u8:              # conv 1 as the step runs it: straight from the u8 observations, no data gradient
                o8r, w8r = (obs.double() / 255.0).requires_grad_(), w8.double().requires_grad_()
                out8 = F.conv2d(o8r, w8r, None, stride=st)
                ref["fwd"] = (out8 + bias.double().view(1, -1, 1, 1)).permute(0, 2, 3, 1).detach()
                ref["wgrad"] = torch.autograd.grad(out8, w8r, dyd)[0]
            else:
                xr, wr = x.double().permute(0, 3, 1, 2).requires_grad_(), wt.double().permute(0, 3, 1, 2).requires_grad_()
                out = F.conv2d(xr, wr, None, stride=st, padding=p)
                gx, gw = torch.autograd.grad(out, (xr, wr), dyd)
                ref = dict(fwd=(out + bias.double().view(1, -1, 1, 1)).permute(0, 2, 3, 1).detach(),
                           dgrad=gx.permute(0, 2, 3, 1), wgrad=gw.permute(0, 2, 3, 1))
            table[name] = {op: dict(rms_err_vs_f64={}, max_err_vs_f64={}) for op in ref}
            for mode in modes:
                _lib.set_conv_precision(mode)
                geom = _lib.with_route(geom0)
                if u8:
                    folds, db = _lib.FoldList(), torch.empty(k, device=device)

                    def u8w():
                        folds.conv2d_u8_bwd_weight(dy, obs, None, 1.0 / 255.0, dw8, geom, ws, dbias=db)
                        folds.run()
                    ops = dict(fwd=(lambda: _lib.conv2d_u8_fwd(obs, None, 1.0 / 255.0, w8, bias, y, geom, False), y),
                               wgrad=(u8w, dw8))
                else:
                    ops = dict(fwd=(lambda: _lib.conv2d_fwd(x, wt, bias, y, geom, False, ws), y),
                               dgrad=(lambda: _lib.conv2d_bwd_data(dy, wt, None, dx, geom), dx),
                               wgrad=(lambda: _lib.conv2d_bwd_weight(dy, x, dw, geom, ws), dw))
                for op, (fn, out_t) in ops.items():
                    out_t.fill_(float("nan"))
                    fn()
                    torch.cuda.synchronize()
                    mx, rms = err(out_t, ref[op])
                    table[name][op]["rms_err_vs_f64"][ROUTE_NAMES[mode]] = float("%.4g" % rms)
                    table[name][op]["max_err_vs_f64"][ROUTE_NAMES[mode]] = float("%.4g" % mx)
    finally:
        _lib.set_conv_precision(was)
        torch.backends.cudnn.enabled = cudnn_was
    return table


def alt_routes(device, steps, warmup, priming, modes=(6, 0, 1)):
    """The SAME workload, steps and timed region as the headline on the other arithmetic routes of the fp32
    contractions (arl_conv_geom.route): a second and a third runner built from scratch with the route stamped into
    every layer geometry, primed, warmed up and timed exactly like `value`.  The default route is not changed."""
    from accel_rl_amd import _lib
    out, was = {}, _lib.conv_precision()
    try:
        for mode in modes:
            _lib.set_conv_precision(mode)
            runner, sampler, algo, policy = build_workload(device, 0, 0, 1, GAME, True)
            itr = 0
            for _ in range(priming + warmup):
                one_step(itr, sampler, algo)
                itr += 1
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                one_step(itr, sampler, algo)
                itr += 1
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            out[ROUTE_NAMES[mode]] = dict(value=round(steps * N_ENVS * HORIZON / dt, 1), ms_per_step=round(dt / steps * 1e3, 4),
                                          steps=steps, products_per_multiply=mode if mode else 1,
                                          fp32_contraction=mode != 1)
            runner.shutdown()
            algo._graph = algo._graph_out = None
            sampler._graph = None
            del runner, sampler, algo, policy
            import gc
            gc.collect()
            torch.cuda.synchronize()
    finally:
        _lib.set_conv_precision(was)
    out["note"] = ("same workload, steps, warm-up and timed region as `value`, which runs on split9 (the default, unchanged): "
                   "split6 drops the three piece products below 2^-24 of a product (m*l, l*m, l*l), fp32_mfma is the "
                   "v_mfma_f32_32x32x2_f32 chain; bf16_operands (ARL_CONV_ROUTE_BF16) is the labelled reduced-precision option "
                   "(SURVEY 8d: fp32 default, bf16 optional): operands rounded to bf16, one product, fp32 accumulation -- NOT an "
                   "fp32 contraction, never `value`; per-layer errors of all four against float64 are in `accuracy`")
    return out


def _served(device, policy):
    """The action server of the CPU baselines: H2D of the u8 observations, the SAME torch policy
    on the GPU, D2H, categorical sampling on the host (the reference's master does exactly this
    with Theano, overlap/sampler.py:129-145, rllab/misc/special.py:22-27)."""
    from oracle import ref_port as P

    class Served(object):
        def get_actions(self, obs):
            prob, value = policy.prob_value(torch.from_numpy(np.ascontiguousarray(obs)).to(device))
            prob, value = prob.cpu().numpy(), value.cpu().numpy()
            return P.sample_actions(prob, np.random.rand(len(prob))), dict(prob=prob, value=value)
    return Served()


def _host_cpu():
    """(model string, physical cores, logical cores) of this box from /proc/cpuinfo."""
    model, cores = "unknown", set()
    try:
        phys = core = None
        with open("/proc/cpuinfo") as f:
            for line in f:
                key, _, val = line.partition(":")
                key, val = key.strip(), val.strip()
                if key == "model name":
                    model = val
                elif key == "physical id":
                    phys = val
                elif key == "core id":
                    core = val
                elif not key and phys is not None:
                    cores.add((phys, core))
                    phys = core = None
        if phys is not None:
            cores.add((phys, core))
    except OSError:
        pass
    logical = os.cpu_count() or 1
    return model, (len(cores) or max(1, logical // 2)), logical


def _windows(one_batch, n_windows, seconds, min_batches=3, batch_times=None):
    """env-steps/s of `one_batch()` over n_windows back-to-back windows; returns (rates, batches, total time).
    batch_times (optional list): every batch's duration in seconds is appended."""
    one_batch()                                      # warm-up batch
    rates, total_b, total_t = [], 0, 0.0
    for _ in range(n_windows):
        t0, batches = time.time(), 0
        t1 = t0
        while t1 - t0 < seconds or batches < min_batches:
            one_batch()
            batches += 1
            t2 = time.time()
            if batch_times is not None:
                batch_times.append(t2 - t1)
            t1 = t2
        dt = t1 - t0
        rates.append(batches * N_ENVS * HORIZON / dt)
        total_b += batches
        total_t += dt
    return rates, total_b, total_t


def start_cpu_pool():
    """Fork the CPU baseline's worker processes BEFORE this process touches HIP (a forked child of an
    initialised HIP runtime is not safe; the reference forks its workers before Theano's first call too).
    They sit on a barrier, using no CPU, until cpu_baseline() runs."""
    from oracle.cpu_sampler_mp import CpuSamplerMP
    _, physical, _ = _host_cpu()
    half = N_ENVS // 2
    # Pool shapes (n_parallel workers per alternating group x envs per worker): the reference launcher's rule -- one
    # worker per simulation core, i.e. the largest divisor of N/2 that fits the physical cores minus one
    # (scripts/example/example_train_ppo.py:30-41, scripts/launching/affinities.py:63-69) -- and two smaller pools
    # (2 x 16 x 8, 2 x 32 x 4): the launcher assumes an exclusive host, and on a shared one a straggler per
    # step-barrier sets the pace of a 128-process pool.  cpu_baseline() times them all and quotes the best.
    top = max(d for d in range(1, half + 1) if half % d == 0 and d <= max(1, physical - 1))
    pools = []
    for n_par in sorted(set(d for d in (16, 32, top) if d <= top)):
        smp = CpuSamplerMP(GAME, HORIZON, n_par, half // n_par, max_path_length=int(27e3), mid_batch_reset=True,
                           start_method="fork", pin=True)
        smp.initialize(12346, discount=0.99, master_rng=np.random.RandomState(12345))
        pools.append(smp)
    return pools


def cpu_baseline(device, policy, pools, sampler, algo, itr0, seconds=4.0, n_windows=3, gae_lambda=0.95):
    """The reference's CPU sampler restated (oracle/, pinned to the real reference by the golden
    fixtures) on this box's host cores, on the same workload; every figure is the MEDIAN of its windows:
      * `value`: multi-process, as the reference runs it -- master + 2*n_parallel workers in two alternating
        groups (oracle/cpu_sampler_mp.py) --, rollout + process_samples, no learner: the BEST of the pool shapes of
        start_cpu_pool() (two windows each; the sweep is in `pool_sweep`, the launcher's own shape is its last row);
      * `whole_loop`: the same sampler followed, serially as the reference's runner does it
        (runners/accel_rl.py:28-37), by the SAME device learner the headline `value` runs (batch copied H2D,
        algo.optimize_policy, wait) -- the like-for-like baseline of the headline number;
      * `single_core`: the sampler's arithmetic walked by ONE process (oracle/ref_port.CpuSamplerPort)."""
    from oracle import ref_port as P
    served = _served(device, policy)
    model, physical, logical = _host_cpu()
    half = N_ENVS // 2

    def sample_and_process(s):
        buf, _ = s.obtain_samples(served)
        lv = policy.value(torch.from_numpy(buf["extra_observations"]).to(device)).cpu().numpy()
        P.process_samples(buf["rewards"].reshape(N_ENVS, HORIZON), buf["dones"].reshape(N_ENVS, HORIZON),
                          buf["value"].reshape(N_ENVS, HORIZON), lv, None, 0.99, gae_lambda)

    dst = sampler.samples_buf
    pairs = lambda buf: (                                            # noqa: E731
        (dst.observations, buf["observations"]), (dst.rewards, buf["rewards"]), (dst.dones, buf["dones"]),
        (dst.env_infos["raw_reward"], buf["raw_reward"]), (dst.env_infos["need_reset"], buf["need_reset"]),
        (dst.actions, buf["actions"]), (dst.agent_infos["prob"], buf["prob"]),
        (dst.agent_infos["value"], buf["value"]), (dst.extra_observations, buf["extra_observations"]))
    itr = [itr0]

    def whole_loop():
        buf, _ = best.obtain_samples(served)
        for d, h in pairs(buf):
            d.copy_(torch.from_numpy(np.ascontiguousarray(h)).view(d.dtype).reshape(d.shape))
        algo.optimize_policy(itr[0], dst)
        itr[0] += 1
        torch.cuda.synchronize()

    np.random.seed(12345)
    load0 = os.getloadavg()[0]                       # the host's 1-minute load before the pools start working
    sweep, best = [], None
    try:
        for smp in pools:                            # two windows per shape; the others sit on their barriers
            bt_k = []
            r_k, b_k, dt_k = _windows(lambda: sample_and_process(smp), 2, seconds, batch_times=bt_k)
            sweep.append(dict(n_parallel=smp.n_parallel, envs_per=half // smp.n_parallel, processes=2 * smp.n_parallel,
                              windows=[round(r, 1) for r in r_k], median=round(float(np.median(r_k)), 1),
                              batch_ms=[round(1e3 * float(x), 2) for x in np.percentile(bt_k, [10, 50, 90])]))
            if best is None or np.median(r_k) > best_rate:
                best, best_rate, rates, batches, dt, bt = smp, np.median(r_k), r_k, b_k, dt_k, bt_k
        n_par = best.n_parallel
        loop_rates, loop_batches, loop_dt = _windows(whole_loop, n_windows, seconds * 0.75)
    finally:
        for smp in pools:
            smp.shutdown()
    one = P.CpuSamplerPort(GAME, HORIZON, 16, N_ENVS // 32, max_path_length=int(27e3), mid_batch_reset=True)
    np.random.seed(12345)
    one.initialize(12346, discount=0.99)
    one_rates, b1, dt1 = _windows(lambda: sample_and_process(one), n_windows, seconds * 0.4)
    med = lambda x: float(np.median(x))                              # noqa: E731
    return dict(value=round(med(rates), 1), unit="env-steps/s", cores=n_par + 1, kind="port",
                windows=[round(r, 1) for r in rates], window_spread=round(max(rates) / min(rates), 2),
                pool_sweep=sweep, whole_loop=round(med(loop_rates), 1),
                whole_loop_windows=[round(r, 1) for r in loop_rates],
                single_core=round(med(one_rates), 1), cpu_model=model, physical_cores=physical,
                logical_cores=logical,
                # The GPU boxes share their host: other tenants' load moves the CPU figure by 2-3x inside one run.
                # batch_ms = 10th / 50th / 90th percentile of the sampler's batch time over all windows (the 10th
                # is the undisturbed sampler: 1280 env-steps per batch); host_load_1min = load average before the
                # pool started.
                batch_ms=[round(1e3 * float(x), 2) for x in np.percentile(bt, [10, 50, 90])],
                undisturbed=round(N_ENVS * HORIZON / float(np.percentile(bt, 10)), 1),
                host_load_1min=round(load0, 1),
                sample="best of %d pool shapes, each the median of 2 windows (chosen: %d batches, %.1f s) of the same workload's rollout + "
                       "process_samples (%d envs x %d steps): numpy restatement of the reference sampler / "
                       "AtariEnv / GAE, master + %d worker processes (2 alternating groups x %d, %d envs each, "
                       "pinned), actions served by the same policy on the GPU, no learner update; whole_loop = "
                       "the same sampler + H2D of the batch + the device learner of `value`, serial "
                       "(%d batches, %.1f s); single_core = the sampler walked by one process (%d batches)" %
                       (len(sweep), batches, dt, N_ENVS, HORIZON, 2 * n_par, n_par, half // n_par, loop_batches, loop_dt, b1))


def replay_pmc_traffic(batch):
    """HBM bytes per arl_replay_extract launch from the committed rocprofv3 --pmc passes (tools/replay_pmc.sh); None
    when the record is missing, of another batch, or of another version of replay.hip."""
    import hashlib
    try:
        with open(os.path.join(ROOT, "profiles", "replay_extract_pmc.json")) as f:
            rec = json.load(f)
        with open(os.path.join(ROOT, "accel_rl_amd", "csrc", "replay.hip"), "rb") as f:
            sha = hashlib.sha1(f.read()).hexdigest()
        if rec.get("batch") != batch or rec.get("replay_hip_sha1") != sha:
            return None
        return rec["hbm_bytes_per_launch"]
    except (OSError, ValueError, KeyError):
        return None


def replay_roofline(device, algo, batch=4096):
    """arl_replay_extract (csrc/replay.hip; extract_batch / extract_observations, frame.py:69-90) at a bandwidth-bound
    batch of the 1M-transition store: per sampled transition two stacked observations (state and the state
    reward_horizon later), each 33 280 B read from the frame ring and 33 280 B written into the minibatch, + 6 B of
    scalars -- the algorithmic bytes; timed as 20 launches per hipGraph with HIP events on the launch stream."""
    from accel_rl_amd import _lib
    buf = algo.replay_buffer
    gen = torch.Generator(device=device).manual_seed(11)
    n_env = buf.frames.shape[0]
    stack = int(np.prod(buf.frames.shape[2:])) * buf.num_img_obs
    e_idx = torch.randint(0, n_env, (batch,), dtype=torch.int32, device=device, generator=gen)
    s_idx = torch.randint(0, buf.env_replay_size - 8, (batch,), dtype=torch.int32, device=device, generator=gen)
    shape = (batch, buf.num_img_obs) + tuple(buf.frames.shape[2:])
    obs, nxt = (torch.empty(shape, dtype=torch.uint8, device=device) for _ in range(2))
    a, r, tm = (torch.empty(batch, dtype=torch.uint8, device=device), torch.empty(batch, device=device),
                torch.empty(batch, dtype=torch.uint8, device=device))
    per_graph = 20
    ms = graph_time_ms(lambda: _lib.replay_extract(buf._rb, e_idx, s_idx, obs, nxt, a, r, tm), per_graph=per_graph, replays=5)
    nbytes = batch * (2 * 2 * stack + 6)
    achieved = nbytes / (ms * 1e-3) / 1e9
    return dict(bound="hbm", kernel="extract_kernel (arl_replay_extract), %d transitions = %d stacked observations out of "
                                    "%.1f GB of frames" % (batch, 2 * batch, buf.frames.numel() / 1e9),
                achieved=round(achieved, 1), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(achieved / HBM_PEAK_GBS, 4),
                traffic=replay_pmc_traffic(batch), bytes_per_launch=nbytes, bytes_per_stack=2 * stack, avg_launch_us=round(ms * 1e3, 2),
                timing="%d launches per hipGraph, 5 replays" % per_graph)


def replay_cpu_baseline(n_env, horizon, reward_horizon, batch, updates_per_step, seconds=12.0):
    """The replay side of config 5 on ONE host core: the oracle's numpy restatement of the reference's frame-dedup
    replay buffer + parted sum tree (oracle/replay_port.py, pinned to the reference's own classes by G11 / G12) doing
    what one bench step asks of it -- append of n_env x horizon env-steps with the n-step back-fill, tree advance, then
    `updates_per_step` x (sample_n of `batch` distinct leaves, importance weights, extract_batch, priority write-back).
    No emulator, no network: the reference's learner (Theano) is absent.  The store is 400 states per environment
    instead of 3 908 (host memory; the per-operation cost does not depend on the capacity)."""
    from oracle import replay_port as R
    frame = (104, 80)
    size = n_env * 400
    rp = R.ReplayPort(n_env, 4, frame, size, reward_horizon, horizon, 0.99)
    tree = R.SumTreePort(rp.S, n_env, zeros_forward=4, zeros_backward=reward_horizon, default_value=1.0,
                         n_advance=horizon)
    rs = np.random.RandomState(0)
    obs = rs.randint(0, 256, size=(n_env, horizon, 4) + frame, dtype=np.uint8)
    acts = rs.randint(0, 18, size=(n_env, horizon)).astype(np.uint8)

    def step():
        rews = rs.randn(n_env, horizon).astype(np.float32)
        dones = rs.rand(n_env, horizon) < 0.02
        rp.append(obs, acts, rews, dones)
        tree.advance()
        for _ in range(updates_per_step):
            e, st, probs = tree.sample_n(batch, rs)
            R.importance_weights(probs, 0.4)
            rp.extract_batch(e, st)
            tree.update_last(rs.rand(batch) + 0.1)
    for _ in range(60):                      # fill enough of the ring for sampling (and warm the caches)
        rews = rs.randn(n_env, horizon).astype(np.float32)
        rp.append(obs, acts, rews, rs.rand(n_env, horizon) < 0.02)
        tree.advance()
    step()
    t0, n = time.time(), 0
    while time.time() - t0 < seconds or n < 3:
        step()
        n += 1
    dt = time.time() - t0
    return dict(value=round(n * n_env * horizon / dt, 1), unit="env-steps/s", cores=1, kind="port",
                sample="%d bench steps (%.1f s) of the replay side only: append of %d x %d env-steps + n-step back-fill + tree "
                       "advance + %d x (sample_n(%d) + importance weights + extract_batch + priority write-back) on the "
                       "oracle's numpy port (400 states per env instead of 3 908); no emulator, no learner" %
                       (n, dt, n_env, horizon, updates_per_step, batch))


def _catdqn_run(device, batch, steps, warmup):
    """One Categorical-DQN workload built from scratch and timed: (runner, algo, env-steps/s, ms per step)."""
    from accel_rl_amd.algos.dqn.cat_dqn import CategoricalDQN
    from accel_rl_amd.envs.synthetic_atari import SynthAtariEnv
    from accel_rl_amd.policies.atari_cnn_specs import cnn_specs
    from accel_rl_amd.policies.dqn.atari_cat_dqn_policy import AtariCatDqnPolicy
    from accel_rl_amd.runners.accel_rl import AccelRLEval
    from accel_rl_amd.sampler.gpu_sampler_with_eval import GpuVecEvalSampler
    n_env, horizon = 256, 4
    sampler = GpuVecEvalSampler(eval_steps=12800, eval_envs_per=1, EnvCls=SynthAtariEnv, env_args=dict(game="seaquest"),
                                horizon=horizon, n_parallel=16, envs_per=n_env // 32, max_path_length=int(27e3),
                                max_decorrelation_steps=0, device=device)
    algo = CategoricalDQN(batch_size=batch, min_steps_learn=40 * n_env * horizon, replay_size=int(1e6), training_intensity=8,
                          reward_horizon=3, prioritized_replay=True, double_dqn=True)
    policy = AtariCatDqnPolicy(**cnn_specs[1])
    runner = AccelRLEval(algo=algo, policy=policy, sampler=sampler, n_steps=1e9, seed=0, eval_interval_steps=1e8)
    runner.startup()
    itr = 0
    for _ in range(40 + warmup):                    # 40 sampling-only steps fill the replay, then the warm-up
        samples, _ = sampler.obtain_samples(itr)
        algo.optimize_policy(itr, samples)
        itr += 1
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        samples, _ = sampler.obtain_samples(itr)
        algo.optimize_policy(itr, samples)
        itr += 1
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    return runner, algo, round(steps * n_env * horizon / el, 1), round(el / steps * 1e3, 4)


def catdqn_main(args):
    """BASELINE config 5 (not the headline metric): Categorical DQN "seaquest", 1M-transition device replay
    (prioritized), 256 envs x horizon 4, spec-1 trunk, reward horizon 3, training intensity 8 -- at the reference's own
    minibatch of 32 rows (accel_rl/algos/dqn/dqn.py:18, used at :88-94,180), i.e. 256 updates per 1024-transition batch;
    `alt` re-times the same workload at 512 rows x 16 updates (the same rows per sampled transition)."""
    from accel_rl_amd.util import logger
    import __graft_entry__
    __graft_entry__.build()
    logger.set_quiet(True)
    device = torch.device("cuda", 0)
    torch.cuda.set_device(device)
    n_env, horizon = 256, 4
    runner, algo, value, ms = _catdqn_run(device, args.dqn_batch, args.steps, args.warmup)
    line = {"metric": "env-steps/sec (whole node), Categorical-DQN Seaquest, 1M-transition replay (BASELINE config 5, "
                      "not the headline metric)",
            "value": value, "unit": "env-steps/s", "n_gpus": 1,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "Cat-DQN seaquest, %d envs x horizon %d, spec-1 trunk, 51 atoms, prioritized replay "
                                   "of %d transitions (frames %.1f GB in HBM), n-step 3, double DQN, minibatch %d x %d "
                                   "updates per step (training intensity 8), adam" %
                                   (n_env, horizon, algo.replay_buffer.env_replay_size * n_env,
                                    algo.replay_buffer.frames.numel() / 1e9, args.dqn_batch, algo._updates_per_optimize),
                       "minibatch": args.dqn_batch, "updates_per_step": algo._updates_per_optimize}}
    if not args.no_roofline:
        line["roofline"] = replay_roofline(device, algo)
    if not args.no_cpu_baseline:
        line["cpu_baseline"] = replay_cpu_baseline(n_env, horizon, 3, args.dqn_batch, algo._updates_per_optimize)
    if not args.no_alt_routes:
        other = 512 if args.dqn_batch != 512 else 32
        runner.shutdown()
        algo._graph = None
        del runner, algo
        import gc
        gc.collect()
        torch.cuda.synchronize()
        runner, algo, v2, ms2 = _catdqn_run(device, other, args.steps, args.warmup)
        line["alt"] = {"minibatch": other, "updates_per_step": algo._updates_per_optimize, "value": v2, "ms_per_step": ms2,
                       "note": "the same workload at another replay minibatch (same training intensity: rows trained per "
                               "sampled transition); the reference's own default is 32 (dqn.py:18)"}
    print(json.dumps(line), flush=True)


def multi_gpu_diagnostics(device, world, rank, algo, policy, sampler, samples, timed, reps, itr, t_step, t_roll, t_learn,
                          tick=lambda *a: None):
    """What the N > 1 line says about itself (every rank computes, rank 0 prints):
      per_rank_ms                         min / max over the ranks of each rank's own step, rollout and learner time
                                          (a straggler shows as max >> min; the headline takes the max);
      params_bit_identical_across_ranks   all-reduce(MIN) == all-reduce(MAX) of a 64-bit checksum of the parameter
                                          bucket after the timed region: the synchronous update's invariant
                                          (sync_ppo_optimizer.py:27-34: same averaged gradient, same local update);
      graph_captured                      the learner of the timed region ran as ONE hipGraph with the collectives
                                          inside (False: every rank fell back to eager minibatches together --
                                          SyncPpoOptimizer.ranks_agree -- or --no-graph);
      allreduce_exposed_ms                learner time with the collectives minus the SAME learner with them left out
                                          (re-captured; measured last, the ranks' parameters diverge from there on):
                                          what the all-reduces add to a step after the overlap with the conv backward.
    All collectives below are issued by every rank in the same order."""
    opt = algo.optimizer
    f64 = lambda *x: torch.tensor(list(x), dtype=torch.float64, device=device)          # noqa: E731
    mine = f64(t_step, t_roll, t_learn) * 1e3
    lo, hi = mine.clone(), mine.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    words = policy.flat_params.view(torch.int32).to(torch.int64)
    pos = torch.arange(1, words.numel() + 1, device=device, dtype=torch.int64)
    check = torch.stack([words.sum(), (words * (pos % 65521)).sum()])                     # position-sensitive
    cmin, cmax = check.clone(), check.clone()
    dist.all_reduce(cmin, op=dist.ReduceOp.MIN)
    dist.all_reduce(cmax, op=dist.ReduceOp.MAX)
    out = {"per_rank_ms": {k: [round(float(lo[i]), 4), round(float(hi[i]), 4)]
                           for i, k in enumerate(("step", "rollout", "learner"))},
           "params_bit_identical_across_ranks": bool(torch.equal(cmin, cmax)),
           "param_checksum": [int(x) for x in cmin.tolist()],
           "graph_captured": bool(getattr(algo, "_graph", None) is not None and opt.graph_ready()),
           "overlapped_allreduce": bool(getattr(opt, "_overlap_allreduce", False)),
           "host_cpus_of_rank0": len(os.sched_getaffinity(0)),
           "allreduce_bytes_per_update": int(policy.flat_grads.numel() * 4)}
    # The same learner without its collectives: two eager calls, a re-capture, then timed.  It runs LAST, issues no
    # collective of its own while it can fail, and cannot take the line with it: a rank that fails says so in ONE
    # all-reduce that every rank reaches exactly once, and the figure is then reported as null.  The graph of the timed
    # region stays alive (a graph that holds RCCL nodes is not destroyed while its communicator lives).
    out["learner_without_collectives_ms"] = out["allreduce_exposed_ms"] = None
    if os.environ.get("ARL_BENCH_NO_EXPOSED") == "1":
        return out
    keep = (getattr(algo, "_graph", None), getattr(algo, "_graph_out", None))          # noqa: F841
    t_quiet, failure = 0.0, None
    try:
        opt._elide_collective = True
        algo._graph, algo._graph_out, algo._warm_calls = None, None, 0
        for i in range(3):
            tick("diagnostics: learner without collectives (re-capture)")
            algo.optimize_policy(itr + i, samples)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(reps):
            tick("diagnostics: learner without collectives")
            algo.optimize_policy(itr + 3 + i, samples)
        torch.cuda.synchronize()
        t_quiet = (time.perf_counter() - t0) / reps
    except Exception as e:                  # noqa: BLE001 -- whatever it is, the line must still be printed
        failure = repr(e)
    quiet = f64(-1.0 if failure else t_quiet * 1e3)
    worst = quiet.clone()
    dist.all_reduce(quiet, op=dist.ReduceOp.MIN)          # (-1 from any rank: the figure is void)
    dist.all_reduce(worst, op=dist.ReduceOp.MAX)
    if float(quiet[0]) >= 0:
        out["learner_without_collectives_ms"] = round(float(worst[0]), 4)
        out["allreduce_exposed_ms"] = round(float(hi[2]) - float(worst[0]), 4)
    elif failure:
        out["exposed_error"] = failure[:200]
    return out


def _stall_marker():
    """The watchdog's note for the NEXT run of THIS build by THIS user on this node: keyed on the built library's hash and
    the uid, so that another checkout, another user or a rebuilt tree (the cause fixed) starts with captured collectives."""
    import hashlib
    tag = "nolib"
    try:
        with open(os.path.join(ROOT, "accel_rl_amd", "libaccel_rl_hip.so"), "rb") as f:
            tag = hashlib.sha256(f.read()).hexdigest()[:12]
    except OSError:
        pass
    return os.path.join(os.environ.get("TMPDIR", "/tmp"), "arl_bench_sync_graph_stalled.%d.%s" % (os.getuid(), tag))


STALL_MARKER = _stall_marker()


def stalled_before(max_age_s=6 * 3600):
    """An earlier multi-rank run of this build on this node stalled with its collectives captured (the watchdog's note)."""
    try:
        with open(STALL_MARKER) as f:
            return time.time() - int(f.read().split()[0]) < max_age_s
    except (OSError, ValueError, IndexError):
        return False


def clear_stall_marker():
    """A run with captured collectives completed: whatever stalled before does not any more."""
    try:
        os.remove(STALL_MARKER)
    except OSError:
        pass


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves, one process per GPU, as the
    reference's runner forks its n-1 workers from one process (accel_rl/runners/multigpu_rl_base.py:20-45): this command
    line again under torch.distributed.run (rendezvous on 127.0.0.1, a free port) -- as a SUPERVISED child.  The
    learner of N > 1 is one hipGraph with RCCL's all-reduces captured inside; a node on which that capture
    throws is handled inside the ranks (`ranks_agree`), one on which it HANGS ends in the ranks' watchdog (exit 3) and no
    JSON line.  In that case -- any non-zero exit, or no line -- the job is run ONCE more with the collectives eager
    (`ARL_SYNC_GRAPH=0`: the same sums in the same order, sync_ppo_optimizer.py:27-34,61-71) and the line says so in
    `graph_fallback`.  Returns the exit code."""
    import signal
    import socket
    import subprocess

    def attempt(extra_env):
        sock = socket.socket()
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
        sock.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ)
        env.update(extra_env)
        sys.stdout.flush()
        child = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env, start_new_session=True)
        limit = float(os.environ.get("ARL_BENCH_ATTEMPT_S", "1500"))
        import threading

        def reaper():                       # belt and braces behind the ranks' own watchdogs
            try:
                os.killpg(child.pid, signal.SIGKILL)
            except OSError:
                pass
        timer = threading.Timer(limit, reaper)
        timer.daemon = True
        timer.start()
        lines = []
        for ln in child.stdout:
            if ln.startswith('{"metric"'):
                lines.append(ln.rstrip("\n"))          # held back: the supervisor prints ONE line at the end
            else:
                sys.stdout.write(ln)
                sys.stdout.flush()
        rc = child.wait()
        timer.cancel()
        return rc, lines

    rc, lines = attempt({})
    if rc == 0 and len(lines) == 1:
        print(lines[0], flush=True)
        return 0
    reason = ("exit code %d" % rc) if rc else ("%d JSON lines" % len(lines))
    if os.environ.get("ARL_SYNC_GRAPH") == "0" or os.environ.get("ARL_BENCH_NO_RETRY") == "1":
        sys.stderr.write("bench.py: the %d-rank run failed (%s)\n" % (n, reason))
        return rc or 1
    sys.stderr.write("bench.py: the %d-rank run failed (%s); retrying once with eager collectives (ARL_SYNC_GRAPH=0)\n" % (n, reason))
    sys.stderr.flush()
    rc, lines = attempt({"ARL_SYNC_GRAPH": "0", "ARL_BENCH_NO_EXPOSED": "1",
                         "ARL_BENCH_GRAPH_FALLBACK": "eager after " + reason})
    if rc == 0 and len(lines) == 1:
        print(lines[0], flush=True)
        return 0
    sys.stderr.write("bench.py: the eager retry failed too (%s)\n" % (("exit code %d" % rc) if rc else "no JSON line"))
    return rc or 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replay")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-alt-routes", action="store_true",
                    help="skip the re-timing of the headline on the six-product and fp32-MFMA routes and the accuracy table")
    ap.add_argument("--roofline-log2", type=int, default=26)
    ap.add_argument("--roofline-only", action="store_true",
                    help="only the GAE-scan roofline leg (used for the rocprofv3 --pmc passes)")
    ap.add_argument("--suite", action="store_true", help="BASELINE config 4: one game per rank")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak: 256 envs and minibatch 512 PER GPU (default); strong: --total-envs and a global "
                         "minibatch of 4096 rows split over the GPUs (SURVEY 8d)")
    ap.add_argument("--total-envs", type=int, default=2048, help="--scaling strong: environments of the whole job")
    ap.add_argument("--dqn-batch", type=int, default=32,
                    help="catdqn workload: replay minibatch size (default: the reference's, accel_rl/algos/dqn/dqn.py:18)")
    ap.add_argument("--workload", choices=["ppo256", "a2c1024", "catdqn"], default="ppo256",
                    help="ppo256 = BASELINE config 2 (the metric's config, default); a2c1024 = config 3 "
                         "(A2C, 1024 envs, 5-step returns, spec-0 CNN, one rmsprop step per batch)")
    args = ap.parse_args()

    global N_ENVS, CNN_SPEC, MINIBATCH
    if args.workload == "a2c1024":
        N_ENVS, CNN_SPEC = 1024, 0
    if args.workload == "catdqn":
        return catdqn_main(args)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args.gpus))
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if args.scaling == "strong":
        # the SAME job on 1, 2, 4, 8 GPUs (multigpu_rl_base.py:62-63 counts sample_size * n_runners per
        # iteration): --total-envs environments and ONE global minibatch of 512 x 8 rows per update, both split
        # evenly over the ranks -- every N takes the same number of optimiser steps on the same amount of data
        if args.workload != "ppo256" or args.total_envs % (32 * world) or (512 * 8) % world:
            raise SystemExit("--scaling strong: ppo256 workload, --total-envs a multiple of 32 x the rank count")
        N_ENVS = args.total_envs // world
        MINIBATCH = 512 * 8 // world
        args.no_cpu_baseline = args.no_roofline = True
    cpu_pool = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not args.roofline_only:
        cpu_pool = start_cpu_pool()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product has no CPU path)")
    # A rank that stops making progress (a rendezvous that never completes, a captured collective waiting for a peer
    # that is not there) would otherwise sit until the launcher's own timeout without a word: a watchdog thread names the
    # phase it stalled in and ends the rank with exit code 3 -- which the supervising parent (spawn_ranks) answers with
    # ONE retry on eager collectives.  Every phase that can legitimately take long ticks: start-up, each priming /
    # warm-up / timed step, each repetition of the phase split and of the diagnostics.
    progress = {"itr": 0, "phase": "start-up", "t": time.time(), "graph_collectives": None}

    def tick(phase, i=None):
        progress.update(phase=phase, t=time.time())
        if i is not None:
            progress["itr"] = i
    if world > 1:
        import threading

        def watchdog(limit=float(os.environ.get("ARL_BENCH_STALL_S", "240"))):
            while True:
                time.sleep(min(5.0, limit / 4))
                if time.time() - progress["t"] > limit:
                    # Leave a note for the NEXT run on this node: the driver launches N = 2, 4, 8 back to back under its
                    # own torch.distributed.run (no supervising parent to retry for it) -- a stall while the collectives
                    # are captured costs this run its line, but the following runs take the eager path by themselves.
                    if progress["graph_collectives"] and os.environ.get("ARL_SYNC_GRAPH") != "0":
                        try:
                            with open(STALL_MARKER, "w") as f:
                                f.write("%d %d %s\n" % (int(time.time()), world, progress["phase"]))
                        except OSError:
                            pass
                    sys.stderr.write("bench.py rank %d: no progress for %.0f s in %s of step %d (graph_collectives=%s, backend=%s); "
                                     "ARL_SYNC_GRAPH=0 runs the collectives eagerly\n" %
                                     (rank, limit, progress["phase"], progress["itr"], progress["graph_collectives"],
                                      os.environ.get("ARL_BENCH_BACKEND", "nccl")))
                    sys.stderr.flush()
                    os._exit(3)
        threading.Thread(target=watchdog, daemon=True).start()
    # ARL_BENCH_ONE_GPU=1 + ARL_BENCH_BACKEND=gloo: development check of the N>1 control flow with every
    # rank on GPU 0 (RCCL refuses two ranks on one device); never a measurement.
    device = torch.device("cuda", 0 if os.environ.get("ARL_BENCH_ONE_GPU") == "1" else local)
    torch.cuda.set_device(device)
    backend = None
    if world > 1 and "ARL_SYNC_GRAPH" not in os.environ and stalled_before():
        os.environ["ARL_SYNC_GRAPH"] = "0"
        os.environ.setdefault("ARL_BENCH_GRAPH_FALLBACK", "eager after a capture stall in an earlier run on this node")
        if rank == 0:
            sys.stderr.write("bench.py: an earlier %d-rank run of this build stalled while its collectives were captured "
                             "(%s): this run issues them eagerly (ARL_SYNC_GRAPH=0; the line says so in graph_fallback). "
                             "Delete the file or set ARL_SYNC_GRAPH=1 to capture again.\n" % (world, STALL_MARKER))
            sys.stderr.flush()
    if world > 1 or os.environ.get("ARL_FORCE_SYNC") == "1":
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("NCCL_DEBUG", "WARN")        # no INFO/VERSION chatter on stdout
        backend = os.environ.get("ARL_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device, rank=rank, world_size=world)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    tick("build")
    import __graft_entry__
    __graft_entry__.build()
    tick("start-up")
    if args.roofline_only:
        leg = roofline_nstep if args.workload == "a2c1024" else roofline_gae
        print(json.dumps({"roofline": leg(device, args.roofline_log2, 30)}), flush=True)
        return

    game = GAMES_8[rank % 8] if args.suite else GAME
    runner, sampler, algo, policy = build_workload(device, 0, rank, world, game, not args.no_graph,
                                                   kind="a2c" if args.workload == "a2c1024" else "ppo",
                                                   pad_actions_to=18 if args.suite else None)   # one head for the 8 games
    progress["graph_collectives"] = getattr(algo.optimizer, "graph_collectives", None)
    tick("priming")
    # test hook (tests/test_bench_launch.py): a capture that hangs instead of throwing -- the last rank never reaches its
    # collectives while the others wait in theirs.  Only while the collectives are to be captured: the eager retry runs.
    inject_hang = (os.environ.get("ARL_BENCH_INJECT") == "capture_hang" and world > 1 and rank == world - 1
                   and os.environ.get("ARL_SYNC_GRAPH") != "0")

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def one_step(i, sampler, algo):             # (shadows the module-level helper: the same two calls + the heartbeat)
        tick("rollout", i)
        samples, _ = sampler.obtain_samples(i)
        tick("learner", i)
        if inject_hang and i == 2:
            tick("learner (injected capture hang)", i)
            time.sleep(1e6)
        algo.optimize_policy(i, samples)

    itr = 0
    # Priming (untimed, whatever --warmup says): the sampler captures its hipGraph in its first batch and the learner in
    # its third call (two eager calls first) -- a caller that asks for fewer than three warm-up steps would otherwise
    # time the captures.  Reported as `priming_steps`.
    PRIMING = 3 if not args.no_graph else 0
    for _ in range(PRIMING):
        one_step(itr, sampler, algo)
        itr += 1
    for _ in range(args.warmup):
        one_step(itr, sampler, algo)
        itr += 1
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step(itr, sampler, algo)
        itr += 1
    barrier()
    elapsed = elapsed_local = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    steps_per_gpu = args.steps * N_ENVS * HORIZON
    line = {
        "metric": "env-steps/sec (whole node), 256-env PPO Atari" if args.workload == "ppo256"
                  else "env-steps/sec (whole node), 1024-env A2C Atari (BASELINE config 3, not the headline metric)",
        "value": round(world * steps_per_gpu / elapsed, 1),
        "unit": "env-steps/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "priming_steps": PRIMING,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "rccl_ranks": dist.get_world_size() if dist.is_initialized() else 1,
        "backend": ("rccl (torch.distributed nccl)" if backend == "nccl" else backend) if backend else "none (single GPU)",
        "config": {"workload": ("A2C %s, %d envs/GPU, horizon %d (5-step returns), spec-%d CNN (fp32), one "
                                "rmsprop step per batch; synthetic fixed-frame emulator; %s" if args.workload == "a2c1024"
                                else "PPO %s, %d envs/GPU, horizon %d, spec-%d CNN (fp32), minibatch %d/GPU x 4 epochs, "
                                "adam; synthetic fixed-frame emulator; %s") %
                               (("8-game suite" if args.suite else GAME, N_ENVS, HORIZON, CNN_SPEC) +
                                ((MINIBATCH,) if args.workload == "ppo256" else ()) +
                                ("hipGraph rollout" if not args.no_graph else "eager",)),
                   "total_envs": N_ENVS * world, "global_minibatch": MINIBATCH * world,
                   "env_steps_per_step_per_gpu": N_ENVS * HORIZON,
                   "parallelism": ("dp%d sync all-reduce (%s)" % (world, "RCCL" if backend == "nccl" else backend))
                                  if world > 1 else "single"},
    }
    # phase split of the same workload (after the timed region): rollout only / learner only
    def timed(fn, reps):
        barrier()
        t = time.perf_counter()
        for i in range(reps):
            tick("phase split / diagnostics")
            fn(i)
        barrier()
        return (time.perf_counter() - t) / reps
    reps = max(10, min(50, args.steps))
    t_roll = timed(lambda i: sampler.obtain_samples(itr + i), reps)
    samples = sampler.samples_buf
    t_learn = timed(lambda i: algo.optimize_policy(itr + i, samples), reps)
    line["phases"] = {"rollout_ms": round(t_roll * 1e3, 4), "learner_ms": round(t_learn * 1e3, 4),
                      "rollout_only_env_steps_per_s": round(N_ENVS * HORIZON / t_roll, 1)}
    tick("diagnostics", itr)
    if rank == 0 and world == 1 and args.workload == "ppo256" and not args.no_alt_routes and not args.no_graph:
        # (here, under the headline's own conditions -- the CPU baseline's worker pools still parked on their barriers --
        #  and not behind cpu_baseline(): the teardown of its 200-odd processes was seen to cost the leg that followed it
        #  15 % on one box, profiles/r06/bench_alt_after_cpu_baseline.json)
        tick("alt routes")
        line["alt_routes"] = alt_routes(device, args.steps, args.warmup, PRIMING)
        line["accuracy"] = route_accuracy(device)
    if world > 1:
        line["multi_gpu"] = multi_gpu_diagnostics(device, world, rank, algo, policy, sampler, samples, timed, reps,
                                                  itr + 2 * reps, elapsed_local / args.steps, t_roll, t_learn, tick)
        if os.environ.get("ARL_BENCH_GRAPH_FALLBACK"):
            line["graph_fallback"] = os.environ["ARL_BENCH_GRAPH_FALLBACK"]
        if rank == 0 and line["multi_gpu"].get("graph_captured"):
            clear_stall_marker()
    if rank == 0 and world == 1:
        if not args.no_roofline and args.workload == "a2c1024":
            # config 3 is sampler-bound (BASELINE.json): its HBM-bound kernels are the 5-step return scan and the env step
            line["roofline"] = roofline_nstep(device, args.roofline_log2, 30)
            line["kernels"] = [env_step_sweep(device)]
        elif not args.no_roofline:
            line["roofline"] = roofline_gae(device, args.roofline_log2, 30)
            line["kernels"] = kernel_table(device, sampler, algo, policy)
            line["kernels"].append(env_step_sweep(device))
            line["mfma"] = mfma_table(device, policy)
        if not args.no_cpu_baseline:
            line["cpu_baseline"] = cb = cpu_baseline(device, policy, cpu_pool, sampler, algo, itr + 2 * reps,
                                                     gae_lambda=1.0 if args.workload == "a2c1024" else 0.95)
            line["gpu_over_cpu"] = {
                "rollout_only": round(line["phases"]["rollout_only_env_steps_per_s"] / cb["value"], 2),
                "whole_loop": round(line["value"] / cb["whole_loop"], 2),
                "value_over_cpu_sampler": round(line["value"] / cb["value"], 2),
                "rollout_only_vs_undisturbed_cpu": round(line["phases"]["rollout_only_env_steps_per_s"] / cb["undisturbed"], 2),
                "note": "rollout_only: GPU rollout (sampler only) vs the CPU sampler port (best pool shape, median window); "
                        "value_over_cpu_sampler: `value` (rollout + PPO learner) vs that same sampler-only CPU figure; whole_loop: "
                        "`value` (rollout + PPO learner on the device) vs the CPU sampler port feeding the same device "
                        "learner serially, as the reference's runner does; rollout_only_vs_undisturbed_cpu: against "
                        "the CPU sampler's 10th-percentile batch time (what it does when the shared host is quiet)"}
    runner.shutdown()
    if dist.is_initialized():
        dist.barrier()
    if rank == 0:
        # RCCL writes its version banner to C stdout (block-buffered when piped): flush it first so that
        # the JSON line is the LAST line this process prints
        import ctypes
        ctypes.CDLL(None).fflush(None)
        print(json.dumps(line), flush=True)
    if dist.is_initialized():
        # The line is out; the teardown of the process group (communicators that captured graphs still refer to, a
        # watchdog thread) gets a deadline so that it can neither hold the line back nor hang the launcher
        import threading
        t = threading.Timer(30.0, lambda: os._exit(0))
        t.daemon = True
        t.start()
        dist.destroy_process_group()
        t.cancel()
    # Leave nothing for interpreter finalisation to destroy against a HIP runtime that is shutting itself down: the
    # captured graphs go now, while the device is alive (a normal exit follows -- profilers flush their output there)
    algo._graph = algo._graph_out = None
    sampler._graph = None
    import gc
    gc.collect()
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
