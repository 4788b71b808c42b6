"""The synchronous multi-rank path end to end on the device: two real ranks (separate processes, both on
GPU 0, `gloo` because RCCL refuses two ranks on one device) run AccelRLSync + mPPO with the real policy,
sampler and kernels.  Checks the invariants of accel_rl/runners/multigpu_rl_base.py and
optimizers/sync/*: broadcast initial parameters, per-rank seeds (different rollouts), ONE all-reduced
gradient per minibatch, identical parameters on every rank after every update."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from mp_util import leave_group, run_ranks

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _rank(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0")
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from accel_rl_amd.algos.pg.ppo import mPPO
    from accel_rl_amd.envs.synthetic_atari import SynthAtariEnv
    from accel_rl_amd.policies.atari_cnn_policy import AtariCnnPolicy
    from accel_rl_amd.policies.atari_cnn_specs import cnn_specs
    from accel_rl_amd.runners.sync import AccelRLSync
    from accel_rl_amd.sampler.gpu_sampler import GpuVecSampler
    from accel_rl_amd.util import logger
    logger.set_quiet(True)
    sampler = GpuVecSampler(EnvCls=SynthAtariEnv, env_args=dict(game="breakout"), horizon=5, n_parallel=4, envs_per=4,
                            max_path_length=40, max_decorrelation_steps=20, device="cuda:0")
    algo = mPPO(optimizer_args=dict(minibatch_size=64, epochs=2))
    policy = AtariCnnPolicy(**cnn_specs[0])
    runner = AccelRLSync(algo=algo, policy=policy, sampler=sampler, n_steps=160 * 2 * 4, seed=5,
                         affinities=dict(gpu=0), log_interval_steps=320, backend="gloo")
    assert runner.n_runners == world and runner.rank == rank
    n_itr = runner.startup()
    init = policy.get_param_values().copy()
    obs0 = None
    for itr in range(n_itr):
        samples, _ = sampler.obtain_samples(itr)
        if itr == 0:
            obs0 = samples["observations"][:64].cpu().numpy().copy()
        algo.optimize_policy(itr, samples)
    torch.cuda.synchronize()
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), init=init, final=policy.get_param_values(), obs0=obs0,
             n_itr=n_itr, tag=np.array(algo.optimizer.parallelism_tag))
    runner.shutdown()
    leave_group(dist)


def test_two_ranks_stay_bit_identical(tmp_path):
    ctx = mp.get_context("spawn")
    port = _free_port()
    codes, _ = run_ranks(ctx, _rank, [(r, 2, port, str(tmp_path)) for r in range(2)], 300)
    assert codes == [0, 0], codes
    a, b = (np.load(os.path.join(str(tmp_path), "rank%d.npz" % r)) for r in range(2))
    assert str(a["tag"]) == "synchronous" and int(a["n_itr"]) == int(b["n_itr"]) == 5      # 1280 / (160 x 2 ranks) + 1
    np.testing.assert_array_equal(a["init"], b["init"])                   # rank 0's parameters were broadcast
    assert not np.array_equal(a["obs0"], b["obs0"])                       # seed + 100 * rank: different rollouts
    np.testing.assert_array_equal(a["final"], b["final"])                 # same averaged gradient, same update, every step
    assert np.isfinite(a["final"]).all() and not np.array_equal(a["final"], a["init"])


def _rccl_world1(port, out_dir, capture=True):
    """One rank on the `nccl` backend (= RCCL on ROCm) with the collective forced: mPPO's eager minibatches with
    the asynchronous tail / head all-reduce of the gradient bucket on RCCL's stream (optimizers/sync.py
    _share_grad_async, optimizers/single.py _overlapped_minibatches), then PPO's single-graph learner from the
    same seed.  A sum over one rank x 1/1 is the identity, so both must produce the same bits."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
                      NCCL_DEBUG="WARN")
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0), rank=0, world_size=1)
    assert dist.get_backend() == "nccl"
    from accel_rl_amd.algos.pg.ppo import PPO, mPPO
    from accel_rl_amd.envs.synthetic_atari import SynthAtariEnv
    from accel_rl_amd.policies.atari_cnn_policy import AtariCnnPolicy
    from accel_rl_amd.policies.atari_cnn_specs import cnn_specs
    from accel_rl_amd.runners.accel_rl import AccelRL
    from accel_rl_amd.runners.sync import AccelRLSync
    from accel_rl_amd.sampler.gpu_sampler import GpuVecSampler
    from accel_rl_amd.util import logger
    logger.set_quiet(True)
    results = dict()
    for tag, Algo, Runner in (("sync", mPPO, AccelRLSync), ("single", PPO, AccelRL)):
        sampler = GpuVecSampler(EnvCls=SynthAtariEnv, env_args=dict(game="breakout"), horizon=5, n_parallel=4,
                                envs_per=8, max_path_length=40, max_decorrelation_steps=20, device="cuda:0")
        algo = Algo(optimizer_args=dict(minibatch_size=128, epochs=2), lr_schedule="linear")
        policy = AtariCnnPolicy(**cnn_specs[1])
        kw = dict(backend="nccl") if Runner is AccelRLSync else dict()
        runner = Runner(algo=algo, policy=policy, sampler=sampler, n_steps=320 * 5, seed=11,
                        affinities=dict(gpu=0), log_interval_steps=320, **kw)
        if tag == "sync":
            algo.optimizer._force_collective = True
            algo.optimizer.graph_collectives = bool(capture)
            if capture == "fails":                     # a rank whose collective cannot be captured
                issue = algo.optimizer._share_grad_async

                def refusing(*a, **k):
                    if torch.cuda.is_current_stream_capturing():
                        raise RuntimeError("injected: no collectives under capture")
                    return issue(*a, **k)
                algo.optimizer._share_grad_async = refusing
        n_itr = runner.startup()
        assert algo.optimizer.parallelism_tag == ("synchronous" if tag == "sync" else "single")
        norms = []
        for itr in range(n_itr):                       # 6 iterations: PPO replays its hipGraph from the third on
            samples, _ = sampler.obtain_samples(itr)
            _, info = algo.optimize_policy(itr, samples)
            norms.append(info["GradNorm"].cpu().numpy().copy())
        torch.cuda.synchronize()
        results[tag + "_final"] = policy.get_param_values()
        results[tag + "_norms"] = np.stack(norms)
        results[tag + "_split"] = np.int64(policy.grad_split_offset)
        results[tag + "_captured"] = np.int64(algo._graph is not None)
        runner.shutdown()
    np.savez(os.path.join(out_dir, "rccl.npz"), **results)
    leave_group(dist)


@pytest.mark.parametrize("capture", [True, False, "fails"], ids=["one hipGraph", "eager", "capture fails"])
def test_rccl_backend_forced_collective_matches_single_gpu(tmp_path, capture):
    """the path that ships for N > 1 (backend `nccl` = RCCL, asynchronous all-reduce of
    the two bucket slices, optimiser step waiting on RCCL's stream) exercised on the device -- as eager minibatches, and
    with the WHOLE optimize_policy call, RCCL's all-reduces included, captured in one hipGraph (the default on `nccl`);
    "capture fails": a collective that refuses to be captured must leave the learner on the eager road with the same bits;
    reference: accel_rl/optimizers/sync/base.py:22-24, sync_ppo_optimizer.py:27-34,56-78."""
    ctx = mp.get_context("spawn")
    codes, _ = run_ranks(ctx, _rccl_world1, [(_free_port(), str(tmp_path), capture)], 600)
    assert codes == [0], codes
    r = np.load(os.path.join(str(tmp_path), "rccl.npz"))
    assert 0 < int(r["sync_split"]) < r["sync_final"].size            # both bucket slices are non-empty
    assert int(r["sync_captured"]) == int(capture is True) and int(r["single_captured"]) == 1
    assert r["sync_norms"].shape == (6, 4) and np.isfinite(r["sync_norms"]).all()
    np.testing.assert_array_equal(r["sync_norms"], r["single_norms"])
    np.testing.assert_array_equal(r["sync_final"], r["single_final"])


# ------------------------------------------------------------------ the N > 1 update IS the mean-gradient update

def _rank_mean_grad(rank, world, port, out_dir, kind):
    """One rank of two (gloo, both on GPU 0) through the path that ships for N > 1 -- mPPO: `_overlapped_minibatches`
    (tail slice reduced from the policy's split hook, head slice after the backward pass); mA2C: the blocking
    `_share_grad` + norm clip.  Records, per optimiser step: this rank's LOCAL gradient as handed to the all-reduce
    (optimizers/sync.py `_grad_tap`), the optimiser state right before the update and the parameters after it."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0")
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from accel_rl_amd.algos.pg.a2c import mA2C
    from accel_rl_amd.algos.pg.ppo import mPPO
    from accel_rl_amd.envs.synthetic_atari import SynthAtariEnv
    from accel_rl_amd.policies.atari_cnn_policy import AtariCnnPolicy
    from accel_rl_amd.policies.atari_cnn_specs import cnn_specs
    from accel_rl_amd.runners.sync import AccelRLSync
    from accel_rl_amd.sampler.gpu_sampler import GpuVecSampler
    from accel_rl_amd.util import logger
    logger.set_quiet(True)
    sampler = GpuVecSampler(EnvCls=SynthAtariEnv, env_args=dict(game="breakout"), horizon=5, n_parallel=4, envs_per=4,
                            max_path_length=40, max_decorrelation_steps=20, device="cuda:0")
    if kind == "ppo":
        algo = mPPO(optimizer_args=dict(minibatch_size=32, epochs=2))
    else:
        algo = mA2C(optimizer_args=dict(grad_norm_clip=0.02))    # rmsprop + norm clip (a2c.py:15-27), low enough to engage
    policy = AtariCnnPolicy(**cnn_specs[0])
    runner = AccelRLSync(algo=algo, policy=policy, sampler=sampler, n_steps=80 * 2 * 3, seed=9,
                         affinities=dict(gpu=0), log_interval_steps=160, backend="gloo")
    n_itr = runner.startup()
    opt = algo.optimizer
    assert opt.parallelism_tag == "synchronous" and opt._n_gpu == world
    opt._grad_tap = taps = []
    steps = []
    apply_update = opt._apply_update
    host = lambda x: None if x is None else x.detach().cpu().numpy().copy()            # noqa: E731

    def hooked(avg_factor=1.0):
        # the local gradient of this step: the slices tapped since the last update, put back together
        local = np.full(policy.flat_grads.numel(), np.nan, np.float32)
        for first, g in taps:
            assert np.isnan(local[first:first + g.numel()]).all(), "a slice of the bucket was all-reduced twice"
            local[first:first + g.numel()] = host(g)
        assert not np.isnan(local).any(), "a slice of the bucket was never all-reduced"
        del taps[:]
        rec = dict(local=local, reduced=host(policy.flat_grads), p=host(policy.flat_params), s0=host(opt._slot0),
                   s1=host(opt._slot1), t=np.float32(opt._step_count.item()), avg=np.float32(avg_factor))
        apply_update(avg_factor)
        torch.cuda.synchronize()
        rec["p_after"] = host(policy.flat_params)
        steps.append(rec)
    opt._apply_update = hooked
    norms = []
    for itr in range(2):
        samples, _ = sampler.obtain_samples(itr)
        _, info = algo.optimize_policy(itr, samples)
        norms.append(host(info["GradNorm"]).reshape(-1))
    out = dict(n=np.int64(len(steps)), norms=np.concatenate(norms), lr=np.float32(opt._learning_rate),
               args=np.array(opt._kernel_args, np.float64), clip=np.float32(opt._grad_norm_clip or 0), split=np.int64(getattr(policy, "grad_split_offset", 0)))
    for k, rec in enumerate(steps):
        for key, val in rec.items():
            if val is not None:
                out["%s_%d" % (key, k)] = val
    np.savez(os.path.join(out_dir, "mean_%s_rank%d.npz" % (kind, rank)), **out)
    runner.shutdown()
    leave_group(dist)


@pytest.mark.parametrize("kind", ["ppo", "a2c"])
def test_two_rank_update_is_the_oracle_step_on_the_mean_gradient(tmp_path, kind):
    """VERDICT r2 item 3a.  Two ranks ending identical would not notice a slice summed twice or 1/n applied on the
    wrong side of the clip.  Here every optimiser step of both ranks is redone by the oracle (Lasagne adam / rmsprop +
    total_norm_constraint, oracle/ref_port.py) from the recorded pre-step state on (g_rank0 + g_rank1) / 2 -- the order
    of accel_rl/optimizers/sync/sync_ppo_optimizer.py:27-34 and optimizers/util.py:63-67: sum, x 1/n, norm (clip),
    update."""
    from oracle import ref_port as P
    ctx = mp.get_context("spawn")
    port = _free_port()
    codes, _ = run_ranks(ctx, _rank_mean_grad, [(r, 2, port, str(tmp_path), kind) for r in range(2)], 300)
    assert codes == [0, 0], codes
    a, b = (np.load(os.path.join(str(tmp_path), "mean_%s_rank%d.npz" % (kind, r))) for r in range(2))
    n = int(a["n"])
    assert n == int(b["n"]) == (20 if kind == "ppo" else 2)         # 2 iterations x (2 epochs x 5 minibatches of 160 samples | 1 step)
    if kind == "ppo":
        assert 0 < int(a["split"]) < a["p_0"].size                  # both slices of the bucket exist
    lr, args, clip = np.float32(a["lr"]), a["args"], float(a["clip"])
    clipped = 0
    for k in range(n):
        g0, g1 = a["local_%d" % k], b["local_%d" % k]
        assert not np.array_equal(g0, g1)                           # different rollouts, different gradients
        gsum = (g0 + g1).astype(np.float32)
        np.testing.assert_array_equal(a["reduced_%d" % k], gsum)    # each element summed exactly once (fp32 a + b)
        np.testing.assert_array_equal(b["reduced_%d" % k], gsum)
        assert a["avg_%d" % k] == b["avg_%d" % k] == np.float32(0.5)
        gavg = (gsum * np.float32(0.5)).astype(np.float32)
        np.testing.assert_array_equal(a["p_%d" % k], b["p_%d" % k])
        if kind == "ppo":
            _, norm = P.clip_by_total_norm(gavg, None)
            want, _, _, t = P.adam_step(a["p_%d" % k], gavg, a["s0_%d" % k], a["s1_%d" % k], a["t_%d" % k], lr,
                                        beta1=args[0], beta2=args[1], eps=args[2])
        else:
            gc, norm = P.clip_by_total_norm(gavg, clip)
            clipped += norm > clip
            want, _ = P.rmsprop_step(a["p_%d" % k], gc, a["s0_%d" % k], lr, rho=args[0], eps=args[2])
            # the wrong orders would land elsewhere: clip each rank's gradient and then average, or clip the SUM
            c0, _ = P.clip_by_total_norm(g0, clip)
            c1, _ = P.clip_by_total_norm(g1, clip)
            wrong, _ = P.rmsprop_step(a["p_%d" % k], ((c0 + c1) * np.float32(0.5)).astype(np.float32), a["s0_%d" % k], lr,
                                      rho=args[0], eps=args[2])
            if norm > clip:
                assert not np.allclose(a["p_after_%d" % k], wrong, rtol=1e-5, atol=1e-6)
        for r in (a, b):
            got = r["p_after_%d" % k]
            assert np.allclose(got, want, rtol=1e-5, atol=1e-6), (kind, k, np.abs(got - want).max())
            assert abs(r["norms"][k] - norm) <= 1e-5 * max(1.0, norm), (k, r["norms"][k], norm)    # norm of the AVERAGE
        np.testing.assert_array_equal(a["p_after_%d" % k], b["p_after_%d" % k])
    if kind == "a2c":
        assert clipped >= 1, "the clip never engaged: the test would not see it applied to the wrong vector"
