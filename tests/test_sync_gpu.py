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
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_stay_bit_identical(tmp_path):
    ctx = mp.get_context("spawn")
    port = _free_port()
    procs = [ctx.Process(target=_rank, args=(r, 2, port, str(tmp_path))) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    a, b = (np.load(os.path.join(str(tmp_path), "rank%d.npz" % r)) for r in range(2))
    assert str(a["tag"]) == "synchronous" and int(a["n_itr"]) == int(b["n_itr"]) == 5      # 1280 / (160 x 2 ranks) + 1
    np.testing.assert_array_equal(a["init"], b["init"])                   # rank 0's parameters were broadcast
    assert not np.array_equal(a["obs0"], b["obs0"])                       # seed + 100 * rank: different rollouts
    np.testing.assert_array_equal(a["final"], b["final"])                 # same averaged gradient, same update, every step
    assert np.isfinite(a["final"]).all() and not np.array_equal(a["final"], a["init"])


def _rccl_world1(port, out_dir):
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
        runner.shutdown()
    np.savez(os.path.join(out_dir, "rccl.npz"), **results)
    dist.barrier()
    dist.destroy_process_group()


def test_rccl_backend_forced_collective_matches_single_gpu(tmp_path):
    """the path that ships for N > 1 (backend `nccl` = RCCL, asynchronous all-reduce of the
    two bucket slices, optimiser step waiting on RCCL's stream) exercised on the device; reference:
    accel_rl/optimizers/sync/base.py:22-24, sync_ppo_optimizer.py:27-34."""
    ctx = mp.get_context("spawn")
    p = ctx.Process(target=_rccl_world1, args=(_free_port(), str(tmp_path)))
    p.start()
    p.join(600)
    assert p.exitcode == 0, p.exitcode
    r = np.load(os.path.join(str(tmp_path), "rccl.npz"))
    assert 0 < int(r["sync_split"]) < r["sync_final"].size            # both bucket slices are non-empty
    assert r["sync_norms"].shape == (6, 4) and np.isfinite(r["sync_norms"]).all()
    np.testing.assert_array_equal(r["sync_norms"], r["single_norms"])
    np.testing.assert_array_equal(r["sync_final"], r["single_final"])
