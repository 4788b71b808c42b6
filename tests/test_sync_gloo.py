"""World-size-2 `gloo` tests (CPU) of the synchronous multi-GPU protocol:
flat-bucket all-reduce + 1/n averaging in the optimizers, and AccelRLSync's
rank seeding, parameter broadcast, lock-step iteration count and trajectory
gather.  The device kernels are replaced by trivial stand-ins HERE (test doubles
inside this file only); the protocol code under test is the product's."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from mp_util import leave_group, run_ranks


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


class _Target(object):
    """Stand-in for the policy's flat buckets (CPU tensors)."""
    def __init__(self, n):
        self.device = torch.device("cpu")
        self.flat_params = torch.zeros(n)
        self.flat_grads = torch.zeros(n)


def _optimizer_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from accel_rl_amd.optimizers import update_methods
    from accel_rl_amd.optimizers.sync import SyncPpoOptimizer, SyncA2cOptimizer
    for cls, kw in ((SyncPpoOptimizer, dict(update_method_args=None, epochs=1, minibatch_size=4)),
                    (SyncA2cOptimizer, dict())):
        opt = cls(learning_rate=1e-3, update_method=update_methods.adam, **kw)
        assert opt.parallelism_tag == "synchronous"
        opt._target = _Target(10)
        opt.init_comm(None, rank, world)
        opt._target.flat_grads.copy_(torch.arange(10.) * (rank + 1))
        opt._share_grad()                                   # all-reduce SUM of the ONE flat vector
        want = torch.arange(10.) * sum(r + 1 for r in range(world))
        assert torch.equal(opt._target.flat_grads, want)
        assert opt._avg_factor() == 1.0 / world             # applied AFTER the sum (sync/base.py:14-16)
        # the consensus the learner takes before it commits to a captured graph: one rank's failure is everybody's
        assert opt.ranks_agree(True) is True
        assert opt.ranks_agree(rank != 1) is False
    out.put((rank, "ok"))
    leave_group(dist)


def test_sync_optimizer_allreduce_world2():
    ctx = mp.get_context("spawn")
    out, port = ctx.Queue(), _free_port()
    codes, got = run_ranks(ctx, _optimizer_worker, [(r, 2, port, out) for r in range(2)], 120,
                           before_join=lambda: sorted(out.get(timeout=110) for _ in range(2)))
    assert codes == [0, 0] and got == [(0, "ok"), (1, "ok")], (codes, got)


def _overlap_worker(rank, world, port, out):
    """The path mPPO ships for N > 1 (optimizers/single.py `_overlapped_minibatches`): the bucket's tail is all-reduced
    from the policy's split hook while the backward pass still writes the head, the head after it; the update must see
    every element summed over the ranks exactly once.  Backward pass and update kernel are stand-ins (CPU tensors)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from accel_rl_amd.optimizers import update_methods
    from accel_rl_amd.optimizers.sync import SyncPpoOptimizer
    opt = SyncPpoOptimizer(learning_rate=1e-3, update_method=update_methods.adam, update_method_args=None, epochs=1,
                           minibatch_size=4)
    n, split = 10, 4
    tgt = opt._target = _Target(n)
    tgt.grad_split_offset = split
    opt.init_comm(None, rank, world)
    assert opt._overlap_allreduce and not opt.graph_ready()      # gloo: eager, never captured
    opt._n_minibatches, opt._idx_dev, opt._losses = 3, [None] * 3, None
    opt._grad_tap = taps = []
    seen = []

    def local(k, r):
        return torch.arange(float(n)) * (r + 1) + 100. * k

    def backward(losses, mb):
        k = len(seen)
        tgt.flat_grads[split:] = local(k, rank)[split:]         # dense + heads first ...
        mb["split_hook"]()                                      # ... their all-reduce starts here
        tgt.flat_grads[:split] = local(k, rank)[:split]         # ... while the conv layers' gradients are written
        return torch.zeros(4)
    opt._backward = backward
    opt._apply_update = lambda avg: seen.append((tgt.flat_grads.clone(), avg))
    opt._recent_grad_norms = lambda count: torch.zeros(count)
    opt._overlapped_minibatches(dict())
    assert len(seen) == 3 and len(taps) == 6                    # two slices per minibatch, each reduced once
    for k, (g, avg) in enumerate(seen):
        want = sum(local(k, r) for r in range(world))
        assert torch.equal(g, want), (k, g, want)
        assert avg == 1.0 / world
    firsts = sorted(set(f for f, _ in taps))
    assert firsts == [0, split]
    out.put((rank, "ok"))
    leave_group(dist)


def test_overlapped_tail_head_allreduce_world2():
    ctx = mp.get_context("spawn")
    out, port = ctx.Queue(), _free_port()
    codes, got = run_ranks(ctx, _overlap_worker, [(r, 2, port, out) for r in range(2)], 120,
                           before_join=lambda: sorted(out.get(timeout=110) for _ in range(2)))
    assert codes == [0, 0] and got == [(0, "ok"), (1, "ok")], (codes, got)


# ----------------------------------------------------------------------------- runner

class _FakePolicy(object):
    recurrent = False

    def __init__(self):
        self.device = torch.device("cpu")

    def initialize(self, env_spec, device=None):
        g = torch.Generator().manual_seed(int(np.random.randint(1 << 30)))   # differs per rank seed
        self.flat_params = torch.randn(16, generator=g)
        self.flat_grads = torch.zeros(16)
        self.n_params = 16

    def get_param_values(self):
        return self.flat_params.numpy().copy()

    def reset(self, n_batch=None):
        pass


class _FakeSampler(object):
    device = "cpu"

    def initialize(self, seed, affinities, discount, need_extra_obs):
        self.seed = seed
        return None, 20, 5, True

    def policy_init(self, policy):
        pass

    def obtain_samples(self, itr):
        from accel_rl_amd.sampler.util import TrajInfo
        import torch.distributed as dist
        rank = dist.get_rank() if dist.is_initialized() else int(os.environ.get("RANK", 0))
        infos = [TrajInfo(Length=10 + rank, Return=float(itr))] if itr % 2 == rank else []
        return dict(), infos

    def shutdown(self):
        pass


class _FakeAlgo(object):
    need_extra_obs = True
    discount = 0.99
    opt_info_keys = ["GradNorm"]

    def __init__(self):
        from accel_rl_amd.optimizers import update_methods
        from accel_rl_amd.optimizers.sync import SyncA2cOptimizer
        self.optimizer = SyncA2cOptimizer(learning_rate=1e-3, update_method=update_methods.rmsprop)
        self.calls = 0

    def initialize(self, policy, env_spec, sample_size, horizon, mid_batch_reset):
        self.policy = policy
        self.optimizer._target = policy

    def set_n_itr(self, n):
        self.n_itr = n

    def optimize_policy(self, itr, samples):
        self.calls += 1
        self.policy.flat_grads.fill_(float(__import__("torch").distributed.get_rank() + 1))
        self.optimizer._share_grad()
        self.policy.flat_params.sub_(self.policy.flat_grads * self.optimizer._avg_factor() * 0.1)
        return None, dict(GradNorm=torch.tensor([1.0]))


def _runner_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from accel_rl_amd.runners.sync import AccelRLSync
    from accel_rl_amd.util import logger
    logger.set_quiet(True)
    algo, policy, sampler = _FakeAlgo(), _FakePolicy(), _FakeSampler()
    runner = AccelRLSync(algo=algo, policy=policy, sampler=sampler, n_steps=200, seed=7,
                         log_interval_steps=80, backend="gloo")
    runner.init_policy = lambda env_spec: policy.initialize(env_spec)
    runner.save_itr_snapshot = lambda itr: None
    runner._log_entropy = False
    orig_init_logging = runner.init_logging

    def init_logging():
        orig_init_logging()
        runner._log_entropy = False
    runner.init_logging = init_logging
    runner.train()
    out.put(dict(rank=rank, seed=runner.seed, sampler_seed=sampler.seed, n_itr=runner._n_itr,
                 calls=algo.calls, params=policy.flat_params.numpy().copy(),
                 trajs=runner._cum_completed_trajs,
                 tab=getattr(runner, "last_tabular", None) and dict(runner.last_tabular)))
    leave_group(dist)


def test_sync_runner_world2():
    ctx = mp.get_context("spawn")
    out, port = ctx.Queue(), _free_port()
    codes, res = run_ranks(ctx, _runner_worker, [(r, 2, port, out) for r in range(2)], 240,
                           before_join=lambda: sorted((out.get(timeout=180) for _ in range(2)), key=lambda d: d["rank"]))
    assert codes == [0, 0], codes
    r0, r1 = res
    assert (r0["seed"], r1["seed"]) == (7, 107)                  # seed + 100*rank (multigpu_rl_base.py:28)
    assert (r0["sampler_seed"], r1["sampler_seed"]) == (8, 108)  # sampler gets seed + 1
    # n_itr counts sample_size * n_runners per iteration (:62-63): 200 // 40 = 5 -> log every 2 -> 4 (+1)
    assert r0["n_itr"] == r1["n_itr"] == 5 and r0["calls"] == r1["calls"] == 5
    # params were broadcast from rank 0 and stay identical under identical averaged updates
    np.testing.assert_array_equal(r0["params"], r1["params"])
    # rank 0 saw both ranks' completed trajectories at log time; rank 1 logs nothing
    assert r0["trajs"] == 4 and r1["trajs"] == 0      # itrs 0..3 contribute one episode each (gather at itr 1, 3)
    assert r0["tab"] is not None and r0["tab"]["CumTotalSteps"] == 4 * 40 and r1["tab"] is None


def _mismatch_worker(rank, world, port, out):
    """Runners whose policies differ in size (e.g. one game per rank with each game's own action set) cannot share one
    flat gradient bucket: AccelRLSync.init_comm must say so on every rank instead of all-reducing mismatched buffers."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from accel_rl_amd.runners.sync import AccelRLSync
    from accel_rl_amd.util import logger
    logger.set_quiet(True)
    algo, policy, sampler = _FakeAlgo(), _FakePolicy(), _FakeSampler()

    def initialize(env_spec):
        n = 16 + 4 * rank
        policy.flat_params, policy.flat_grads, policy.n_params = torch.zeros(n), torch.zeros(n), n
    runner = AccelRLSync(algo=algo, policy=policy, sampler=sampler, n_steps=200, seed=7,
                         log_interval_steps=80, backend="gloo")
    runner.init_policy = initialize
    try:
        runner.startup()
        out.put((rank, "no error"))
    except ValueError as e:
        out.put((rank, str(e)))
    leave_group(dist)


def test_runners_of_one_clique_must_build_the_same_network():
    ctx = mp.get_context("spawn")
    out, port = ctx.Queue(), _free_port()
    codes, got = run_ranks(ctx, _mismatch_worker, [(r, 2, port, out) for r in range(2)], 120,
                           before_join=lambda: sorted(out.get(timeout=110) for _ in range(2)))
    assert codes == [0, 0], codes
    assert all("differ in size (16 ... 20 parameters" in msg for _, msg in got), got
