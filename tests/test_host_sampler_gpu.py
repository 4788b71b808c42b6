"""HostEnvSampler -- any rllab-style EnvCls stepped by worker processes on the host, served from the GPU -- against the
rollouts recorded from the reference's real multi-process sampler + real AtariEnv (fixture G7).  The environment class is
injected FROM THIS TEST: the oracle's port of AtariEnv over the synthetic emulator (oracle/ref_port.py, test
infrastructure); the product never imports it.  Every array the learner reads -- actions, prob, value, rewards, dones,
env_infos, every observation row, the bootstrap observations, the completed-trajectory multiset -- bit for bit."""
import ast

import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import ref_port as P
from test_sampler_gpu import DeviceTablePolicy, crc_rows

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


class PortEnv(P.PortedAtariEnv):
    """The port as an rllab Env: a spec with action / observation spaces (accel_rl/envs/base.py), step() info dict."""

    def __init__(self, **kw):
        super().__init__(**kw)
        from accel_rl_amd.spaces import Discrete, EnvSpec, UintBox
        self.action_space = Discrete(self.n_actions)
        self.observation_space = UintBox(shape=(self.n_stack, P.OBS_H, P.OBS_W), bits=8)
        self.spec = EnvSpec(self.observation_space, self.action_space)


def make(tag, device_env=False):
    from accel_rl_amd.sampler import ActsrvAltOvrlpSampler
    from accel_rl_amd.sampler.host_sampler import HostEnvSampler
    from accel_rl_amd.util import logger
    logger.set_quiet(True)
    g = load_golden("g7_rollout_" + tag)
    n_parallel, envs_per, horizon, n_batches, seed, mbr, maxlen = [int(x) for x in g["cfg"]]
    env_args = dict(ast.literal_eval(str(g["env_args"])))
    env_args["game"] = str(g["game"])
    smp = ActsrvAltOvrlpSampler(EnvCls=PortEnv, env_args=env_args, horizon=horizon, n_parallel=n_parallel, envs_per=envs_per,
                                mid_batch_reset=bool(mbr), max_path_length=np.inf if maxlen < 0 else maxlen,
                                max_decorrelation_steps=0, device=DEV)
    assert isinstance(smp, HostEnvSampler)                 # picked by the env class: no batched_device_env marker
    np.random.seed(seed)                                   # runner: set_seed(seed)
    smp.initialize(seed=seed + 1, affinities=dict(), discount=float(g["discount"]), need_extra_obs=True)
    smp.policy_init(DeviceTablePolicy(g["prob_table"], g["value_table"]))
    return g, smp, horizon, n_batches, bool(mbr)


@pytest.mark.parametrize("tag", ["breakout", "pong_maxlen", "seaquest_nomid", "breakout_noop0"])
def test_host_sampler_reproduces_the_reference_rollout(tag):
    g, smp, t, n_batches, mbr = make(tag)
    try:
        assert smp.alternating is True and smp.total_n_envs == smp.sample_size // t
        traj = []
        for b in range(n_batches):
            buf, infos = smp.obtain_samples(b)
            msg = "%s batch %d" % (tag, b)
            assert buf.observations.device.type == "cuda" and buf.observations.dtype == torch.uint8
            np.testing.assert_array_equal(buf.actions.cpu().numpy(), g["actions"][b], err_msg=msg)
            np.testing.assert_array_equal(buf.agent_infos["prob"].cpu().numpy(), g["prob"][b], err_msg=msg)
            np.testing.assert_array_equal(buf.agent_infos["value"].cpu().numpy(), g["value"][b], err_msg=msg)
            # (no valids mask needed even without mid-batch resets: the stale rows of a frozen env are the reference's
            #  stale rows too -- this sampler skips exactly the writes the NonResetCollector skips)
            np.testing.assert_array_equal(buf.rewards.cpu().numpy(), g["rewards"][b], err_msg=msg)
            np.testing.assert_array_equal(buf.dones.cpu().numpy().astype(bool), g["dones"][b], err_msg=msg)
            np.testing.assert_array_equal(buf.env_infos["raw_reward"].cpu().numpy(), g["raw_reward"][b], err_msg=msg)
            np.testing.assert_array_equal(buf.env_infos["need_reset"].cpu().numpy().astype(bool), g["need_reset"][b], err_msg=msg)
            np.testing.assert_array_equal(crc_rows(buf.observations), g["obs_crc"][b], err_msg=msg)
            np.testing.assert_array_equal(crc_rows(buf.extra_observations), g["extra_crc"][b], err_msg=msg)
            if b == 0:
                np.testing.assert_array_equal(buf.observations[:, -1].cpu().numpy(), g["first_batch_newest_frames"])
            for ti in infos:
                traj.append((b, float(ti.Length), float(ti.Return), float(ti.RawReturn), float(ti.NonzeroRewards),
                             float(ti.DiscountedReturn)))
        want = sorted((int(b),) + tuple(float(x) for x in row) for b, row in zip(g["traj_batch"], g["traj"]))
        assert len(want) > 0 and sorted(traj) == want
    finally:
        smp.shutdown()


def test_host_sampler_feeds_the_device_learner():
    """BASELINE config 1's form inside the product: A2C on host-stepped environments, the learner untouched -- the same
    algo / policy / runner objects as with the device sampler, a finite loss and parameters that moved."""
    from accel_rl_amd.algos.pg.a2c import A2C
    from accel_rl_amd.policies.atari_cnn_policy import AtariCnnPolicy
    from accel_rl_amd.policies.atari_cnn_specs import cnn_specs
    from accel_rl_amd.runners.accel_rl import AccelRL
    from accel_rl_amd.sampler import ActsrvAltOvrlpSampler
    from accel_rl_amd.util import logger
    logger.set_quiet(True)
    sampler = ActsrvAltOvrlpSampler(EnvCls=PortEnv, env_args=dict(game="pong"), horizon=5, n_parallel=2, envs_per=4,
                                    mid_batch_reset=False, max_decorrelation_steps=20, device=DEV)
    policy = AtariCnnPolicy(**cnn_specs[0])
    runner = AccelRL(algo=A2C(), policy=policy, sampler=sampler, n_steps=16 * 5 * 6, seed=3, log_interval_steps=16 * 5 * 3)
    before = None
    orig = runner.init_logging

    def init_logging():
        nonlocal before
        orig()
        before = policy.flat_params.clone()
    runner.init_logging = init_logging
    runner.save_itr_snapshot = lambda itr: None
    runner.train()
    torch.cuda.synchronize()
    after = policy.flat_params
    assert torch.isfinite(after).all() and not torch.equal(before, after)
    assert runner.last_tabular["CumTotalSteps"] > 0 and not sampler.workers


def test_a_dead_worker_is_reported_not_waited_for():
    class Dies(PortEnv):
        steps = 0

        def step(self, a):
            Dies.steps += 1
            if Dies.steps > 3:
                raise RuntimeError("env crashed")
            return super().step(a)
    from accel_rl_amd.sampler.host_sampler import HostEnvSampler
    from accel_rl_amd.util import logger
    logger.set_quiet(True)
    g = load_golden("g7_rollout_breakout")
    smp = HostEnvSampler(EnvCls=Dies, env_args=dict(game="breakout"), horizon=5, n_parallel=1, envs_per=1,
                         max_decorrelation_steps=0, device=DEV)
    np.random.seed(1)
    smp.initialize(seed=2, discount=0.99, need_extra_obs=True)
    smp.policy_init(DeviceTablePolicy(g["prob_table"], g["value_table"]))
    try:
        with pytest.raises(RuntimeError, match="simulation worker"):
            smp.obtain_samples(0)
    finally:
        smp.shutdown()
