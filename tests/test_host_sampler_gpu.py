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
    try:
        smp.policy_init(DeviceTablePolicy(g["prob_table"], g["value_table"]))
        with pytest.raises(RuntimeError, match="simulation worker"):
            smp.obtain_samples(0)
    finally:
        smp.shutdown()


# ---- round 6: evaluation variant, recurrent state and epsilon-greedy serving on host environments ---------------------

@pytest.mark.parametrize("tag", ["breakout", "pong_nomid"])
def test_host_eval_sampler_reproduces_the_reference(tag):
    """G13 (the reference's own AAOEvalSampler: training batches interleaved with evaluate_policy calls) THROUGH
    HostEnvEvalSampler with the AtariEnv port injected from here: evaluation trajectories and every later training array
    bit for bit (evaluation envs share their worker's RNG stream; evaluation action draws advance the master's)."""
    from accel_rl_amd.sampler import AAOEvalSampler
    from accel_rl_amd.sampler.host_sampler import HostEnvEvalSampler
    from accel_rl_amd.util import logger
    logger.set_quiet(True)
    g = load_golden("g13_eval_" + tag)
    n_parallel, envs_per, horizon, n_batches, seed, mbr, maxlen, eval_steps, eval_per, eval_h = [int(x) for x in g["cfg"]]
    env_args = dict(ast.literal_eval(str(g["env_args"])))
    env_args["game"] = str(g["game"])
    smp = AAOEvalSampler(eval_steps=eval_steps, eval_envs_per=eval_per, EnvCls=PortEnv, env_args=env_args, horizon=horizon,
                         n_parallel=n_parallel, envs_per=envs_per, mid_batch_reset=bool(mbr),
                         max_path_length=np.inf if maxlen < 0 else maxlen, max_decorrelation_steps=0, device=DEV)
    assert isinstance(smp, HostEnvEvalSampler) and smp.eval_horizon == eval_h
    try:
        np.random.seed(seed)
        smp.initialize(seed=seed + 1, affinities=dict(), discount=float(g["discount"]), need_extra_obs=True)
        smp.policy_init(DeviceTablePolicy(g["prob_table"], g["value_table"]))
        eval_at = set(int(x) for x in g["eval_batches"])
        traj, eval_traj = [], []
        row = lambda b, ti: (b, float(ti.Length), float(ti.Return), float(ti.RawReturn), float(ti.NonzeroRewards),     # noqa: E731
                             float(ti.DiscountedReturn))
        for b in range(n_batches):
            if b in eval_at:
                eval_traj += [row(b, ti) for ti in smp.evaluate_policy(b)]
            buf, infos = smp.obtain_samples(b)
            msg = "%s batch %d" % (tag, b)
            np.testing.assert_array_equal(buf.actions.cpu().numpy(), g["actions"][b], err_msg=msg)
            np.testing.assert_array_equal(buf.agent_infos["prob"].cpu().numpy(), g["prob"][b], err_msg=msg)
            np.testing.assert_array_equal(buf.agent_infos["value"].cpu().numpy(), g["value"][b], err_msg=msg)
            # (stale rows of frozen envs included: this sampler skips exactly the writes the NonResetCollector skips)
            np.testing.assert_array_equal(buf.rewards.cpu().numpy(), g["rewards"][b], err_msg=msg)
            np.testing.assert_array_equal(buf.dones.cpu().numpy().astype(bool), g["dones"][b], err_msg=msg)
            np.testing.assert_array_equal(buf.env_infos["raw_reward"].cpu().numpy(), g["raw_reward"][b], err_msg=msg)
            np.testing.assert_array_equal(buf.env_infos["need_reset"].cpu().numpy().astype(bool), g["need_reset"][b], err_msg=msg)
            np.testing.assert_array_equal(crc_rows(buf.observations), g["obs_crc"][b], err_msg=msg)
            np.testing.assert_array_equal(crc_rows(buf.extra_observations), g["extra_crc"][b], err_msg=msg)
            traj += [row(b, ti) for ti in infos]
        as_rows = lambda bs, rows: sorted((int(b),) + tuple(float(x) for x in r) for b, r in zip(bs, rows))     # noqa: E731
        assert sorted(traj) == as_rows(g["traj_batch"], g["traj"])
        assert sorted(eval_traj) == as_rows(g["eval_at"], g["eval_traj"]) and len(eval_traj) >= 10
    finally:
        smp.shutdown()


class GroupedRecurrentTablePolicy(object):
    """Device twin of the G14 stand-in policy (h' = 0.5 h + [key/64, 1]; tables indexed by (key + floor(4 h[0])) mod 64)
    with the serving protocol of a sampler that serves its envs in groups: act_step(observations, rows=(lo, hi))."""
    recurrent = True
    state_info_keys = ["hprev_0"]

    def __init__(self, prob_table, value_table):
        self.prob_table = torch.from_numpy(prob_table).to(DEV)
        self.value_table = torch.from_numpy(value_table).to(DEV)
        self._h = None

    def reset(self, n_batch):
        self._h = torch.zeros((n_batch, 2), dtype=torch.float32, device=DEV)

    def get_prev_hiddens(self):
        return [self._h]

    def reset_rows(self, mask_u8):
        self._h.mul_((mask_u8 == 0).to(torch.float32).unsqueeze(1))

    def act_step(self, obs, rows=None):
        h = self._h if rows is None else self._h[rows[0]:rows[1]]
        hp = h.clone()
        key = obs.reshape(obs.shape[0], -1).sum(dim=1, dtype=torch.int64) % 64
        idx = (key + torch.floor(4 * h[:, 0]).to(torch.int64)) % 64
        add = torch.stack([key.to(torch.float32) / 64, torch.ones_like(h[:, 1])], dim=1)
        h.copy_(0.5 * h + add)
        return self.prob_table[idx].contiguous(), self.value_table[idx].contiguous(), hp

    def get_action(self, ob):
        np.random.rand()
        self.act_step(ob[None])
        return None, None


def test_host_sampler_recurrent_plumbing_matches_reference():
    """G14 (the reference's real sampler driving a recurrent stand-in policy on its BaseRecurrentPolicy state handling)
    THROUGH HostEnvSampler: the previous hidden state stored at every (env, step), the timing of reset_one from the
    workers' flags, the policy's state after every batch -- bit for bit, frozen envs' stale rows included."""
    from accel_rl_amd.sampler import ActsrvAltOvrlpSampler
    from accel_rl_amd.util import logger
    logger.set_quiet(True)
    g = load_golden("g14_recurrent_seaquest")
    n_parallel, envs_per, horizon, n_batches, seed, mbr, maxlen = [int(x) for x in g["cfg"]]
    env_args = dict(ast.literal_eval(str(g["env_args"])))
    env_args["game"] = str(g["game"])
    smp = ActsrvAltOvrlpSampler(EnvCls=PortEnv, env_args=env_args, horizon=horizon, n_parallel=n_parallel, envs_per=envs_per,
                                mid_batch_reset=False, max_path_length=maxlen, max_decorrelation_steps=0, device=DEV)
    try:
        np.random.seed(seed)
        smp.initialize(seed=seed + 1, affinities=dict(), discount=float(g["discount"]), need_extra_obs=True)
        policy = GroupedRecurrentTablePolicy(g["prob_table"], g["value_table"])
        smp.policy_init(policy)
        resets = 0
        for b in range(n_batches):
            buf, infos = smp.obtain_samples(b)
            resets += len(infos)
            msg = "batch %d" % b
            np.testing.assert_array_equal(buf.actions.cpu().numpy(), g["actions"][b], err_msg=msg)
            np.testing.assert_array_equal(buf.agent_infos["prob"].cpu().numpy(), g["prob"][b], err_msg=msg)
            np.testing.assert_array_equal(buf.agent_infos["hprev_0"].cpu().numpy(), g["hprev"][b], err_msg=msg)
            np.testing.assert_array_equal(buf.rewards.cpu().numpy(), g["rewards"][b], err_msg=msg)
            np.testing.assert_array_equal(crc_rows(buf.observations), g["obs_crc"][b], err_msg=msg)
            np.testing.assert_array_equal(crc_rows(buf.extra_observations), g["extra_crc"][b], err_msg=msg)
            np.testing.assert_array_equal(policy._h.cpu().numpy(), g["state_after"][b], err_msg=msg)
        assert resets > 0, "the fixture must exercise reset_one"
    finally:
        smp.shutdown()


def _dqn_policy(epsilon):
    from accel_rl_amd.policies.atari_cnn_specs import cnn_specs
    from accel_rl_amd.policies.dqn.atari_dqn_policy import AtariDqnPolicy
    return AtariDqnPolicy(epsilon=epsilon, **cnn_specs[0])


@pytest.mark.parametrize("with_eval", [False, True])
def test_epsilon_greedy_serving_on_host_envs_matches_the_device_sampler(with_eval):
    """The DQN policies' epsilon-greedy serving (atari_dqn_policy.py:123-128: per (step, group) np.random.rand(B), then
    action_space.sample_n(#random)) on host environments: the SAME network, seeds and draws through HostEnvSampler with the
    AtariEnv port and through GpuVecSampler with the device emulator (itself pinned to G7 / G13) give the same actions,
    rewards, flags and observations, batch after batch -- also across evaluate_policy calls, which consume draws of the
    master's stream."""
    from accel_rl_amd.envs.synthetic_atari import SynthAtariEnv
    from accel_rl_amd.sampler import AAOEvalSampler, ActsrvAltOvrlpSampler
    from accel_rl_amd.util import logger
    logger.set_quiet(True)
    common = dict(env_args=dict(game="seaquest"), horizon=4, n_parallel=2, envs_per=2, mid_batch_reset=True,
                  max_path_length=23, max_decorrelation_steps=0, device=DEV)
    runs = []
    for env_cls in (SynthAtariEnv, PortEnv):
        if with_eval:
            smp = AAOEvalSampler(eval_steps=8 * 30, eval_envs_per=2, EnvCls=env_cls, **common)
        else:
            smp = ActsrvAltOvrlpSampler(EnvCls=env_cls, **common)
        try:
            from accel_rl_amd.util.seed import set_seed
            set_seed(77)                               # the runner's call (also seeds the conv initialiser's private stream)
            env_spec = smp.initialize(seed=78, affinities=dict(), discount=0.99, need_extra_obs=False)[0]
            policy = _dqn_policy(epsilon=0.35)
            policy.initialize(env_spec, device=DEV)
            smp.policy_init(policy)
            rec = []
            for b in range(7):
                if with_eval and b in (2, 5):
                    policy.set_epsilon(0.05)
                    ev = smp.evaluate_policy(b)
                    policy.set_epsilon(0.35)
                    rec.append(("eval", sorted((ti.Length, ti.Return, ti.RawReturn) for ti in ev)))
                buf, infos = smp.obtain_samples(b)
                rec.append((buf.actions.cpu().numpy().copy(), buf.rewards.cpu().numpy().copy(),
                            buf.dones.cpu().numpy().astype(bool), crc_rows(buf.observations),
                            sorted((ti.Length, ti.Return) for ti in infos)))
            runs.append((rec, policy.get_param_values()))
        finally:
            smp.shutdown()
    (dev_rec, dev_params), (host_rec, host_params) = runs
    np.testing.assert_array_equal(dev_params, host_params)          # same start-up draws -> the same network
    n_random = 0
    for i, (a, b) in enumerate(zip(dev_rec, host_rec)):
        if isinstance(a[0], str):
            assert a == b and len(a[1]) > 0, "evaluation %d" % i
            continue
        for x, y in zip(a[:4], b[:4]):
            np.testing.assert_array_equal(x, y, err_msg="record %d" % i)
        assert a[4] == b[4]
        n_random += 1
    assert n_random == 7


def test_dqn_trains_on_host_envs():
    """A reference DQN example on a host AtariEnv no longer stops at construction: the sampler family with evaluation,
    the epsilon-greedy policy and the replay-based learner run end to end on host-stepped environments."""
    from accel_rl_amd.algos.dqn.dqn import DQN
    from accel_rl_amd.runners.accel_rl import AccelRLEval
    from accel_rl_amd.sampler import AAOEvalSampler
    from accel_rl_amd.util import logger
    logger.set_quiet(True)
    sampler = AAOEvalSampler(eval_steps=4 * 100, eval_envs_per=1, EnvCls=PortEnv, env_args=dict(game="pong"), horizon=4,
                             n_parallel=2, envs_per=2, max_path_length=40, max_decorrelation_steps=0, device=DEV)
    policy = _dqn_policy(epsilon=1)
    algo = DQN(batch_size=16, min_steps_learn=64, replay_size=2048, training_intensity=4)
    runner = AccelRLEval(algo=algo, policy=policy, sampler=sampler, n_steps=32 * 12, seed=5, eval_interval_steps=32 * 6)
    runner.save_itr_snapshot = lambda itr: None
    runner.train()
    torch.cuda.synchronize()
    assert torch.isfinite(policy.flat_params).all() and not sampler.workers
    assert runner.last_tabular["TrajsInEval"] >= 4            # 100 evaluation steps per env, episodes end at Length 40
