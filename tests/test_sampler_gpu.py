"""GPU parity of the whole rollout: GpuVecSampler (HIP env + sampling kernels
behind the reference's sampler interface) against
  (1) the golden rollouts recorded from the reference's real multi-process
      sampler + AtariEnv (fixture G7), and
  (2) the oracle's sequential sampler port at BASELINE sizes (256 / 1024 envs).
Everything compared here is integer / byte / copied-float data: bit-exact.
"""
import ast
import zlib

import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import ref_port as P

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def crc_rows(t):
    a = np.ascontiguousarray(t.cpu().numpy())
    return np.array([zlib.crc32(a[i].tobytes()) for i in range(len(a))], np.uint32)


class DeviceTablePolicy(object):
    """Device twin of the fixtures' table policy: key = pixel sum mod 64."""
    recurrent = False

    def __init__(self, prob_table, value_table, serves_rows=False):
        self.prob_table = torch.from_numpy(prob_table).to(DEV)
        self.value_table = torch.from_numpy(value_table).to(DEV)
        self.serves_rows = serves_rows       # True: the sampler hands over (rollout buffer, rows) and writes each
        self.row_calls = 0                   # stacked observation once (arl_env_step single_write)

    def _keys(self, obs):
        return obs.reshape(obs.shape[0], -1).sum(dim=1, dtype=torch.int64) % 64

    def reset(self, n_batch):
        pass

    def prob_value(self, obs, rows=None):
        if rows is not None:
            self.row_calls += 1
            obs = obs[rows.long()]
        k = self._keys(obs)
        return self.prob_table[k].contiguous(), self.value_table[k].contiguous()

    def get_action(self, ob):
        np.random.rand()             # the reference policy samples one action here
        return None, None


class HostTablePolicy(object):
    """numpy twin for the oracle sampler (same tables, global numpy RNG)."""

    def __init__(self, prob_table, value_table):
        self.prob_table, self.value_table = prob_table, value_table

    def get_actions(self, obs):
        k = obs.reshape(obs.shape[0], -1).astype(np.int64).sum(axis=1) % 64
        prob, value = self.prob_table[k], self.value_table[k]
        return P.sample_actions(prob, np.random.rand(len(k))), dict(prob=prob, value=value)


def make_gpu_sampler(game, horizon, n_parallel, envs_per, seed, mid_batch_reset, max_path_length,
                     env_kwargs, tables, discount, use_graph, serves_rows=False):
    from accel_rl_amd.envs.synthetic_atari import SynthAtariEnv
    from accel_rl_amd.sampler.gpu_sampler import GpuVecSampler
    from accel_rl_amd.util import logger
    logger.set_quiet(True)
    env_args = dict(env_kwargs)
    env_args["game"] = game
    smp = GpuVecSampler(EnvCls=SynthAtariEnv, env_args=env_args, horizon=horizon,
                        n_parallel=n_parallel, envs_per=envs_per, mid_batch_reset=mid_batch_reset,
                        max_path_length=max_path_length, max_decorrelation_steps=0,
                        device=DEV, use_graph=use_graph)
    np.random.seed(seed)                                   # runner: set_seed(seed)
    smp.initialize(seed=seed + 1, affinities=dict(), discount=discount, need_extra_obs=True)
    smp.policy_init(DeviceTablePolicy(*tables, serves_rows=serves_rows))
    return smp


@pytest.mark.parametrize("serves_rows", [False, True])
@pytest.mark.parametrize("use_graph", [False, True])
@pytest.mark.parametrize("tag", ["breakout", "pong_maxlen", "seaquest_nomid", "breakout_noop0"])
def test_gpu_sampler_matches_reference_rollout(tag, use_graph, serves_rows):
    """serves_rows: the policy reads the current observations as rows of the rollout buffer and the step kernel
    writes each stacked observation once (only with mid_batch_reset; 'seaquest_nomid' keeps step_obs current)."""
    g = load_golden("g7_rollout_" + tag)
    n_parallel, envs_per, horizon, n_batches, seed, mbr, maxlen = [int(x) for x in g["cfg"]]
    smp = make_gpu_sampler(str(g["game"]), horizon, n_parallel, envs_per, seed, bool(mbr),
                           np.inf if maxlen < 0 else maxlen,
                           dict(ast.literal_eval(str(g["env_args"]))),
                           (g["prob_table"], g["value_table"]), float(g["discount"]), use_graph, serves_rows)
    assert smp._single_write == (serves_rows and bool(mbr))
    t = horizon
    traj = []
    for b in range(n_batches):
        buf, infos = smp.obtain_samples(b)
        msg = "%s batch %d" % (tag, b)
        np.testing.assert_array_equal(buf.actions.cpu().numpy(), g["actions"][b], err_msg=msg)
        np.testing.assert_array_equal(buf.agent_infos["prob"].cpu().numpy(), g["prob"][b], err_msg=msg)
        np.testing.assert_array_equal(buf.agent_infos["value"].cpu().numpy(), g["value"][b], err_msg=msg)
        need = buf.env_infos["need_reset"].cpu().numpy().astype(bool)
        if mbr:
            valid = np.ones(len(need), bool)
        else:   # NonResetCollector leaves stale rows: compare under the valids mask
            valid = P.valid_mask(g["need_reset"][b].reshape(-1, t)).reshape(-1).astype(bool)
            np.testing.assert_array_equal(P.valid_mask(need.reshape(-1, t)).reshape(-1).astype(bool), valid)
        np.testing.assert_array_equal(buf.rewards.cpu().numpy()[valid], g["rewards"][b][valid], err_msg=msg)
        np.testing.assert_array_equal(buf.dones.cpu().numpy().astype(bool)[valid], g["dones"][b][valid], err_msg=msg)
        np.testing.assert_array_equal(buf.env_infos["raw_reward"].cpu().numpy()[valid], g["raw_reward"][b][valid])
        np.testing.assert_array_equal(need[valid], g["need_reset"][b][valid], err_msg=msg)
        np.testing.assert_array_equal(crc_rows(buf.observations)[valid], g["obs_crc"][b][valid], err_msg=msg)
        np.testing.assert_array_equal(crc_rows(buf.extra_observations), g["extra_crc"][b], err_msg=msg)
        if b == 0:
            np.testing.assert_array_equal(buf.observations[:, -1].cpu().numpy(),
                                          g["first_batch_newest_frames"])
        for ti in infos:
            traj.append((b, float(ti.Length), float(ti.Return), float(ti.RawReturn),
                         float(ti.NonzeroRewards), float(ti.DiscountedReturn)))
    want = sorted((int(b),) + tuple(float(x) for x in row) for b, row in zip(g["traj_batch"], g["traj"]))
    assert len(want) > 0 and sorted(traj) == want
    assert (smp.policy.row_calls > 0) == smp._single_write
    smp.shutdown()


@pytest.mark.parametrize("n_parallel,envs_per,game,n_batches", [
    (16, 8, "breakout", 4),       # BASELINE config 2: 256 envs, T=5
    (64, 8, "breakout", 2),       # BASELINE config 3: 1024 envs
    (4, 2, "pong", 3),            # config 1 size (16 envs), 6 actions
    (3, 5, "seaquest", 3),        # 18 actions, odd group sizes
])
def test_gpu_sampler_matches_oracle_at_baseline_sizes(n_parallel, envs_per, game, n_batches):
    seed, horizon = 17, 5
    rs = np.random.RandomState(77)
    n_act = {"breakout": 4, "pong": 6, "seaquest": 18}[game]
    logits = rs.randn(64, n_act) * 1.5
    p = np.exp(logits - logits.max(1, keepdims=True))
    tables = ((p / p.sum(1, keepdims=True)).astype(np.float32), (rs.randn(64) * 2).astype(np.float32))
    # short episodes so that resets, life losses and over-length all occur within a few batches
    kw = dict(max_start_noops=30)
    smp = make_gpu_sampler(game, horizon, n_parallel, envs_per, seed, True, 9, kw, tables, 0.99, True)

    ora = P.CpuSamplerPort(game, horizon, n_parallel, envs_per, max_path_length=9,
                           mid_batch_reset=True, env_kwargs=kw)
    np.random.seed(seed)
    ora.initialize(seed + 1, discount=0.99)
    shape = ora.step_obs.shape[1:]
    for _ in range(2):
        np.random.randint(low=0, high=255, size=shape, dtype=np.uint8)
        np.random.randint(n_act, dtype=np.uint8)
    np.random.randint(low=0, high=255, size=shape, dtype=np.uint8)
    np.random.rand()
    host_policy = HostTablePolicy(*tables)
    state = np.random.get_state()
    n_completed = 0
    for b in range(n_batches):
        # both sides draw the batch's uniforms from the same global stream position
        np.random.set_state(state)
        buf, infos = smp.obtain_samples(b)
        np.random.set_state(state)
        want, completed = ora.obtain_samples(host_policy)
        state = np.random.get_state()
        for key, got in (("actions", buf.actions), ("rewards", buf.rewards), ("dones", buf.dones),
                         ("raw_reward", buf.env_infos["raw_reward"]), ("need_reset", buf.env_infos["need_reset"]),
                         ("prob", buf.agent_infos["prob"]), ("value", buf.agent_infos["value"])):
            np.testing.assert_array_equal(got.cpu().numpy().astype(want[key].dtype), want[key],
                                          err_msg="%s batch %d" % (key, b))
        np.testing.assert_array_equal(buf.observations.cpu().numpy(), want["observations"])
        np.testing.assert_array_equal(buf.extra_observations.cpu().numpy(), want["extra_observations"])
        got_t = sorted((ti.Length, ti.Return, ti.RawReturn, ti.NonzeroRewards, ti.DiscountedReturn) for ti in infos)
        want_t = sorted(ti.as_tuple() for ti in completed)
        assert got_t == want_t
        n_completed += len(want_t)
    assert n_completed > 0
    smp.shutdown()


SUITE_GAMES = ["pong", "breakout", "seaquest", "space_invaders", "qbert", "beam_rider", "enduro", "ms_pacman"]


@pytest.mark.parametrize("game", SUITE_GAMES)
def test_gpu_sampler_matches_oracle_on_every_suite_game(game):
    _suite_game_against_the_oracle(game, None)


@pytest.mark.parametrize("game", ["breakout", "pong", "beam_rider", "seaquest"])
def test_suite_games_on_one_shared_action_space(game):
    """Config 4 as bench.py --suite runs it: eight games behind ONE policy head (a synchronous clique all-reduces one
    flat bucket), every game's minimal action set padded with NOOP to 18 entries (SynthAtariEnv(pad_actions_to=18)).
    The padded entries must behave as NOOP on the device exactly as in the oracle's env port."""
    _suite_game_against_the_oracle(game, 18)


def test_nearest_resample_mode_through_the_sampler():
    """SynthAtariEnv(resample="nearest") reaches every pixel path of the device sampler (fused step kernel, resets,
    bootstrap observation): observations bit-identical to the oracle's env port in the same mode."""
    _suite_game_against_the_oracle("breakout", None, extra_env=dict(resample="nearest"))


def _suite_game_against_the_oracle(game, pad, extra_env=None):
    """BASELINE config 4's workload: each of the 8 suite games at the per-GPU shard size (256 envs = 2 x 16 x 8,
    horizon 5), every array of 14 batches bit-identical to the oracle's sampler port.  Covers the 6- and 9-action
    sets and the 0 / 3 / 4 / 5 start-lives rules (envs/synthetic_atari.py GAMES); 70 steps with max_path_length 66
    so that life losses (first at emulator tick 251 = step 55-62, depending on the start no-ops) and over-length
    resets both occur (asserted)."""
    from accel_rl_amd.envs.synthetic_atari import GAMES
    seed, horizon, n_parallel, envs_per, n_batches, max_len = 23, 5, 16, 8, 14, 66
    n_act = len(GAMES[game][1]) if pad is None else pad
    rs = np.random.RandomState(100 + GAMES[game][0])
    logits = rs.randn(64, n_act) * 1.5
    p = np.exp(logits - logits.max(1, keepdims=True))
    tables = ((p / p.sum(1, keepdims=True)).astype(np.float32), (rs.randn(64) * 2).astype(np.float32))
    kw = dict(max_start_noops=30) if pad is None else dict(max_start_noops=30, pad_actions_to=pad)
    kw.update(extra_env or {})
    smp = make_gpu_sampler(game, horizon, n_parallel, envs_per, seed, True, max_len, kw, tables, 0.99, True,
                           serves_rows=GAMES[game][0] % 2 == 0)
    assert smp.env_spec.action_space.n == n_act
    ora = P.CpuSamplerPort(game, horizon, n_parallel, envs_per, max_path_length=max_len, mid_batch_reset=True, env_kwargs=kw)
    np.random.seed(seed)
    ora.initialize(seed + 1, discount=0.99)
    shape = ora.step_obs.shape[1:]
    for _ in range(2):                                   # the master's step-buffer examples (act_server/buffers.py:24-30)
        np.random.randint(low=0, high=255, size=shape, dtype=np.uint8)
        np.random.randint(n_act, dtype=np.uint8)
    np.random.randint(low=0, high=255, size=shape, dtype=np.uint8)
    np.random.rand()
    host_policy = HostTablePolicy(*tables)
    state = np.random.get_state()
    n_completed, n_done, n_need = 0, 0, 0
    for b in range(n_batches):
        np.random.set_state(state)
        buf, infos = smp.obtain_samples(b)
        np.random.set_state(state)
        want, completed = ora.obtain_samples(host_policy)
        state = np.random.get_state()
        for key, got in (("actions", buf.actions), ("rewards", buf.rewards), ("dones", buf.dones),
                         ("raw_reward", buf.env_infos["raw_reward"]), ("need_reset", buf.env_infos["need_reset"]),
                         ("prob", buf.agent_infos["prob"]), ("value", buf.agent_infos["value"])):
            np.testing.assert_array_equal(got.cpu().numpy().astype(want[key].dtype), want[key],
                                          err_msg="%s %s batch %d" % (game, key, b))
        np.testing.assert_array_equal(buf.observations.cpu().numpy(), want["observations"])
        np.testing.assert_array_equal(buf.extra_observations.cpu().numpy(), want["extra_observations"])
        got_t = sorted((ti.Length, ti.Return, ti.RawReturn, ti.NonzeroRewards, ti.DiscountedReturn) for ti in infos)
        assert got_t == sorted(ti.as_tuple() for ti in completed)
        n_completed += len(got_t)
        n_done += int(want["dones"].sum())
        n_need += int(want["need_reset"].sum())
        assert int(want["actions"].max()) == n_act - 1   # the whole action set is in play
    assert n_completed >= 256 and n_need >= 256                   # every env ran over length once
    if GAMES[game][2] > 0:
        assert n_done > n_need                                    # life losses: done without need_reset
    smp.shutdown()


def test_decorrelation_runs_and_desynchronises():
    rs = np.random.RandomState(1)
    p = np.full((64, 4), 0.25, np.float32)
    smp = make_gpu_sampler("breakout", 5, 8, 4, 3, True, np.inf, dict(), (p, rs.randn(64).astype(np.float32)), 0.99, False)
    ticks0 = smp._st.tick.cpu().numpy().copy()
    smp.max_decorrelation_steps = 300
    with torch.cuda.device(smp.device):
        smp._decorrelate()
    ticks = smp._st.tick.cpu().numpy()
    assert len(np.unique(ticks)) > len(np.unique(ticks0))
    assert smp._st.traj_len.cpu().numpy().max() > 10
    buf, _ = smp.obtain_samples(0)
    assert buf.observations.shape == (64 * 5, 4, 104, 80)


def test_unannounced_mid_batch_reset_is_reported():
    """arl_env_step ranks a stream's resets from the previous launch's forecast; a reset the forecast did not announce
    (here: the length limit drops between two eager batches) would misorder the stream's no-op draws -- the kernel
    counts it (epoch[2]) and the sampler refuses the batch."""
    rs = np.random.RandomState(2)
    p = np.full((64, 4), 0.25, np.float32)
    smp = make_gpu_sampler("breakout", 5, 4, 4, 5, True, 1000, dict(), (p, rs.randn(64).astype(np.float32)), 0.99, False)
    buf, infos = smp.obtain_samples(0)
    list(infos)                                                   # resolves the batch: nothing to report
    assert int(smp._st.epoch[2].item()) == 0
    smp.max_path_length = 3                                       # every env is already longer: all reset at once
    buf, infos = smp.obtain_samples(1)
    assert int(smp._st.epoch[2].item()) > 0
    with pytest.raises(RuntimeError, match="not announced"):
        len(infos)
    smp._pending = None
    smp.shutdown()


# ---- evaluation sampler + AccelRLEval ------------------------------------------

@pytest.mark.parametrize("use_graph", [False, True])
@pytest.mark.parametrize("tag", ["breakout", "pong_nomid"])
def test_gpu_eval_sampler_matches_reference(tag, use_graph):
    """Training batches interleaved with evaluate_policy calls, recorded from the reference's own
    AAOEvalSampler (G13): the evaluation trajectories and every later training array are bit-identical
    (evaluation resets and action draws advance the shared worker / master RNG streams)."""
    from accel_rl_amd.envs.synthetic_atari import SynthAtariEnv
    from accel_rl_amd.sampler.gpu_sampler_with_eval import GpuVecEvalSampler
    from accel_rl_amd.util import logger
    logger.set_quiet(True)
    g = load_golden("g13_eval_" + tag)
    n_parallel, envs_per, horizon, n_batches, seed, mbr, maxlen, eval_steps, eval_per, eval_h = [int(x) for x in g["cfg"]]
    env_args = dict(ast.literal_eval(str(g["env_args"])))
    env_args["game"] = str(g["game"])
    smp = GpuVecEvalSampler(eval_steps=eval_steps, eval_envs_per=eval_per, EnvCls=SynthAtariEnv, env_args=env_args,
                            horizon=horizon, n_parallel=n_parallel, envs_per=envs_per, mid_batch_reset=bool(mbr),
                            max_path_length=np.inf if maxlen < 0 else maxlen, max_decorrelation_steps=0,
                            device=DEV, use_graph=use_graph)
    assert smp.eval_horizon == eval_h
    np.random.seed(seed)
    smp.initialize(seed=seed + 1, affinities=dict(), discount=float(g["discount"]), need_extra_obs=True)
    smp.policy_init(DeviceTablePolicy(g["prob_table"], g["value_table"]))
    eval_at = set(int(x) for x in g["eval_batches"])
    t, traj, eval_traj = horizon, [], []
    row = lambda b, ti: (b, float(ti.Length), float(ti.Return), float(ti.RawReturn), float(ti.NonzeroRewards),     # noqa: E731
                         float(ti.DiscountedReturn))
    for b in range(n_batches):
        if b in eval_at:
            eval_traj += [row(b, ti) for ti in smp.evaluate_policy(b)]
        buf, infos = smp.obtain_samples(b)
        msg = "%s batch %d" % (tag, b)
        np.testing.assert_array_equal(buf.actions.cpu().numpy(), g["actions"][b], err_msg=msg)
        np.testing.assert_array_equal(buf.agent_infos["prob"].cpu().numpy(), g["prob"][b], err_msg=msg)
        need = buf.env_infos["need_reset"].cpu().numpy().astype(bool)
        valid = np.ones(len(need), bool)
        if not mbr:
            valid = P.valid_mask(g["need_reset"][b].reshape(-1, t)).reshape(-1).astype(bool)
        np.testing.assert_array_equal(buf.rewards.cpu().numpy()[valid], g["rewards"][b][valid], err_msg=msg)
        np.testing.assert_array_equal(buf.dones.cpu().numpy().astype(bool)[valid], g["dones"][b][valid], err_msg=msg)
        np.testing.assert_array_equal(need[valid], g["need_reset"][b][valid], err_msg=msg)
        np.testing.assert_array_equal(crc_rows(buf.observations)[valid], g["obs_crc"][b][valid], err_msg=msg)
        np.testing.assert_array_equal(crc_rows(buf.extra_observations), g["extra_crc"][b], err_msg=msg)
        traj += [row(b, ti) for ti in infos]
    as_rows = lambda bs, rows: sorted((int(b),) + tuple(float(x) for x in r) for b, r in zip(bs, rows))     # noqa: E731
    assert sorted(traj) == as_rows(g["traj_batch"], g["traj"])
    assert sorted(eval_traj) == as_rows(g["eval_at"], g["eval_traj"]) and len(eval_traj) >= 10
    smp.shutdown()


@pytest.mark.parametrize("use_graph", [False, True])
def test_eval_sampler_with_a_length_limit_below_the_decorrelation_walk(use_graph):
    """The start-up walk resets at Length > L (sampler/util.py:50), the served steps of the eval sampler family at
    Length >= L (worker_with_eval.py:48): with L = 25 and up to 1000 decorrelation steps about one env in 26 leaves
    the walk at Length L - 1 and must reset in the first served launch.  That reset is announced (the forecast is
    restated under the served rule after the walk), so no batch is refused, no trajectory outlives L, and every
    completed trajectory of the first batches that hit the limit has Length == L exactly."""
    from accel_rl_amd.envs.synthetic_atari import SynthAtariEnv
    from accel_rl_amd.sampler.gpu_sampler_with_eval import GpuVecEvalSampler
    from accel_rl_amd.util import logger
    logger.set_quiet(True)
    L = 25
    smp = GpuVecEvalSampler(eval_steps=64 * 30, eval_envs_per=1, EnvCls=SynthAtariEnv, env_args=dict(game="pong"),
                            horizon=5, n_parallel=8, envs_per=16, mid_batch_reset=True, max_path_length=L,
                            max_decorrelation_steps=1000, device=DEV, use_graph=use_graph)
    np.random.seed(5)
    smp.initialize(seed=9, affinities=dict(), discount=0.99, need_extra_obs=True)
    n = smp.total_n_envs
    lens = smp._st.traj_len.cpu().numpy()
    assert lens.max() <= L and (lens == L - 1).sum() >= 1, "the case under test: an env one step short of the limit"
    rs = np.random.RandomState(0)
    smp.policy_init(DeviceTablePolicy(np.full((64, 6), 1. / 6, np.float32), rs.randn(64).astype(np.float32)))
    seen = []
    for b in range(8):
        if b == 3:
            ev = smp.evaluate_policy(b)
            assert all(ti.Length <= L for ti in ev)
        buf, infos = smp.obtain_samples(b)
        seen += [ti.Length for ti in infos]                        # raises if a reset was not announced
        assert int(smp._st.traj_len.max().item()) < L
    assert int(smp._st.epoch[2].item()) == 0
    # an env that LEFT the walk at Length L (the walk ends episodes at Length > L) is one step over the served rule's
    # limit when its first served step ends it: Length L + 1, as in the reference; every other episode ends at L
    assert len(seen) >= n and set(seen) <= {L, L + 1} and seen.count(L + 1) == int((lens == L).sum())
    smp.shutdown()


def test_eval_runner_trains_and_logs():
    """AccelRLEval (accel_rl/runners/accel_rl.py:108-180): evaluation every log interval, its tabular keys."""
    from accel_rl_amd.algos.pg.ppo import PPO
    from accel_rl_amd.envs.synthetic_atari import SynthAtariEnv
    from accel_rl_amd.policies.atari_cnn_policy import AtariCnnPolicy
    from accel_rl_amd.policies.atari_cnn_specs import cnn_specs
    from accel_rl_amd.runners.accel_rl import AccelRLEval
    from accel_rl_amd.sampler.gpu_sampler_with_eval import GpuVecEvalSampler
    from accel_rl_amd.util import logger
    logger.set_quiet(True)
    sampler = GpuVecEvalSampler(eval_steps=16 * 60, eval_envs_per=2, EnvCls=SynthAtariEnv,
                                env_args=dict(game="breakout"), horizon=5, n_parallel=4, envs_per=4,
                                max_path_length=30, max_decorrelation_steps=0, device=DEV)
    algo = PPO(optimizer_args=dict(minibatch_size=64, epochs=2))
    runner = AccelRLEval(algo=algo, policy=AtariCnnPolicy(**cnn_specs[0]), sampler=sampler, n_steps=160 * 8,
                         seed=3, eval_interval_steps=640)
    runner.train()
    tab = runner.last_tabular
    for key in ("Iteration", "CumCompletedSteps", "StepsInEval", "TrajsInEval", "LengthAverage", "ReturnAverage",
                "GradNormAverage", "ParamsNorm", "NormFromInit", "CumTrainTime", "CumEvalTime", "CumTotalTime",
                "SamplesPerSecond"):
        assert key in tab, key
    assert tab["Iteration"] == 8 and tab["TrajsInEval"] >= 16 and tab["LengthAverage"] == 30
    assert tab["StepsInEval"] == tab["TrajsInEval"] * 30 and tab["SamplesPerSecond"] > 0


# ---- recurrent policies through the sampler ---------------------------------

class DeviceRecurrentTablePolicy(object):
    """Device twin of the G14 stand-in policy: h' = 0.5 h + [key/64, 1]; tables indexed by
    (key + floor(4 h[0])) mod 64.  One state row per env (no alternation on the device)."""
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

    def _forward(self, obs, h):
        key = obs.reshape(obs.shape[0], -1).sum(dim=1, dtype=torch.int64) % 64
        idx = (key + torch.floor(4 * h[:, 0]).to(torch.int64)) % 64
        add = torch.stack([key.to(torch.float32) / 64, torch.ones_like(h[:, 1])], dim=1)
        return self.prob_table[idx].contiguous(), self.value_table[idx].contiguous(), 0.5 * h + add

    def act_step(self, obs):
        hp = self._h.clone()
        prob, value, new_h = self._forward(obs, self._h)
        self._h.copy_(new_h)
        return prob, value, hp

    def prob_value(self, obs):
        prob, value, _ = self._forward(obs, self._h)
        return prob, value

    def get_action(self, ob):
        np.random.rand()
        self.act_step(ob[None])
        return None, None


@pytest.mark.parametrize("use_graph", [False, True])
def test_gpu_sampler_recurrent_plumbing_matches_reference(use_graph):
    """G14: the reference's real sampler driving a recurrent stand-in policy -- stored previous hidden
    states at every (env, step), reset timing, final policy state; bit-identical."""
    from accel_rl_amd.envs.synthetic_atari import SynthAtariEnv
    from accel_rl_amd.sampler.gpu_sampler import GpuVecSampler
    from accel_rl_amd.util import logger
    logger.set_quiet(True)
    g = load_golden("g14_recurrent_seaquest")
    n_parallel, envs_per, horizon, n_batches, seed, mbr, maxlen = [int(x) for x in g["cfg"]]
    env_args = dict(ast.literal_eval(str(g["env_args"])))
    env_args["game"] = str(g["game"])
    smp = GpuVecSampler(EnvCls=SynthAtariEnv, env_args=env_args, horizon=horizon, n_parallel=n_parallel,
                        envs_per=envs_per, mid_batch_reset=False, max_path_length=maxlen,
                        max_decorrelation_steps=0, device=DEV, use_graph=use_graph)
    np.random.seed(seed)
    smp.initialize(seed=seed + 1, affinities=dict(), discount=float(g["discount"]), need_extra_obs=True)
    policy = DeviceRecurrentTablePolicy(g["prob_table"], g["value_table"])
    smp.policy_init(policy)
    t = horizon
    for b in range(n_batches):
        buf, infos = smp.obtain_samples(b)
        len(infos)
        msg = "batch %d" % b
        valid = P.valid_mask(g["need_reset"][b].reshape(-1, t)).reshape(-1).astype(bool)
        np.testing.assert_array_equal(buf.actions.cpu().numpy(), g["actions"][b], err_msg=msg)
        np.testing.assert_array_equal(buf.agent_infos["prob"].cpu().numpy(), g["prob"][b], err_msg=msg)
        np.testing.assert_array_equal(buf.agent_infos["hprev_0"].cpu().numpy(), g["hprev"][b], err_msg=msg)
        np.testing.assert_array_equal(buf.rewards.cpu().numpy()[valid], g["rewards"][b][valid], err_msg=msg)
        np.testing.assert_array_equal(crc_rows(buf.observations)[valid], g["obs_crc"][b][valid], err_msg=msg)
        np.testing.assert_array_equal(crc_rows(buf.extra_observations), g["extra_crc"][b], err_msg=msg)
        np.testing.assert_array_equal(policy._h.cpu().numpy(), g["state_after"][b], err_msg=msg)
    smp.shutdown()


def test_batches_enqueued_ahead_of_their_records():
    """Two host-side buffer sets (round 6): a caller may enqueue batch i + 1 before it has read batch i's trajectory
    records.  Whatever the order in which the lazily read records are looked at, every batch reports exactly what a run
    that reads each batch at once reports (served step, hipGraph), and the no-op ring's top-ups stay in step."""
    from accel_rl_amd.envs.synthetic_atari import SynthAtariEnv
    from accel_rl_amd.policies.atari_cnn_policy import AtariCnnPolicy
    from accel_rl_amd.policies.atari_cnn_specs import cnn_specs
    from accel_rl_amd.sampler.gpu_sampler import GpuVecSampler
    from accel_rl_amd.util import logger
    logger.set_quiet(True)

    def run(lazy):
        smp = GpuVecSampler(EnvCls=SynthAtariEnv, env_args=dict(game="breakout", max_start_noops=30), horizon=5, n_parallel=2,
                            envs_per=4, mid_batch_reset=True, max_path_length=3, max_decorrelation_steps=0, device=DEV,
                            use_graph=True)
        np.random.seed(5)
        env_spec, *_ = smp.initialize(seed=6, affinities=dict(), discount=0.99, need_extra_obs=True)
        from accel_rl_amd.util.seed import set_seed
        set_seed(7)                                 # (numpy's stream AND the conv initialiser's private one)
        policy = AtariCnnPolicy(**cnn_specs[1])
        policy.initialize(env_spec, device=DEV)
        with torch.no_grad():                       # a sharp head: the actions depend on the network, not just on the draws
            policy.flat_params[policy._offsets[policy._k_head]:].mul_(40.0)
        smp.policy_init(policy)
        assert len(smp._sets) == 2
        out, held = [], []
        for b in range(40):
            buf, infos = smp.obtain_samples(b)
            crc = zlib.crc32(buf.actions.cpu().numpy().tobytes())        # (reads the device: the batch itself is complete)
            if lazy:
                held.append((b, crc, infos))
                if len(held) == 2:                  # the newer batch first: it reads the older one's records before its own
                    for bb, cc, inf in reversed(held):
                        out.append((bb, cc, sorted((ti._env, ti.Length, ti.Return) for ti in inf)))
                    held = []
            else:
                out.append((b, crc, sorted((ti._env, ti.Length, ti.Return) for ti in infos)))
        cursors = smp._st.noop_cursor.cpu().numpy().copy()
        smp.shutdown()
        return sorted(out), cursors

    eager, c0 = run(False)
    lazy, c1 = run(True)
    assert eager == lazy and sum(len(t) for _, _, t in eager) > 100
    np.testing.assert_array_equal(c0, c1)
