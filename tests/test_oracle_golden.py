"""
Pins the oracle (oracle/ref_port.py, a CPU restatement) to golden vectors that
were produced by the reference's own code (tests/golden/gen_golden.py).
CPU-only; runs everywhere.
"""
import ast
import zlib

import numpy as np
import pytest

from conftest import load_golden
from oracle import ref_port as P


def _crc_rows(a):
    a = np.ascontiguousarray(a)
    return np.array([zlib.crc32(a[i].tobytes()) for i in range(len(a))], np.uint32)


# -- G1 / G2 ------------------------------------------------------------------

def _scan_cases():
    g = load_golden("g1_g2_scans")
    params = g["params"]
    for i in range(int(g["n_inputs"])):
        ik = "i%02d" % i
        for pi, (gam, lam) in enumerate(params):
            yield g, ik, pi, float(gam), float(lam)


def test_gae_bit_exact_vs_reference():
    n = 0
    for g, ik, pi, gam, lam in _scan_cases():
        adv, ret = P.gae_scan(g[ik + "_r"], g[ik + "_v"], g[ik + "_d"], g[ik + "_lv"], gam, lam)
        key = "%s_p%d" % (ik, pi)
        np.testing.assert_array_equal(adv, g[key + "_adv"], err_msg=key)
        np.testing.assert_array_equal(ret, g[key + "_ret"], err_msg=key)
        n += 1
    assert n == 96                      # 8 shapes (incl. SURVEY 8c: (16,5), (256,5), (64,128)) x 3 done rates x 4 (gamma, lambda)


def test_gae_legacy_promotion_bit_exact_and_close():
    for g, ik, pi, gam, lam in _scan_cases():
        key = "%s_p%d" % (ik, pi)
        adv, _ = P.gae_scan(g[ik + "_r"], g[ik + "_v"], g[ik + "_d"], g[ik + "_lv"], gam, lam,
                            promo="legacy")
        np.testing.assert_array_equal(adv, g[key + "_adv_legacy"], err_msg=key)
        # the two promotions agree within the north_star tolerance
        tol = 1e-5 * np.maximum(1, np.abs(g[key + "_adv"]))
        assert np.all(np.abs(adv - g[key + "_adv"]) <= tol), key


def test_nstep_bit_exact_vs_reference():
    n = 0
    for g, ik, pi, gam, lam in _scan_cases():
        key = "%s_p%d" % (ik, pi)
        if key + "_nret" not in g:
            continue
        ret, adv = P.nstep_returns(g[ik + "_r"], g[ik + "_d"], g[ik + "_v"], g[ik + "_lv"], gam)
        np.testing.assert_array_equal(ret, g[key + "_nret"], err_msg=key)
        np.testing.assert_array_equal(adv, g[key + "_nadv"], err_msg=key)
        ret64, _ = P.nstep_returns(g[ik + "_r"], g[ik + "_d"], g[ik + "_v"], g[ik + "_lv"], gam,
                                   promo="legacy")
        np.testing.assert_array_equal(ret64, g[key + "_nret_legacy"], err_msg=key)
        tol = 1e-5 * np.maximum(1, np.abs(ret))
        assert np.all(np.abs(ret64 - ret) <= tol), key
        n += 1
    assert n == 72


# -- G3 -----------------------------------------------------------------------

def test_valids_and_zeroing():
    g = load_golden("g3_valids")
    for c in range(int(g["n_cases"])):
        k = "c%02d" % c
        valids = P.valid_mask(g[k + "_flags"])
        np.testing.assert_array_equal(valids, g[k + "_valids"], err_msg=k)
        a, r, v = P.zero_invalid(valids, g[k + "_adv"], g[k + "_ret"], g[k + "_val"])
        np.testing.assert_array_equal(a, g[k + "_adv_z"])
        np.testing.assert_array_equal(r, g[k + "_ret_z"])
        np.testing.assert_array_equal(v, g[k + "_val_z"])


# -- G4 -----------------------------------------------------------------------

def test_process_samples():
    g = load_golden("g4_process_samples")
    for c in range(int(g["n_cases"])):
        k = "c%02d" % c
        gam, lam, use_valids, std_adv = g[k + "_cfg"]
        out = P.process_samples(g[k + "_r"], g[k + "_d"], g[k + "_v"], g[k + "_lv"],
                                g[k + "_need"], float(gam), float(lam) if lam != 1 else 1,
                                use_valids=bool(use_valids), standardize_adv=bool(std_adv))
        np.testing.assert_array_equal(out["returns"], g[k + "_ret"], err_msg=k)
        np.testing.assert_array_equal(out["advantages"], g[k + "_adv"], err_msg=k)
        if use_valids:
            np.testing.assert_array_equal(out["valids"], g[k + "_valids"])
            np.testing.assert_array_equal(out["value"], g[k + "_value_after"])


# -- G5 -----------------------------------------------------------------------

def test_action_sampling_bit_exact():
    g = load_golden("g5_sampling")
    for c in range(int(g["n_cases"])):
        k = "c%02d" % c
        acts = P.sample_actions(g[k + "_prob"], g[k + "_u"])
        assert acts.dtype == g[k + "_act"].dtype
        np.testing.assert_array_equal(acts, g[k + "_act"], err_msg=k)
        # and the uniforms are what numpy's legacy global stream yields
        rs = np.random.RandomState(int(g[k + "_seed"]))
        np.testing.assert_array_equal(rs.rand(len(acts)), g[k + "_u"])


# -- G6 -----------------------------------------------------------------------

@pytest.mark.parametrize("gi", range(4))
def test_env_port_matches_reference_env(gi):
    g = load_golden("g6_env")
    k = "g%d" % gi
    kwargs = dict(ast.literal_eval(str(g[k + "_kwargs"])))
    rng = np.random.RandomState(int(g[k + "_seed"]))
    env = P.PortedAtariEnv(game=str(g[k + "_game"]), rng=rng, **kwargs)
    obs = env.reset()
    keep = {int(i): j for j, i in enumerate(g[k + "_keep_idx"])}
    np.testing.assert_array_equal(obs[-1], g[k + "_keep_last"][keep[-1]])
    for i, a in enumerate(g[k + "_acts"]):
        o, r, d, info = env.step(a)
        assert r == g[k + "_rew"][i] and type(r) is np.float32, (i, r)
        assert d == g[k + "_done"][i], i
        assert info.get("raw_reward", r) == g[k + "_raw"][i], i
        assert info.get("need_reset", d) == g[k + "_need"][i], i
        if d and info.get("need_reset", True):
            assert g[k + "_reset"][i]
            o = env.reset()
        assert env.tick == g[k + "_tick"][i], i
        assert zlib.crc32(o.tobytes()) == g[k + "_crc"][i], i
        if i in keep:
            np.testing.assert_array_equal(o[-1], g[k + "_keep_last"][keep[i]])
            np.testing.assert_array_equal([bool(f.any()) for f in o],
                                          g[k + "_keep_nonzero"][keep[i]])
    assert g[k + "_done"].sum() > 0


# -- G7 -----------------------------------------------------------------------

class TablePolicyPort(object):
    """The fixtures' table policy (tests/golden/gen_golden.py:TablePolicy), with
    sampling through the oracle's sample_actions + the global numpy RNG."""

    def __init__(self, prob_table, value_table):
        self.prob_table, self.value_table = prob_table, value_table

    @staticmethod
    def keys(obs):
        obs = np.asarray(obs)
        return obs.reshape(obs.shape[0], -1).astype(np.int64).sum(axis=1) % 64

    def get_actions(self, obs):
        k = self.keys(obs)
        prob, value = self.prob_table[k], self.value_table[k]
        return P.sample_actions(prob, np.random.rand(len(k))), dict(prob=prob, value=value)


def replay_rollout(g, sampler_factory=None):
    n_parallel, envs_per, horizon, n_batches, seed, mbr, maxlen = [int(x) for x in g["cfg"][:7]]
    env_kwargs = dict(ast.literal_eval(str(g["env_args"])))
    extra = dict()
    if len(g["cfg"]) > 7:                               # G13: the evaluation-sampler variant
        extra = dict(eval_steps=int(g["cfg"][7]), eval_envs_per=int(g["cfg"][8]))
    smp = P.CpuSamplerPort(str(g["game"]), horizon, n_parallel, envs_per,
                           max_path_length=np.inf if maxlen < 0 else maxlen,
                           mid_batch_reset=bool(mbr), env_kwargs=env_kwargs, **extra)
    np.random.seed(seed)
    n_act, sample_size = smp.initialize(seed + 1, discount=float(g["discount"]))
    # master-side draws the reference makes between initialize and the first serve
    # (build_step_buffer x2 [+ x2 for the eval step buffers, sampler.py:204-206]:
    #  act_server/buffers.py:24-30; build_policy_buffer: :33-38)
    obs_shape = smp.step_obs.shape[1:]
    for _ in range(4 if extra else 2):
        np.random.randint(low=0, high=255, size=obs_shape, dtype=np.uint8)
        np.random.randint(n_act, dtype=np.uint8)
    np.random.randint(low=0, high=255, size=obs_shape, dtype=np.uint8)
    np.random.rand()                                    # policy.get_action -> weighted_sample
    return smp, TablePolicyPort(g["prob_table"], g["value_table"]), n_batches


@pytest.mark.parametrize("tag", ["breakout", "pong_maxlen", "seaquest_nomid", "breakout_noop0"])
def test_sampler_port_matches_reference_sampler(tag):
    g = load_golden("g7_rollout_" + tag)
    smp, policy, n_batches = replay_rollout(g)
    mbr = bool(g["cfg"][5])
    traj = []
    for b in range(n_batches):
        buf, completed = smp.obtain_samples(policy)
        np.testing.assert_array_equal(buf["actions"], g["actions"][b], err_msg="b%d" % b)
        np.testing.assert_array_equal(buf["prob"], g["prob"][b])
        np.testing.assert_array_equal(buf["value"], g["value"][b])
        if mbr:
            np.testing.assert_array_equal(buf["rewards"], g["rewards"][b], err_msg="b%d" % b)
            np.testing.assert_array_equal(buf["dones"], g["dones"][b])
            np.testing.assert_array_equal(buf["raw_reward"], g["raw_reward"][b])
            np.testing.assert_array_equal(buf["need_reset"], g["need_reset"][b])
            np.testing.assert_array_equal(_crc_rows(buf["observations"]), g["obs_crc"][b])
        else:
            # NonResetCollector leaves stale rows after the first reset condition
            # (SURVEY.md appendix A.4): compare under the valids mask
            t = smp.horizon
            valid = P.valid_mask(g["need_reset"][b].reshape(-1, t)).reshape(-1).astype(bool)
            valid_here = P.valid_mask(buf["need_reset"].reshape(-1, t)).reshape(-1).astype(bool)
            np.testing.assert_array_equal(valid_here, valid)
            for key in ("rewards", "dones", "raw_reward", "need_reset"):
                np.testing.assert_array_equal(buf[key][valid], g[key][b][valid], err_msg=key)
            np.testing.assert_array_equal(_crc_rows(buf["observations"])[valid],
                                          g["obs_crc"][b][valid])
        np.testing.assert_array_equal(_crc_rows(buf["extra_observations"]), g["extra_crc"][b])
        if b == 0:
            np.testing.assert_array_equal(buf["observations"][:, -1],
                                          g["first_batch_newest_frames"])
        for ti in completed:
            traj.append((b,) + ti.as_tuple())
    want = sorted((int(b),) + tuple(float(x) for x in row)
                  for b, row in zip(g["traj_batch"], g["traj"]))
    got = sorted((b,) + tuple(float(x) for x in row) for (b, *row) in traj)
    assert len(want) > 0
    assert got == want


# -- G13: evaluation sampler ---------------------------------------------------

@pytest.mark.parametrize("tag", ["breakout", "pong_nomid"])
def test_eval_sampler_port_matches_reference(tag):
    """Training batches interleaved with evaluate_policy calls, recorded from the reference's own
    AAOEvalSampler: the evaluation trajectories AND every later training array must match (the
    eval resets and eval action draws advance the worker / master RNG streams)."""
    g = load_golden("g13_eval_" + tag)
    smp, policy, n_batches = replay_rollout(g)
    assert smp.eval_horizon == int(g["cfg"][9])
    mbr, t = bool(g["cfg"][5]), smp.horizon
    eval_at = set(int(x) for x in g["eval_batches"])
    traj, eval_traj = [], []
    for b in range(n_batches):
        if b in eval_at:
            eval_traj += [(b,) + ti.as_tuple() for ti in smp.evaluate_policy(policy)]
        buf, completed = smp.obtain_samples(policy)
        np.testing.assert_array_equal(buf["actions"], g["actions"][b], err_msg="b%d" % b)
        np.testing.assert_array_equal(buf["prob"], g["prob"][b])
        valid = np.ones(len(buf["rewards"]), bool)
        if not mbr:
            valid = P.valid_mask(g["need_reset"][b].reshape(-1, t)).reshape(-1).astype(bool)
        for key in ("rewards", "dones", "raw_reward", "need_reset"):
            np.testing.assert_array_equal(buf[key][valid], g[key][b][valid], err_msg="b%d %s" % (b, key))
        np.testing.assert_array_equal(_crc_rows(buf["observations"])[valid], g["obs_crc"][b][valid])
        np.testing.assert_array_equal(_crc_rows(buf["extra_observations"]), g["extra_crc"][b])
        traj += [(b,) + ti.as_tuple() for ti in completed]
    as_rows = lambda bs, rows: sorted((int(b),) + tuple(float(x) for x in r) for b, r in zip(bs, rows))     # noqa: E731
    assert sorted((b,) + tuple(float(x) for x in r) for (b, *r) in traj) == as_rows(g["traj_batch"], g["traj"])
    got = sorted((b,) + tuple(float(x) for x in r) for (b, *r) in eval_traj)
    assert got == as_rows(g["eval_at"], g["eval_traj"]) and len(got) >= 10


# -- G8 / G9 ------------------------------------------------------------------

def test_minibatch_indices():
    g = load_golden("g8_mbidx")
    for c in range(int(g["n_cases"])):
        bs, n, seed = [int(x) for x in g["c%d_cfg" % c]]
        np.random.seed(seed)
        for ep in range(3):
            mbs = P.minibatch_indices(bs, n, True)
            want = g["c%d_idx" % c][ep]
            assert len(mbs) == len(want) == max((n - bs) // bs + 1, 0) if n >= bs else len(mbs) == 0
            for a, b in zip(mbs, want):
                np.testing.assert_array_equal(a, b)
        ns = P.minibatch_indices(bs, n, False)
        np.testing.assert_array_equal(
            np.array([[m[0], m[-1] + 1] for m in ns], np.int64).reshape(-1, 2),
            g["c%d_noshuffle" % c])


def test_n_itr_table():
    for n_steps, sample_size, log_steps, n_itr, log_itrs in load_golden("g9_nitr")["table"]:
        assert P.n_itr_for(int(n_steps), int(sample_size), int(log_steps)) == (n_itr, log_itrs)


# -- G14: recurrent policy through the sampler ------------------------------------

class RecurrentTablePort(object):
    """Host twin of the fixture's recurrent stand-in policy, with the reference's pair-of-states
    handling for the alternating sampler (policies/base.py:32-93)."""
    recurrent = True

    def __init__(self, prob_table, value_table):
        self.prob_table, self.value_table = prob_table, value_table
        self._pair, self._j = None, 0

    def reset(self, n_batch):
        self._pair = [np.zeros((n_batch, 2), np.float32), np.zeros((n_batch, 2), np.float32)]
        self._j = 0

    def reset_one(self, idx):
        self._pair[self._j][idx] = 0

    def get_actions(self, obs):
        h = self._pair[self._j]
        key = obs.reshape(obs.shape[0], -1).astype(np.int64).sum(axis=1) % 64
        idx = (key + np.floor(4 * h[:, 0]).astype(np.int64)) % 64
        new_h = (np.float32(0.5) * h + np.stack([key.astype(np.float32) / np.float32(64),
                                                 np.ones(len(key), np.float32)], axis=1)).astype(np.float32)
        prob, value = self.prob_table[idx], self.value_table[idx]
        acts = P.sample_actions(prob, np.random.rand(len(key)))
        self._pair[self._j] = new_h
        self._j ^= 1
        return acts, dict(prob=prob, value=value, hprev_0=h)

    def state(self):
        return np.concatenate(self._pair)


def test_recurrent_policy_plumbing_matches_reference_sampler():
    """G14 (real multi-process sampler + a recurrent stand-in policy): which previous hidden state is
    stored at which (env, step) and when reset_one is applied, mid_batch_reset False."""
    g = load_golden("g14_recurrent_seaquest")
    smp, _, n_batches = replay_rollout(g)
    policy = RecurrentTablePort(g["prob_table"], g["value_table"])
    policy.reset(smp.half)
    t = smp.horizon
    for b in range(n_batches):
        buf, _ = smp.obtain_samples(policy)
        valid = P.valid_mask(g["need_reset"][b].reshape(-1, t)).reshape(-1).astype(bool)
        np.testing.assert_array_equal(buf["actions"], g["actions"][b], err_msg="b%d" % b)
        np.testing.assert_array_equal(buf["prob"], g["prob"][b])
        np.testing.assert_array_equal(buf["hprev_0"], g["hprev"][b], err_msg="b%d" % b)
        np.testing.assert_array_equal(buf["rewards"][valid], g["rewards"][b][valid])
        np.testing.assert_array_equal(_crc_rows(buf["observations"])[valid], g["obs_crc"][b][valid])
        np.testing.assert_array_equal(_crc_rows(buf["extra_observations"]), g["extra_crc"][b])
        np.testing.assert_array_equal(policy.state(), g["state_after"][b])
    assert (g["hprev"][:, :, 0].reshape(n_batches, -1, t)[:, :, 1:] == 0).any()       # resets really happened mid-segment
