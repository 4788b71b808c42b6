#!/usr/bin/env python
"""
Golden-vector generator.  Runs ONLY in the build container (needs
/root/reference); the .npz files it writes are committed and are what travels.

    python tests/golden/gen_golden.py            # regenerate every fixture

Every array below is an input or an output of the reference's OWN code
(imported from /root/reference via tests/golden/ref_shims.py); nothing is
computed by this repo's oracle or product.  Fixtures hold plain numpy arrays
only -- no reference source, bytecode or pickled reference objects.

G1 gae_*          accel_rl.algos.pg.util.gen_adv_est
G2 nstep_*        accel_rl.algos.pg.util.discount_returns (+ adv = ret - v)
G3 valids         accel_rl.algos.pg.util.update_valids / zero_after_reset
G4 process_*      accel_rl.algos.pg.aac_base.AdvActorCriticBase.process_samples
G5 sample_*       rllab.misc.special.weighted_sample_n (via accel_rl Discrete)
G6 env_*          accel_rl.envs.atari_env.AtariEnv over oracle.synth_ale.SynthALE
G7 rollout_*      accel_rl.sampler...overlap.sampler.ActsrvAltOvrlpSampler
                  (the real multi-process sampler) + AtariEnv + table policy
G8 mbidx          accel_rl.optimizers.util.iterate_mb_idxs
G9 nitr           accel_rl.runners.accel_rl_base.AccelRLBase.get_n_itr
"""

import os
import sys
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shims  # noqa: E402

ref_shims.install()

from accel_rl.algos.pg import util as ref_pg_util  # noqa: E402
from accel_rl.algos.pg.aac_base import AdvActorCriticBase  # noqa: E402
from accel_rl.buffers.batch import buffer_with_segs_view, batch_buffer  # noqa: E402
from accel_rl.spaces.discrete import Discrete  # noqa: E402
from accel_rl.envs.atari_env import AtariEnv  # noqa: E402
from accel_rl.optimizers.util import iterate_mb_idxs  # noqa: E402
from accel_rl.runners.accel_rl_base import AccelRLBase  # noqa: E402
from accel_rl.sampler.act_server.alternating.overlap.sampler import \
    ActsrvAltOvrlpSampler  # noqa: E402
from accel_rl.util.misc import struct  # noqa: E402

F32 = np.float32


def crc_rows(a):
    a = np.ascontiguousarray(a)
    return np.array([zlib.crc32(a[i].tobytes()) for i in range(len(a))], np.uint32)


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print("wrote %-28s %8.1f KB" % (name + ".npz", os.path.getsize(path) / 1024.))


# -----------------------------------------------------------------------------
# shared random scan inputs
# -----------------------------------------------------------------------------

SCAN_SHAPES = [(16, 5), (64, 5), (8, 128), (7, 1), (33, 32), (5, 17),
               (256, 5), (64, 128)]     # SURVEY 8c's list: config 2's batch and a long horizon (appended: cases 0-17 keep their draws)
SCAN_PARAMS = [(0.99, 0.95), (0.99, 1.0), (1.0, 1.0), (0.9, 0.0)]
DONE_PROBS = [0.0, 0.1, 1.0]


def scan_inputs(rs, n, t, p_done):
    r = rs.choice([-1., 0., 0., 0., 1.], size=(n, t)).astype(F32)
    r += (rs.rand(n, t) < 0.2) * rs.randn(n, t).astype(F32)      # some non-integer rewards
    r = r.astype(F32)
    v = (rs.randn(n, t) * 3).astype(F32)
    d = rs.rand(n, t) < p_done
    lv = (rs.randn(n) * 3).astype(F32)
    return r, v, d, lv


def g1_g2():
    rs = np.random.RandomState(11)
    out = dict()
    case = 0
    for (n, t) in SCAN_SHAPES:
        for p_done in DONE_PROBS:
            r, v, d, lv = scan_inputs(rs, n, t, p_done)
            ikey = "i%02d" % case
            case += 1
            out[ikey + "_r"], out[ikey + "_v"], out[ikey + "_d"], out[ikey + "_lv"] = r, v, d, lv
            for pi, (gam, lam) in enumerate(SCAN_PARAMS):
                key = "%s_p%d" % (ikey, pi)
                # --- gen_adv_est, as run under this container's numpy (NEP 50)
                adv = np.zeros((n, t), F32)
                ret = np.zeros((n, t), F32)
                for e in range(n):
                    ref_pg_util.gen_adv_est(r[e], v[e], d[e], lv[e], gam, lam,
                                            adv_dest=adv[e], ret_dest=ret[e])
                out[key + "_adv"], out[key + "_ret"] = adv, ret
                # --- numpy-1.x promotion emulated by float64 inputs, f32 dests
                adv64 = np.zeros((n, t), F32)
                ret64 = np.zeros((n, t), F32)
                for e in range(n):
                    ref_pg_util.gen_adv_est(r[e].astype(np.float64), v[e].astype(np.float64),
                                            d[e], np.float64(lv[e]), gam, lam,
                                            adv_dest=adv64[e], ret_dest=ret64[e])
                out[key + "_adv_legacy"] = adv64
                if pi == 0:
                    continue            # n-step returns do not depend on lambda (same gamma as p1)
                # --- discount_returns (+ adv = ret - v, aac_base.py:121)
                nret = np.zeros((n, t), F32)
                for e in range(n):
                    ref_pg_util.discount_returns(r[e], d[e], lv[e], gam, ret_dest=nret[e])
                out[key + "_nret"] = nret
                out[key + "_nadv"] = nret - v
                nret64 = np.zeros((n, t), F32)
                for e in range(n):
                    ref_pg_util.discount_returns(r[e].astype(np.float64), d[e],
                                                 np.float64(lv[e]), gam, ret_dest=nret64[e])
                out[key + "_nret_legacy"] = nret64
    out["n_inputs"] = np.array(case)
    out["params"] = np.array(SCAN_PARAMS, np.float64)
    save("g1_g2_scans", **out)


def g3():
    rs = np.random.RandomState(12)
    out = dict()
    case = 0
    for (n, t) in [(16, 5), (9, 1), (32, 20)]:
        for p in [0.0, 0.15, 0.6, 1.0]:
            for with_key in (True, False):
                need = rs.rand(n, t) < p
                dones = need | (rs.rand(n, t) < 0.2)
                adv = rs.randn(n, t).astype(F32)
                ret = rs.randn(n, t).astype(F32)
                val = rs.randn(n, t).astype(F32)
                valids = np.full((n, t), 7, np.int8)
                a2, r2, v2 = adv.copy(), ret.copy(), val.copy()
                for e in range(n):
                    env_infos = dict(need_reset=need[e]) if with_key else dict()
                    path = dict(env_infos=env_infos, dones=dones[e])
                    ref_pg_util.update_valids(path, valids[e])
                    ref_pg_util.zero_after_reset(a2[e], r2[e], v2[e], path)
                key = "c%02d" % case
                case += 1
                out[key + "_flags"] = need if with_key else dones
                out[key + "_adv"], out[key + "_ret"], out[key + "_val"] = adv, ret, val
                out[key + "_valids"] = valids
                out[key + "_adv_z"], out[key + "_ret_z"], out[key + "_val_z"] = a2, r2, v2
    out["n_cases"] = np.array(case)
    save("g3_valids", **out)


class _ValuePolicy(object):
    def __init__(self, last_values):
        self._lv = last_values

    def value(self, observations):
        return self._lv


def g4():
    rs = np.random.RandomState(13)
    out = dict()
    case = 0
    for (n, t) in [(16, 5), (256, 5), (8, 20)]:
        for (gam, lam) in [(0.99, 0.95), (0.99, 1)]:
            for use_valids in (False, True):
                for std_adv in (False, True):
                    r, v, d, lv = scan_inputs(rs, n, t, 0.1)
                    need = d & (rs.rand(n, t) < 0.5)
                    examples = dict(
                        rewards=F32(0), dones=False,
                        env_infos=dict(need_reset=False, raw_reward=F32(0)),
                        agent_infos=dict(value=F32(0)),
                    )
                    samples = buffer_with_segs_view(examples, n * t, t, shared=False)
                    samples.rewards[:] = r.reshape(-1)
                    samples.dones[:] = d.reshape(-1)
                    samples.env_infos.need_reset[:] = need.reshape(-1)
                    samples.agent_infos.value[:] = v.reshape(-1)
                    samples.extra_observations = np.zeros((n, 1), np.uint8)
                    algo = AdvActorCriticBase(discount=gam, gae_lambda=lam,
                                              standardize_adv=std_adv)
                    algo.policy = _ValuePolicy(lv)
                    algo._use_valids = use_valids
                    opt_ex = dict(advantages=F32(1), returns=F32(1))
                    if use_valids:
                        opt_ex["valids"] = np.int8(1)
                    algo._opt_buf = buffer_with_segs_view(opt_ex, n * t, t, shared=False)
                    opt = algo.process_samples(0, samples)
                    key = "c%02d" % case
                    case += 1
                    out[key + "_r"], out[key + "_v"], out[key + "_d"], out[key + "_lv"] = r, v, d, lv
                    out[key + "_need"] = need
                    out[key + "_cfg"] = np.array([gam, lam, use_valids, std_adv], np.float64)
                    out[key + "_adv"] = opt.advantages.reshape(n, t).copy()
                    out[key + "_ret"] = opt.returns.reshape(n, t).copy()
                    if use_valids:
                        out[key + "_valids"] = opt.valids.reshape(n, t).copy()
                        out[key + "_value_after"] = \
                            samples.agent_infos.value.reshape(n, t).copy()
    out["n_cases"] = np.array(case)
    save("g4_process_samples", **out)


def g5():
    out = dict()
    case = 0
    rs = np.random.RandomState(14)
    for a in (4, 6, 18, 2, 9):
        for b in (1, 128, 257):
            logits = rs.randn(b, a) * 2
            p = np.exp(logits - logits.max(1, keepdims=True))
            p = (p / p.sum(1, keepdims=True)).astype(F32)
            if b > 4:
                p[0] = 0
                p[0, a - 1] = 1                 # all mass on the last action
                p[1] = 0
                p[1, 0] = 1                     # all mass on the first
                p[2] = F32(1. / a)              # uniform
                p[3] *= F32(0.5)                # row summing to 0.5 -> clamp to A-1 often
            space = Discrete(a)
            seed = 100 + case
            np.random.seed(seed)
            acts = space.weighted_sample_n(p)
            np.random.seed(seed)
            u = np.random.rand(b)               # the uniforms weighted_sample_n consumed
            key = "c%02d" % case
            case += 1
            out[key + "_prob"], out[key + "_u"], out[key + "_act"] = p, u, acts
            out[key + "_seed"] = np.array(seed)
    out["n_cases"] = np.array(case)
    save("g5_sampling", **out)


def g6():
    """Scripted single-env runs of the real AtariEnv (over SynthALE)."""
    out = dict()
    for gi, (game, n_steps, kwargs) in enumerate([
            ("breakout", 700, dict()),
            ("pong", 400, dict()),
            ("seaquest", 600, dict(max_start_noops=5)),
            ("breakout", 400, dict(episodic_lives=False, clip_reward=False, num_img_obs=1)),
    ]):
        np.random.seed(200 + gi)
        env = AtariEnv(game=game, **kwargs)
        rs = np.random.RandomState(300 + gi)
        acts = rs.randint(0, env.action_space.n, size=n_steps).astype(np.uint8)
        obs0 = env.reset()
        rew, done, raw, need, ticks, crcs, resets = [], [], [], [], [], [], []
        keep_obs, keep_idx = [obs0], [-1]
        for i in range(n_steps):
            o, r, d, info = env.step(acts[i])
            rew.append(r)
            done.append(d)
            raw.append(info.get("raw_reward", r))
            need.append(info.get("need_reset", d))
            did_reset = bool(d and info.get("need_reset", True))
            if did_reset:
                o = env.reset()
            resets.append(did_reset)
            ticks.append(env.ale.tick)
            crcs.append(zlib.crc32(o.tobytes()))
            if i < 3 or d or (i % 197 == 0):
                keep_obs.append(o)
                keep_idx.append(i)
        key = "g%d" % gi
        out[key + "_game"] = np.array(game)
        out[key + "_kwargs"] = np.array(repr(sorted(kwargs.items())))
        out[key + "_seed"] = np.array(200 + gi)
        out[key + "_acts"] = acts
        out[key + "_rew"] = np.array(rew, F32)
        out[key + "_done"] = np.array(done, bool)
        out[key + "_raw"] = np.array(raw, F32)
        out[key + "_need"] = np.array(need, bool)
        out[key + "_reset"] = np.array(resets, bool)
        out[key + "_tick"] = np.array(ticks, np.int64)
        out[key + "_crc"] = np.array(crcs, np.uint32)
        # keep only the newest frame of the kept observations (the rest of the
        # stack is covered by the CRC) to bound the fixture size
        out[key + "_keep_idx"] = np.array(keep_idx, np.int64)
        out[key + "_keep_last"] = np.stack([o[-1] for o in keep_obs])
        out[key + "_keep_nonzero"] = np.array(
            [[bool(f.any()) for f in o] for o in keep_obs], bool)
    out["n_games"] = np.array(4)
    save("g6_env", **out)


class TablePolicy(object):
    """Deterministic table policy used for the rollout fixtures: the key is the
    integer pixel sum of the observation mod 64.  Sampling goes through the
    reference's own Discrete.weighted_sample(_n) (global numpy RNG)."""
    recurrent = False

    def __init__(self, space, table_seed=77):
        rs = np.random.RandomState(table_seed)
        logits = rs.randn(64, space.n) * 1.5
        p = np.exp(logits - logits.max(1, keepdims=True))
        self.prob_table = (p / p.sum(1, keepdims=True)).astype(F32)
        self.value_table = (rs.randn(64) * 2).astype(F32)
        self.space = space

    def keys(self, obs):
        obs = np.asarray(obs)
        return obs.reshape(obs.shape[0], -1).astype(np.int64).sum(axis=1) % 64

    def reset(self, n_batch):
        pass

    def reset_one(self, idx):
        pass

    def get_action(self, ob):
        k = self.keys(ob[None])[0]
        prob, value = self.prob_table[k], self.value_table[k]
        return self.space.weighted_sample(prob), dict(prob=prob, value=value)

    def get_actions(self, obs):
        k = self.keys(obs)
        prob, value = self.prob_table[k], self.value_table[k]
        return self.space.weighted_sample_n(prob), dict(prob=prob, value=value)

    def value(self, obs):
        return self.value_table[self.keys(obs)]


def run_rollout(tag, game, n_parallel, envs_per, horizon, n_batches, seed,
                mid_batch_reset=True, max_path_length=np.inf, env_args=None,
                discount=0.99):
    env_args = dict(env_args or {})
    env_args["game"] = game
    sampler = ActsrvAltOvrlpSampler(
        EnvCls=AtariEnv, env_args=env_args, horizon=horizon,
        n_parallel=n_parallel, envs_per=envs_per, mid_batch_reset=mid_batch_reset,
        max_path_length=max_path_length, max_decorrelation_steps=0)
    np.random.seed(seed)                                   # runner: set_seed(seed)
    env_spec, sample_size, hor, mbr = sampler.initialize(
        seed=seed + 1, affinities=dict(sim_cpus=list(range(2 * n_parallel))),
        discount=discount, need_extra_obs=True)
    policy = TablePolicy(env_spec.action_space)
    sampler.policy_init(policy)
    n = 2 * n_parallel * envs_per
    rec = dict((k, []) for k in ("rewards", "dones", "raw_reward", "need_reset",
                                 "actions", "prob", "value", "obs_crc", "extra_crc",
                                 "traj", "traj_batch"))
    first_obs = None
    for b in range(n_batches):
        buf, traj_infos = sampler.obtain_samples(b)
        if b == 0:
            first_obs = buf.observations[:, -1].copy()     # newest frame of each row
        rec["rewards"].append(buf.rewards.copy())
        rec["dones"].append(buf.dones.copy())
        rec["raw_reward"].append(buf.env_infos.raw_reward.copy())
        rec["need_reset"].append(buf.env_infos.need_reset.copy())
        rec["actions"].append(buf.actions.copy())
        rec["prob"].append(buf.agent_infos["prob"].copy())
        rec["value"].append(buf.agent_infos["value"].copy())
        rec["obs_crc"].append(crc_rows(buf.observations))
        rec["extra_crc"].append(crc_rows(buf.extra_observations))
        for ti in traj_infos:
            rec["traj"].append([ti["Length"], ti["Return"], ti["RawReturn"],
                                ti["NonzeroRewards"], ti["DiscountedReturn"]])
            rec["traj_batch"].append(b)
    sampler.shutdown()
    out = dict(
        cfg=np.array([n_parallel, envs_per, horizon, n_batches, seed,
                      int(mid_batch_reset),
                      -1 if np.isinf(max_path_length) else int(max_path_length)], np.int64),
        game=np.array(game), discount=np.array(discount),
        env_args=np.array(repr(sorted((k, v) for k, v in env_args.items() if k != "game"))),
        prob_table=policy.prob_table, value_table=policy.value_table,
        first_batch_newest_frames=first_obs,
        traj=np.array(rec["traj"], np.float64).reshape(-1, 5),
        traj_batch=np.array(rec["traj_batch"], np.int64),
    )
    for k in ("rewards", "dones", "raw_reward", "need_reset", "actions", "prob",
              "value", "obs_crc", "extra_crc"):
        out[k] = np.stack(rec[k])
    assert out["rewards"].shape == (n_batches, n * horizon)
    save("g7_rollout_" + tag, **out)


def g7():
    # breakout, 8 envs, default noops, 80 batches: life losses + a game over per env
    run_rollout("breakout", "breakout", 2, 2, 5, 80, seed=5)
    # forced over-length resets (max_path_length) and a 6-action game with no lives
    run_rollout("pong_maxlen", "pong", 1, 3, 4, 30, seed=9, max_path_length=11)
    # mid_batch_reset=False (NonResetCollector): short episodes via max_path_length
    run_rollout("seaquest_nomid", "seaquest", 2, 1, 5, 30, seed=21,
                mid_batch_reset=False, max_path_length=13,
                env_args=dict(max_start_noops=7))
    # deterministic starts (max_start_noops=0), envs_per > 1
    run_rollout("breakout_noop0", "breakout", 1, 4, 5, 70, seed=33,
                env_args=dict(max_start_noops=0))


def g8():
    out = dict()
    for ci, (bs, n, seed) in enumerate([(512, 1280, 1), (64, 200, 2), (5, 5, 3), (8, 7, 4)]):
        np.random.seed(seed)
        epochs = []
        for _ in range(3):
            mbs = [b[0].copy() for b in iterate_mb_idxs(bs, n, shuffle=True)]
            epochs.append(np.stack(mbs) if mbs else np.zeros((0, bs), np.int64))
        out["c%d_cfg" % ci] = np.array([bs, n, seed])
        out["c%d_idx" % ci] = np.stack(epochs)
        out["c%d_noshuffle" % ci] = np.array(
            [list(b) for b in iterate_mb_idxs(bs, n, shuffle=False)], np.int64).reshape(-1, 2)
    out["n_cases"] = np.array(4)
    save("g8_mbidx", **out)


def g9():
    rows = []
    for n_steps in (1, 1000, 12345, 10 ** 6, 10 ** 7, 5 * 10 ** 7):
        for sample_size in (80, 1280, 5120, 10240):
            for log_steps in (1, 10 ** 4, 10 ** 5, 10 ** 6):
                r = AccelRLBase.__new__(AccelRLBase)
                r.n_steps = n_steps
                r._log_steps = log_steps
                n_itr = r.get_n_itr(sample_size)
                rows.append([n_steps, sample_size, log_steps, n_itr, r._log_interval_itrs])
    save("g9_nitr", table=np.array(rows, np.int64))


if __name__ == "__main__":
    which = sys.argv[1:] or ["g1", "g3", "g4", "g5", "g6", "g7", "g8", "g9"]
    for w in which:
        dict(g1=g1_g2, g3=g3, g4=g4, g5=g5, g6=g6, g7=g7, g8=g8, g9=g9)[w]()
