#!/usr/bin/env python
"""
G14: the reference's real multi-process sampler driving a RECURRENT policy (build container only).

The policy is a small deterministic stand-in built on the reference's own BaseRecurrentPolicy
state handling (accel_rl/policies/base.py:32-93: pair of states for the alternating sampler,
reset / reset_one / advance_hiddens): hidden h (2 numbers per env) evolves as
h' = 0.5 h + [key / 64, 1] with key = pixel sum mod 64; prob / value come from tables indexed by
(key + floor(4 h[0])) mod 64.  What the fixture pins is the sampler's recurrent plumbing: which
previous state is stored at which (env, step), and when reset_one is applied -- with
mid_batch_reset False (the only mode the reference's recurrent algorithms allow).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as G  # noqa: E402

from accel_rl.sampler.act_server.alternating.overlap.sampler import ActsrvAltOvrlpSampler  # noqa: E402
from accel_rl.envs.atari_env import AtariEnv  # noqa: E402
from accel_rl.policies.base import BaseRecurrentPolicy  # noqa: E402

F32 = np.float32


class _Param(object):
    def __init__(self, v):
        self._v = v

    def get_value(self):
        return self._v


class RecurrentTablePolicy(BaseRecurrentPolicy):
    recurrent = True

    def __init__(self, space, table_seed=78):
        rs = np.random.RandomState(table_seed)
        logits = rs.randn(64, space.n) * 1.5
        p = np.exp(logits - logits.max(1, keepdims=True))
        self.prob_table = (p / p.sum(1, keepdims=True)).astype(F32)
        self.value_table = (rs.randn(64) * 2).astype(F32)
        self.space = space
        self.alternating_sampler = True
        self._network = type("N", (), dict(hid_init_params=[_Param(np.zeros(2, F32))]))()
        self.hid_init_params = self._network.hid_init_params
        self._prev_hiddens = None
        self._prev_hiddens_pair = None
        self._j = 0

    state_info_keys = ["hprev_0"]

    def _forward(self, obs, h):
        obs = np.asarray(obs)
        key = obs.reshape(obs.shape[0], -1).astype(np.int64).sum(axis=1) % 64
        idx = (key + np.floor(4 * h[:, 0]).astype(np.int64)) % 64
        new_h = (F32(0.5) * h + np.stack([key.astype(F32) / F32(64), np.ones(len(key), F32)], axis=1)).astype(F32)
        return self.prob_table[idx], self.value_table[idx], new_h

    def get_action(self, ob):
        prev = self.get_prev_hiddens()
        prob, value, new_h = self._forward(ob[None], prev[0])
        info = dict(prob=prob[0], value=value[0], hprev_0=prev[0][0])
        self.advance_hiddens([new_h])
        return self.space.weighted_sample(prob[0]), info

    def get_actions(self, obs):
        prev = self.get_prev_hiddens()
        prob, value, new_h = self._forward(obs, prev[0])
        acts = self.space.weighted_sample_n(prob)
        info = dict(prob=prob, value=value, hprev_0=prev[0])
        self.advance_hiddens([new_h])
        return acts, info


def run(tag, game, n_parallel, envs_per, horizon, n_batches, seed, max_path_length, env_args=None):
    env_args = dict(env_args or {})
    env_args["game"] = game
    sampler = ActsrvAltOvrlpSampler(EnvCls=AtariEnv, env_args=env_args, horizon=horizon, n_parallel=n_parallel,
                                    envs_per=envs_per, mid_batch_reset=False, max_path_length=max_path_length,
                                    max_decorrelation_steps=0)
    np.random.seed(seed)
    env_spec, sample_size, hor, mbr = sampler.initialize(
        seed=seed + 1, affinities=dict(sim_cpus=list(range(2 * n_parallel))), discount=0.99, need_extra_obs=True)
    policy = RecurrentTablePolicy(env_spec.action_space)
    sampler.policy_init(policy)
    rec = dict((k, []) for k in ("rewards", "dones", "need_reset", "actions", "prob", "value", "hprev", "obs_crc",
                                 "extra_crc", "state_after"))
    for b in range(n_batches):
        buf, _ = sampler.obtain_samples(b)
        rec["rewards"].append(buf.rewards.copy())
        rec["dones"].append(buf.dones.copy())
        rec["need_reset"].append(buf.env_infos.need_reset.copy())
        rec["actions"].append(buf.actions.copy())
        rec["prob"].append(buf.agent_infos["prob"].copy())
        rec["value"].append(buf.agent_infos["value"].copy())
        rec["hprev"].append(buf.agent_infos["hprev_0"].copy())
        rec["obs_crc"].append(G.crc_rows(buf.observations))
        rec["extra_crc"].append(G.crc_rows(buf.extra_observations))
        rec["state_after"].append(policy.get_state_info()[0].copy())
    sampler.shutdown()
    out = dict(cfg=np.array([n_parallel, envs_per, horizon, n_batches, seed, 0, int(max_path_length)], np.int64),
               game=np.array(game), discount=np.array(0.99),
               env_args=np.array(repr(sorted((k, v) for k, v in env_args.items() if k != "game"))),
               prob_table=policy.prob_table, value_table=policy.value_table)
    for k in rec:
        out[k] = np.stack(rec[k])
    G.save("g14_recurrent_" + tag, **out)


if __name__ == "__main__":
    run("seaquest", "seaquest", 2, 2, 5, 24, seed=51, max_path_length=13, env_args=dict(max_start_noops=7))
