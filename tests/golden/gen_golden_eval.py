#!/usr/bin/env python
"""
Golden vectors for SURVEY 8(f2): the reference's evaluation sampler.  Runs ONLY in the build
container (needs /root/reference); writes tests/golden/g13_eval_*.npz.

Recorded from the reference's OWN AAOEvalSampler (real worker processes,
accel_rl/sampler/act_server/alternating/overlap/{sampler_with_eval,worker_with_eval}.py) over
AtariEnv + the synthetic emulator and the table policy of gen_golden.py: training batches
interleaved with evaluate_policy calls -- every training array of every batch (so the
RNG-stream interplay of eval resets / eval action draws with training is pinned) and the
completed evaluation trajectories (a multiset: queue order across workers is arrival order).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as G  # noqa: E402  (installs the shims, defines TablePolicy / crc_rows / save)

from accel_rl.sampler.act_server.alternating.overlap.sampler_with_eval import AAOEvalSampler  # noqa: E402
from accel_rl.envs.atari_env import AtariEnv  # noqa: E402


def run(tag, game, n_parallel, envs_per, horizon, n_batches, seed, eval_steps, eval_envs_per, eval_at,
        mid_batch_reset=True, max_path_length=np.inf, env_args=None, discount=0.99):
    env_args = dict(env_args or {})
    env_args["game"] = game
    sampler = AAOEvalSampler(eval_steps=eval_steps, eval_envs_per=eval_envs_per,
                             EnvCls=AtariEnv, env_args=env_args, horizon=horizon, n_parallel=n_parallel,
                             envs_per=envs_per, mid_batch_reset=mid_batch_reset,
                             max_path_length=max_path_length, max_decorrelation_steps=0)
    np.random.seed(seed)
    env_spec, sample_size, hor, mbr = sampler.initialize(
        seed=seed + 1, affinities=dict(sim_cpus=list(range(2 * n_parallel))), discount=discount,
        need_extra_obs=True)
    policy = G.TablePolicy(env_spec.action_space)
    sampler.policy_init(policy)
    rec = dict((k, []) for k in ("rewards", "dones", "raw_reward", "need_reset", "actions", "prob", "value",
                                 "obs_crc", "extra_crc", "traj", "traj_batch", "eval_traj", "eval_at"))
    for b in range(n_batches):
        if b in eval_at:
            for ti in sampler.evaluate_policy(b):
                rec["eval_traj"].append([ti["Length"], ti["Return"], ti["RawReturn"], ti["NonzeroRewards"],
                                         ti["DiscountedReturn"]])
                rec["eval_at"].append(b)
        buf, traj_infos = sampler.obtain_samples(b)
        rec["rewards"].append(buf.rewards.copy())
        rec["dones"].append(buf.dones.copy())
        rec["raw_reward"].append(buf.env_infos.raw_reward.copy())
        rec["need_reset"].append(buf.env_infos.need_reset.copy())
        rec["actions"].append(buf.actions.copy())
        rec["prob"].append(buf.agent_infos["prob"].copy())
        rec["value"].append(buf.agent_infos["value"].copy())
        rec["obs_crc"].append(G.crc_rows(buf.observations))
        rec["extra_crc"].append(G.crc_rows(buf.extra_observations))
        for ti in traj_infos:
            rec["traj"].append([ti["Length"], ti["Return"], ti["RawReturn"], ti["NonzeroRewards"],
                                ti["DiscountedReturn"]])
            rec["traj_batch"].append(b)
    sampler.shutdown()
    out = dict(
        cfg=np.array([n_parallel, envs_per, horizon, n_batches, seed, int(mid_batch_reset),
                      -1 if np.isinf(max_path_length) else int(max_path_length), eval_steps, eval_envs_per,
                      sampler.eval_horizon], np.int64),
        eval_batches=np.array(sorted(eval_at), np.int64),
        game=np.array(game), discount=np.array(discount),
        env_args=np.array(repr(sorted((k, v) for k, v in env_args.items() if k != "game"))),
        prob_table=policy.prob_table, value_table=policy.value_table,
        traj=np.array(rec["traj"], np.float64).reshape(-1, 5), traj_batch=np.array(rec["traj_batch"], np.int64),
        eval_traj=np.array(rec["eval_traj"], np.float64).reshape(-1, 5),
        eval_at=np.array(rec["eval_at"], np.int64))
    for k in ("rewards", "dones", "raw_reward", "need_reset", "actions", "prob", "value", "obs_crc", "extra_crc"):
        out[k] = np.stack(rec[k])
    print(tag, "eval trajectories:", len(rec["eval_traj"]), "training trajectories:", len(rec["traj"]))
    G.save("g13_eval_" + tag, **out)


if __name__ == "__main__":
    # breakout, 8 training envs + 4 eval envs, evaluations before batches 0, 7 and 15; short episodes via max_path_length
    run("breakout", "breakout", 2, 2, 5, 22, seed=41, eval_steps=240, eval_envs_per=1, eval_at={0, 7, 15},
        max_path_length=25)
    # NonResetCollector variant, 2 eval envs per worker
    run("pong_nomid", "pong", 1, 2, 4, 18, seed=43, eval_steps=200, eval_envs_per=2, eval_at={3, 11},
        mid_batch_reset=False, max_path_length=17, env_args=dict(max_start_noops=5))
