"""A2C with a recurrent policy at the example's scale: env-steps/s of rollout + BPTT update.
usage: python tools/recurrent_bench.py [lstm|gru|rnn] [n_envs] [steps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch

from accel_rl_amd.algos.pg.a2c import A2C
from accel_rl_amd.envs.synthetic_atari import SynthAtariEnv
from accel_rl_amd.policies.atari_cnn_specs import cnn_specs
from accel_rl_amd.runners.accel_rl import AccelRL
from accel_rl_amd.sampler.gpu_sampler import GpuVecSampler
from accel_rl_amd.util import logger


def main():
    kind = sys.argv[1] if len(sys.argv) > 1 else "lstm"
    n_envs = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 100
    if kind == "lstm":
        from accel_rl_amd.policies.atari_lstm_policy import AtariLstmPolicy as Policy
    elif kind == "gru":
        from accel_rl_amd.policies.atari_gru_policy import AtariGruPolicy as Policy
    else:
        from accel_rl_amd.policies.atari_rnn_policy import AtariRnnPolicy as Policy
    logger.set_quiet(True)
    horizon = 5
    sampler = GpuVecSampler(EnvCls=SynthAtariEnv, env_args=dict(game="breakout"), horizon=horizon, n_parallel=16,
                            envs_per=n_envs // 32, max_path_length=int(27e3), mid_batch_reset=False,
                            max_decorrelation_steps=200, device="cuda:0")
    policy = Policy(**cnn_specs[0])
    algo = A2C(discount=0.99, gae_lambda=1)
    runner = AccelRL(algo=algo, policy=policy, sampler=sampler, n_steps=1e9, seed=0, affinities=dict(gpu=0),
                     log_interval_steps=1e8)
    runner.startup()
    for itr in range(10):
        samples, _ = sampler.obtain_samples(itr)
        algo.optimize_policy(itr, samples)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for itr in range(10, 10 + steps):
        samples, _ = sampler.obtain_samples(itr)
        algo.optimize_policy(itr, samples)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("%s A2C, %d envs x horizon %d, spec-0 CNN: %.0f env-steps/s, %.3f ms/step"
          % (kind, n_envs, horizon, steps * n_envs * horizon / dt, dt / steps * 1e3))


if __name__ == "__main__":
    main()
