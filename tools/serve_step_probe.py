"""Rollout time of GpuVecSampler with the steps served in ONE launch each (arl_env_step_served) against the separate
launches, by number of environments: where does one 16-wave workgroup per env stop paying?
usage: python tools/serve_step_probe.py [--spec K] [n_env ...]      (spec-K CNN (default 1), breakout, horizon 5; hipGraph
replays; spec 0's first layer (16 filters) stays a launch of its own)"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from accel_rl_amd.envs.synthetic_atari import SynthAtariEnv          # noqa: E402
from accel_rl_amd.policies.atari_cnn_policy import AtariCnnPolicy    # noqa: E402
from accel_rl_amd.policies.atari_cnn_specs import cnn_specs          # noqa: E402
from accel_rl_amd.sampler.gpu_sampler import GpuVecSampler           # noqa: E402
from accel_rl_amd.util import logger                                 # noqa: E402

DEV = "cuda:0"


SPEC = 1


def rollout_ms(n_env, served, replays=60):
    class Sampler(GpuVecSampler):
        _serve_in_step = served
        _serve_in_step_max_envs = (1 << 30, 1 << 30)

    smp = Sampler(EnvCls=SynthAtariEnv, env_args=dict(game="breakout"), horizon=5, n_parallel=n_env // 16, envs_per=8,
                  mid_batch_reset=True, max_decorrelation_steps=0, device=DEV, use_graph=True)
    np.random.seed(1)
    env_spec, *_ = smp.initialize(seed=2, affinities=dict(), discount=0.99, need_extra_obs=True)
    policy = AtariCnnPolicy(**cnn_specs[SPEC])
    policy.initialize(env_spec, device=DEV)
    smp.policy_init(policy)
    assert smp._serve_fused == served
    for i in range(5):
        smp.obtain_samples(i)
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        for i in range(replays):
            smp.obtain_samples(i)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / replays * 1e3)
    smp.shutdown()
    return best


if __name__ == "__main__":
    logger.set_quiet(True)
    argv = sys.argv[1:]
    if argv[:1] == ["--pmc"]:          # a few replays of both forms, for counter passes (tools/serve_pmc.sh)
        for served in (False, True):
            rollout_ms(int(argv[1]), served, replays=4)
        sys.exit(0)
    if argv[:1] == ["--spec"]:
        SPEC, argv = int(argv[1]), argv[2:]
    sizes = [int(a) for a in argv] or [128, 256, 384, 512, 768, 1024]
    print("spec %d" % SPEC)
    print("envs   separate launches   one launch per step   (ms per 5-step rollout, best of 3 x 60 graph replays)")
    for n in sizes:
        a, b = rollout_ms(n, False), rollout_ms(n, True)
        print("%5d   %8.4f            %8.4f              %+.1f %%" % (n, a, b, (b / a - 1) * 100))
