"""Time arl_env_step (the env side of one rollout step, one launch) inside a hipGraph of 40 chained launches, with the
stacked observation written twice (step_obs kept current) and once (policies that serve rows of the rollout buffer).
usage: python tools/env_step_probe.py [n_envs] [0 | 1: only that single_write mode]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__; __graft_entry__.build()
from accel_rl_amd import _lib
from accel_rl_amd.envs.synthetic_atari import SynthAtariEnv
from accel_rl_amd.sampler.gpu_sampler import GpuVecSampler
from accel_rl_amd.util import logger
logger.set_quiet(True)
dev = torch.device("cuda", 0)
n_env = int(sys.argv[1]) if len(sys.argv) > 1 else 256
smp = GpuVecSampler(EnvCls=SynthAtariEnv, env_args=dict(game="breakout"), horizon=5, n_parallel=16, envs_per=n_env // 32,
                    max_path_length=27000, max_decorrelation_steps=0, device=dev)
np.random.seed(0)
smp.initialize(seed=1, discount=0.99, need_extra_obs=True)
class P(object):
    recurrent = False; serves_rows = True
    def reset(self, n_batch): pass
    def get_action(self, ob): return None, None
    def prob_value(self, obs, rows=None): return self.p, self.v
pol = P(); pol.p = torch.full((n_env, 4), 0.25, device=dev); pol.v = torch.zeros(n_env, device=dev)
smp.policy_init(pol)
u = torch.rand(n_env, dtype=torch.float64, device=dev)
_lib.load().arl_dev_env_variant(int(os.environ.get("ARL_ENV_VARIANT", "0")))     # timing knock-outs (accel_rl_hip_dev.h)
for single in ((int(sys.argv[2]),) if len(sys.argv) > 2 else (0, 1)):
    def go():
        for s in range(40):
            _lib.env_step(smp._game, smp._state, smp._rollout, pol.p, pol.v, u, s % 4, True, 27000, 0.99, 30, single_write=single)
    go(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        go()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): g.replay()
    torch.cuda.synchronize()
    print("dbg n_env=%d single_write=%d variant=%s: %.2f us per launch" % (n_env, single, os.environ.get("ARL_ENV_VARIANT", "0"),
                                                                            (time.perf_counter() - t0) / 800 * 1e6))
