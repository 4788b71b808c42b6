"""Sampler interface (reference: accel_rl/sampler/base.py:11-51)."""
import numpy as np

from accel_rl_amd.util.quick_args import save_args


class Sampler(object):

    def initialize(self, **kwargs):
        raise NotImplementedError

    def policy_init(self, policy):
        raise NotImplementedError

    def obtain_samples(self, itr):
        raise NotImplementedError

    def shutdown(self):
        raise NotImplementedError

    @property
    def alternating(self):
        return False


class BaseMbSampler(Sampler):
    """Constructor arguments as in the reference (sampler/base.py:32-47)."""

    def __init__(self, EnvCls, env_args, horizon, n_parallel=1, envs_per=1,
                 max_path_length=np.inf, mid_batch_reset=True,
                 max_decorrelation_steps=2000, profile_pathname=None):
        save_args(vars(), underscore=False)

    @property
    def total_n_envs(self):
        return self._total_n_envs
