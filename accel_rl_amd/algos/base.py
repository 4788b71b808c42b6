"""Algorithm interface (reference: accel_rl/algos/base.py:3-13)."""


class RLAlgorithm(object):

    def initialize(self, policy, env_spec, sample_size, horizon, mid_batch_reset):
        raise NotImplementedError

    def optimize_policy(self, itr, samples_data):
        raise NotImplementedError

    @property
    def opt_info_keys(self):
        return []
