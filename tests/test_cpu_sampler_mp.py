"""The multi-process CPU baseline sampler (oracle/cpu_sampler_mp.py: master + 2*n_parallel
workers, alternating groups, semaphore hand-offs as in the reference) must reproduce the
sequential restatement CpuSamplerPort -- itself pinned to the real reference sampler by the
G7 fixtures -- bit for bit.  Runs on CPU."""
import numpy as np
import pytest

from oracle import ref_port as P
from oracle.cpu_sampler_mp import CpuSamplerMP


class TablePolicy(object):
    """Deterministic stand-in for the action server: actions / prob / value are functions of
    the observation bytes only."""

    def __init__(self, n_actions):
        self.n = n_actions

    def get_actions(self, obs):
        h = obs.reshape(len(obs), -1).astype(np.int64)
        key = (h[:, ::997].sum(axis=1) + 31 * h[:, -1]) % 1000
        prob = np.zeros((len(obs), self.n), np.float32)
        for a in range(self.n):
            prob[:, a] = 1 + ((key + 7 * a) % 13)
        prob /= prob.sum(axis=1, keepdims=True)
        acts = (key % self.n).astype(np.uint8)
        return acts, dict(prob=prob, value=(key / 1000.).astype(np.float32))


@pytest.mark.parametrize("mid_batch_reset,max_len", [(True, 23), (False, 17)])
def test_matches_sequential_port(mid_batch_reset, max_len):
    kw = dict(game="breakout", horizon=5, n_parallel=2, envs_per=2, max_path_length=max_len,
              mid_batch_reset=mid_batch_reset, env_kwargs=dict(max_start_noops=5))
    seq = P.CpuSamplerPort(**kw)
    seq.initialize(77, discount=0.99, master_rng=np.random.RandomState(5))
    par = CpuSamplerMP(start_method="fork", **kw)
    try:
        par.initialize(77, discount=0.99, master_rng=np.random.RandomState(5))
        pol = TablePolicy(seq.n_actions)
        done_seq = done_par = 0
        for _ in range(12):
            b0, completed = seq.obtain_samples(pol)
            b1, new = par.obtain_samples(pol)
            done_seq += len(completed)
            done_par += new
            for k in ("observations", "rewards", "dones", "raw_reward", "need_reset", "actions", "prob", "value",
                      "extra_observations"):
                np.testing.assert_array_equal(b0[k], b1[k], err_msg=k)
        assert done_seq == done_par and done_seq > 0
    finally:
        par.shutdown()
