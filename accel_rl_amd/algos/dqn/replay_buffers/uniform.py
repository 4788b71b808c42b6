"""Uniform replay (reference: accel_rl/algos/dqn/replay_buffers/uniform.py:6-59).  The index
draws stay on the host RNG -- the same two np.random.randint calls in the same order, so a
seeded run samples the reference's transitions -- and the batch is gathered on the device."""
import numpy as np

from accel_rl_amd.algos.dqn.replay_buffers.frame import FrameReplayBuffer


class UniformReplayBuffer(FrameReplayBuffer):

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self._buffer_full = False

    def append_data(self, samples_data):
        super().append_data(samples_data)
        if self.idx == 0:                       # wrapped (always lands on 0)
            self._buffer_full = True

    def sample_batch(self, batch_size):
        env_idxs, step_idxs = self.sample_idxs(batch_size)
        return self.extract_batch(env_idxs, step_idxs)

    def sample_idxs(self, batch_size):
        """uniform.py:29-59: with replacement; the states whose observation or n-step target is
        not valid yet (frame overlap at the write cursor, last reward_horizon states) are skipped."""
        n, idx, h_r, size = self.num_img_obs, self.idx, self.reward_horizon, self.env_replay_size
        env_idxs = np.random.randint(low=0, high=self.n_environments, size=batch_size)
        high = size - (n - 1) - h_r if self._buffer_full else idx - h_r
        step_idxs = np.random.randint(low=0, high=high, size=batch_size)
        if idx <= h_r:
            step_idxs += n - 1 + idx
        elif idx >= size - (n - 1):
            step_idxs += (n - 1 + idx) % size
        else:
            step_idxs[step_idxs >= idx - h_r] += (n - 1) + h_r
        return env_idxs, step_idxs
