"""Prioritized replay (reference: accel_rl/algos/dqn/replay_buffers/prioritized.py:5-38)."""
import numpy as np
import torch

from accel_rl_amd import _lib

from accel_rl_amd.algos.dqn.replay_buffers.frame import FrameReplayBuffer
from accel_rl_amd.algos.dqn.replay_buffers.sum_tree import PartedSumTree


class PrioritizedReplayBuffer(FrameReplayBuffer):

    def __init__(self, alpha, beta_initial, default_priority, **kwargs):
        super().__init__(**kwargs)
        self.priority_tree = PartedSumTree(
            part_size=self.env_replay_size,
            num_parts=self.n_environments,
            zeros_forward=self.num_img_obs,
            zeros_backward=self.reward_horizon,
            default_value=default_priority ** alpha,
            n_advance=self.sampling_horizon,
            device=self.device,
        )
        self.alpha = alpha
        self.beta = beta_initial
        self.default_probability = default_priority ** alpha

    def set_beta(self, value):
        self.beta = value

    def append_data(self, samples_data):
        super().append_data(samples_data)
        self.priority_tree.advance()

    def sample_batch(self, batch_size, device_weights=False):
        """-> extract_batch(...) + (importance weights,): host f64 numpy as the reference returns them, or
        (device_weights=True, what the DQN algorithms ask for) a device f32 tensor -- then leaf selection,
        probabilities, batch extraction and weights never visit the host, which waits for one integer."""
        is_weights = self._weights_out(batch_size) if device_weights else None
        dev = self.priority_tree.sample_n_device(batch_size, self.beta, is_weights) if device_weights else None
        if dev is not None:
            env_idxs, step_idxs, probs = dev          # (the importance weights came with the same launch)
            batch_data = self.extract_batch(env_idxs, step_idxs)
            if self.priority_tree.confirm_unique():
                return batch_data + (is_weights,)
            env_idxs, step_idxs, probs = self.priority_tree.top_up()      # the reference would have drawn more
        else:
            env_idxs, step_idxs, probs = self.priority_tree.sample_n(batch_size)
        batch_data = self.extract_batch(env_idxs, step_idxs)
        is_weights = (1. / probs) ** self.beta          # (normalised by the max just below)
        is_weights /= max(is_weights)
        if device_weights:
            out = self._weights_out(batch_size)
            out.copy_(torch.from_numpy(is_weights.astype(np.float32)))
            is_weights = out
        return batch_data + (is_weights,)

    def _weights_out(self, b):
        cache = self.__dict__.setdefault("_isw_cache", dict())
        if self.reuse_outputs and b in cache:
            return cache[b]
        out = torch.empty(b, dtype=torch.float32, device=self.device)
        if self.reuse_outputs:
            out._arl_static = True
            cache[b] = out
        return out

    def update_batch_priorities(self, priorities):
        if isinstance(priorities, torch.Tensor) and priorities.is_cuda and priorities.dtype == torch.float32:
            self.priority_tree.update_last_samples_pow(priorities, self.alpha)
            return
        priorities = np.asarray(priorities.cpu() if hasattr(priorities, "cpu") else priorities)
        self.priority_tree.update_last_samples(priorities ** self.alpha)
