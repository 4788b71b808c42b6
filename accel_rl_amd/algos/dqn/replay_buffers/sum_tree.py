"""Parted sum tree on the device (reference: accel_rl/algos/dqn/replay_buffers/sum_tree.py:4-104).

The f64 tree (16 MB for 1M leaves) lives in HBM; descent (`find`), leaf-to-root updates
(`reconstruct`, in the reference's np.add.at order) and gathers are csrc/replay.hip kernels.
What stays on the host is what must see the values to consume the host RNG exactly as the
reference does: the np.random.rand draws and the sorted-unique / top-up loop of `sample_n`
(a few dozen integers per call)."""
import numpy as np
import torch

from accel_rl_amd import _lib


class PartedSumTree(object):

    def __init__(self, part_size, num_parts, zeros_forward, zeros_backward, default_value, n_advance,
                 device="cuda:0"):
        _lib.load()
        self.part_size, self.num_parts = part_size, num_parts
        self.zeros_forward, self.zeros_backward = zeros_forward, zeros_backward
        self.default_value = default_value
        self.n_leaves = part_size * num_parts
        self.tree_level = int(np.ceil(np.log2(self.n_leaves + 1)) + 1)
        self.tree_size = 2 ** self.tree_level - 1
        self.t_l_shift = 2 ** (self.tree_level - 1) - 1
        self.device = torch.device(device)
        self.tree = torch.zeros(self.tree_size, dtype=torch.float64, device=self.device)
        self.n_advance = n_advance
        assert part_size % n_advance == 0
        self.step_cursor = 0
        self.n_ons = n_advance * num_parts
        # initial values: the last zeros_backward leaves of every part start switched off (:43-52)
        last = (np.arange(num_parts)[:, None] + 1) * part_size - 1 - np.arange(zeros_backward)[None, :]
        self.reconstruct(last.reshape(-1) + self.t_l_shift, -default_value * np.ones(last.size))

    def _dev(self, array, dtype):
        return torch.from_numpy(np.ascontiguousarray(array)).to(dtype).to(self.device)

    def reconstruct(self, tree_idxs, diffs):
        if not isinstance(tree_idxs, torch.Tensor):
            tree_idxs = self._dev(tree_idxs, torch.int32)
        if not isinstance(diffs, torch.Tensor):
            diffs = self._dev(diffs, torch.float64)
        _lib.sumtree_add(self.tree, self.tree_level, tree_idxs, diffs)

    def advance(self):
        """:59-72"""
        c, p = self.step_cursor, self.part_size
        steps = np.arange(self.n_advance)
        parts = np.arange(self.num_parts)[:, None] * p
        on = (parts + ((c - self.zeros_backward + steps) % p)[None, :]).reshape(-1)
        off = (parts + ((c - p + self.zeros_forward + steps) % p)[None, :]).reshape(-1)
        idxs = self._dev(np.concatenate([on, off]) + self.t_l_shift, torch.int32)
        diffs = torch.full((2 * self.n_ons,), float(self.default_value), dtype=torch.float64, device=self.device)
        _lib.sumtree_gather(self.tree, idxs[self.n_ons:], diffs[self.n_ons:], scale=-1.0)
        _lib.sumtree_add(self.tree, self.tree_level, idxs, diffs)
        self.step_cursor = (c + self.n_advance) % p

    def find(self, random_values):
        """:88-98; returns host int64 tree indices."""
        u = self._dev(np.asarray(random_values, np.float64), torch.float64)
        out = torch.empty(u.numel(), dtype=torch.int32, device=self.device)
        _lib.sumtree_find(self.tree, self.tree_level, u, out)
        return out.cpu().numpy().astype(np.int64)

    def sample_n(self, n):
        """:77-86: n distinct leaves (sorted), their parts / steps / probabilities."""
        tree_idxs = np.unique(self.find(np.random.rand(int(1.05 * n))))
        i = 0
        while len(tree_idxs) < n:
            i += 1
            if i > 100:
                raise RuntimeError("After 100 tries, unable to get unique idxs")
            new_idxs = self.find(np.random.rand(2 * (n - len(tree_idxs))))
            tree_idxs = np.unique(np.concatenate([tree_idxs, new_idxs]))
        tree_idxs = tree_idxs[:n]
        self.last_tree_idxs = self._dev(tree_idxs, torch.int32)
        probs = torch.empty(n, dtype=torch.float64, device=self.device)
        _lib.sumtree_gather(self.tree, self.last_tree_idxs, probs)
        self.last_probs = probs
        env_idxs, step_idxs = np.divmod(tree_idxs - self.t_l_shift, self.part_size)
        return env_idxs, step_idxs, probs.cpu().numpy()

    def update_last_samples(self, new_values):
        """:74-75"""
        new = new_values if isinstance(new_values, torch.Tensor) else self._dev(new_values, torch.float64)
        self.reconstruct(self.last_tree_idxs, new.to(torch.float64) - self.last_probs)
