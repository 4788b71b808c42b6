"""Parted sum tree on the device (reference: accel_rl/algos/dqn/replay_buffers/sum_tree.py:4-104).

The f64 tree (16 MB for 1M leaves) lives in HBM; descent (`find`), leaf-to-root updates
(`reconstruct`, in the reference's np.add.at order) and gathers are csrc/replay.hip kernels.
What stays on the host is what must see values to consume the host RNG exactly as the reference
does: the np.random.rand draws of `sample_n` and the decision to draw more when too few distinct
leaves came back.  `sample_n` is the reference's interface (host arrays out);
`sample_n_device` + `confirm_unique` is what the replay buffer uses: descent, sort, unique and
the probabilities stay on the device (arl_sumtree_sample) and the host waits for ONE integer --
the number of distinct leaves -- falling back to the host loop only when the reference would have
topped up (rare at replay sizes)."""
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
        return self._finish_sample(tree_idxs, n)

    # ---- sample_n without the host in the data path ------------------------------------------
    _SAMPLE_MAX = 4096

    def sample_n_device(self, n, beta=None, weights_out=None):
        """Enqueue :77-86 for the common case (enough distinct leaves in the first 1.05 n draws); returns
        device (env_idxs i32[n], step_idxs i32[n], probs f64[n]) -- valid only if `confirm_unique()` says so.
        None when n is too large for the kernel (use sample_n).
        weights_out (f32[n], with beta): the same launch leaves the batch's importance-sampling weights there
        (prioritized.py:33-35).  The host hands over its uniforms in page-locked memory (the kernel reads them where they
        lie) and gets the one integer it needs -- how many distinct leaves -- by a store into page-locked memory that
        confirm_unique() polls: no memcpy node, no event, in either direction."""
        m = int(1.05 * n)
        if m > self._SAMPLE_MAX or m < n:
            return None
        st = getattr(self, "_dev_sample", None)
        if st is None or st["n"] != n:
            i32 = lambda k: torch.empty(k, dtype=torch.int32, device=self.device)      # noqa: E731
            st = self._dev_sample = dict(
                n=n, u_host=torch.empty(m, dtype=torch.float64).pin_memory(),
                u=torch.empty(m, dtype=torch.float64, device=self.device), idx=i32(n), env=i32(n), step=i32(n),
                probs=torch.empty(n, dtype=torch.float64, device=self.device), count=i32(1),
                notify=torch.zeros(1, dtype=torch.int64).pin_memory(), ticket=0, event=torch.cuda.Event())
            st["u_np"], st["notify_np"] = st["u_host"].numpy(), st["notify"].numpy()
        # the previous call's kernel must have read its uniforms before they are overwritten: its notification, which
        # confirm_unique waits for, is the last thing it does (a caller that skipped confirm_unique waits here)
        if not st.get("confirmed", True):
            st["event"].synchronize()
        st["u_np"][:] = np.random.rand(m)                               # the reference's first draw (:79)
        st["ticket"] = (st["ticket"] + 1) & 0x7fffffff
        _lib.sumtree_sample_batch(self.tree, self.tree_level, st["u_host"], n, self.part_size, st["idx"], st["env"],
                                  st["step"], st["probs"], st["count"], beta=0.0 if beta is None else beta,
                                  is_weights=weights_out, notify=st["notify"], ticket=st["ticket"])
        st["event"].record(torch.cuda.current_stream(self.device))
        st["confirmed"] = False
        self.last_tree_idxs, self.last_probs = st["idx"], st["probs"]
        return st["env"], st["step"], st["probs"]

    _POLL_SPINS = 200000        # ~0.1 s of polling before falling back to the stream event

    def confirm_unique(self):
        """Wait for the one integer of the last sample_n_device: were there n distinct leaves?"""
        st = self._dev_sample
        word, want = st["notify_np"], st["ticket"]
        for _ in range(self._POLL_SPINS):
            v = int(word[0])
            if (v >> 32) == want:
                break
        else:
            st["event"].synchronize()
            v = int(word[0])
            assert (v >> 32) == want, "sumtree_sample_batch did not report"
        st["confirmed"] = True
        st["count_host"] = v & 0xffffffff
        return st["count_host"] >= st["n"]

    def top_up(self):
        """The reference's while-loop (:80-86), continuing from the distinct leaves of the last
        sample_n_device; returns what sample_n returns."""
        st = self._dev_sample
        n = st["n"]
        tree_idxs = st["idx"][:int(st["count_host"])].cpu().numpy().astype(np.int64)
        return self._finish_sample(tree_idxs, n)

    def _finish_sample(self, tree_idxs, n):
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

    def update_last_samples_pow(self, priorities, alpha):
        """update_last_samples(priorities ** alpha) for device f32 priorities, without leaving the device."""
        _lib.sumtree_update_pow(self.tree, self.tree_level, self.last_tree_idxs, priorities, self.last_probs, alpha)
