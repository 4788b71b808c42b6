"""TEST INFRASTRUCTURE ONLY -- never imported by the product (accel_rl_amd/).

CPU restatement (numpy) of SURVEY 8(f1): the reference's frame-dedup replay buffer with
n-step return back-fill, uniform index sampling, and the parted sum tree of prioritized
replay.  Struct-of-arrays over all environments (the reference keeps one Python object per
environment); same results, pinned bit for bit by tests/golden/g11_replay.npz and
g12_sumtree.npz, which were recorded from the reference's own classes
(tests/golden/gen_golden_replay.py).

Reference: accel_rl/algos/dqn/replay_buffers/frame.py:23-166 (storage, write_samples,
extract_*), uniform.py:13-59 (sample_idxs), prioritized.py:8-38, sum_tree.py:12-98.
"""
import numpy as np

F32, F64 = np.float32, np.float64


class ReplayPort(object):
    """Per environment: `size` states; frames are stored once each in a ring of
    size + F - 1 slots (the last F-1 slots mirror the first ones after a wrap,
    frame.py:104-107,134-137)."""

    def __init__(self, n_env, n_frames, frame_shape, size, reward_horizon, sampling_horizon, discount,
                 promo="nep50"):
        sampling_size = sampling_horizon * n_env                     # frame.py:41-45
        n_chunks = -(-size // sampling_size)
        self.S = S = n_chunks * sampling_size // n_env
        self.E, self.F, self.T, self.h_r = n_env, n_frames, sampling_horizon, reward_horizon
        self.discount, self.promo = discount, promo
        self.frames = np.zeros((n_env, S + n_frames - 1) + tuple(frame_shape), np.uint8)
        self.n_blanks = np.zeros((n_env, S + n_frames - 1), np.uint8)
        self.acts = np.zeros((n_env, S), np.uint8)
        self.terminals = np.zeros((n_env, S), bool)
        self.rewards = np.zeros((n_env, S), F32)
        self.returns = np.zeros((n_env, S), F32)
        self.idx = 0
        self.full = False

    # ---------------------------------------------------------------- append
    def append(self, obs, acts, rews, dones):
        """obs u8[E,T,F,...], acts u8[E,T], rews f32[E,T], dones bool[E,T] (frame.py:57-60,121-166)."""
        E, S, F, T, h_r, idx = self.E, self.S, self.F, self.T, self.h_r, self.idx
        if idx == 0:                                                 # mirror the ring's tail (:134-137)
            self.frames[:, :F - 1] = self.frames[:, S:S + F - 1]
            self.n_blanks[:, :F - 1] = self.n_blanks[:, S:S + F - 1]
        self.acts[:, idx:idx + T] = acts
        self.rewards[:, idx:idx + T] = rews
        self.terminals[:, idx:idx + T] = dones
        self.frames[:, idx + F - 1:idx + F - 1 + T] = obs[:, :, F - 1]     # newest frame only (:142-143)
        ramp = np.arange(F - 1, 0, -1, dtype=np.uint8)
        for e in range(E):                                           # blank-history marks (:144-155)
            nb = self.n_blanks[e]
            for t in range(T):
                p = idx + t
                if dones[e, t]:
                    nb[p + 1:p + F] = ramp
                elif nb[p + 1] and nb[p + 1] >= nb[p]:
                    nb[p + 1] = 0
        for e in range(E):                                           # n-step returns, h_r - 1 behind (:156-166)
            rw, tm = self.rewards[e], self.terminals[e]
            for t in range(T):
                j = (idx - (h_r - 1) + t) % S
                if self.promo == "nep50":
                    ret = F32(rw[j])
                else:
                    ret = F64(rw[j])
                if not tm[j]:
                    for i in range(1, h_r):
                        k = (j + i) % S
                        if self.promo == "nep50":                    # python float x f32 -> f32, f32 += f32
                            ret = F32(ret + F32(F32(self.discount ** i) * rw[k]))
                        else:                                        # numpy 1.x: python float x f32 scalar -> f64
                            ret = ret + (self.discount ** i) * F64(rw[k])
                        if tm[k]:
                            tm[j] = True
                            break
                self.returns[e, j] = ret
        self.idx = (idx + T) % S
        if self.idx == 0:                                            # uniform.py:13-16
            self.full = True

    # ---------------------------------------------------------------- uniform sampling
    def sample_idxs(self, batch_size, rng=np.random):
        """uniform.py:29-59: two randint draws, then the window of invalid states is skipped."""
        F, S, h_r, idx = self.F, self.S, self.h_r, self.idx
        env_idxs = rng.randint(low=0, high=self.E, size=batch_size)
        high = S - (F - 1) - h_r if self.full else idx - h_r
        step_idxs = rng.randint(low=0, high=high, size=batch_size)
        if idx <= h_r:
            step_idxs += F - 1 + idx
        elif idx >= S - (F - 1):
            step_idxs += (F - 1 + idx) % S
        else:
            step_idxs[step_idxs >= idx - h_r] += (F - 1) + h_r
        return env_idxs, step_idxs

    # ---------------------------------------------------------------- extraction
    def extract_observations(self, env_idxs, step_idxs):
        """frame.py:81-90: F consecutive ring slots, the first n_blanks of them zeroed."""
        env_idxs, step_idxs = np.asarray(env_idxs), np.asarray(step_idxs)
        gather = step_idxs[:, None] + np.arange(self.F)[None, :]
        obs = self.frames[env_idxs[:, None], gather]
        blanks = self.n_blanks[env_idxs, step_idxs]
        obs[np.arange(self.F)[None, :] < blanks[:, None]] = 0
        return obs

    def extract_batch(self, env_idxs, step_idxs):
        """frame.py:69-79"""
        env_idxs, step_idxs = np.asarray(env_idxs), np.asarray(step_idxs)
        nxt = (step_idxs + self.h_r) % self.S
        return (self.extract_observations(env_idxs, step_idxs), self.extract_observations(env_idxs, nxt),
                self.acts[env_idxs, step_idxs], self.returns[env_idxs, step_idxs],
                self.terminals[env_idxs, step_idxs])


class SumTreePort(object):
    """sum_tree.py:12-98.  One f64 sum tree over num_parts x part_size leaves; the states just
    written (zeros_backward behind the cursor) and about to be overwritten (zeros_forward ahead)
    carry zero mass.  Multiple updates of one node are applied in input order (np.add.at), which
    fixes the f64 rounding."""

    def __init__(self, part_size, num_parts, zeros_forward, zeros_backward, default_value, n_advance):
        self.P, self.E = part_size, num_parts
        self.zf, self.zb, self.default, self.n_adv = zeros_forward, zeros_backward, default_value, n_advance
        n_leaves = part_size * num_parts
        self.level = int(np.ceil(np.log2(n_leaves + 1)) + 1)
        self.tree = np.zeros(2 ** self.level - 1)
        self.shift = 2 ** (self.level - 1) - 1
        self.cursor = 0
        last = (np.arange(num_parts)[:, None] + 1) * part_size - 1 - np.arange(zeros_backward)[None, :]
        self.add(last.reshape(-1) + self.shift, np.full(last.size, -default_value))     # :43-52
        self.last_idxs = self.last_probs = None

    def add(self, tree_idxs, diffs):
        """`reconstruct`: push leaf differences up to the root (:54-57)."""
        tree_idxs = np.asarray(tree_idxs).copy()
        for _ in range(self.level):
            np.add.at(self.tree, tree_idxs, diffs)
            tree_idxs = (tree_idxs - 1) // 2

    def advance(self):
        """:59-72 -- switch on the n_advance states that now have a complete n-step return,
        switch off the ones the next append will overwrite (and their frame overlap)."""
        c, P = self.cursor, self.P
        steps = np.arange(self.n_adv)
        on = (c - self.zb + steps) % P
        off = (c - P + self.zf + steps) % P
        parts = np.arange(self.E)[:, None] * P
        on_idx = (parts + on[None, :]).reshape(-1) + self.shift
        off_idx = (parts + off[None, :]).reshape(-1) + self.shift
        idxs = np.concatenate([on_idx, off_idx])
        diffs = np.concatenate([np.full(on_idx.size, self.default), -self.tree[off_idx]])
        self.add(idxs, diffs)
        self.cursor = (c + self.n_adv) % P

    def find(self, uniforms):
        """:88-98 -- descend by prefix mass; `uniforms` in [0, 1], scaled by the root."""
        v = np.array(uniforms, F64) * self.tree[0]
        idx = np.zeros(len(v), np.int64)
        for _ in range(self.level - 1):
            idx = 2 * idx + 1
            left = self.tree[idx]
            right = v > left
            v = np.where(right, v - left, v)
            idx = idx + right
        return idx

    def sample_n(self, n, rng=np.random):
        """:77-86 -- n distinct leaves: sorted unique of 1.05 n draws, topped up 2x the deficit."""
        idxs = np.unique(self.find(rng.rand(int(1.05 * n))))
        tries = 0
        while len(idxs) < n:
            tries += 1
            if tries > 100:
                raise RuntimeError("After 100 tries, unable to get unique idxs")
            idxs = np.unique(np.concatenate([idxs, self.find(rng.rand(2 * (n - len(idxs))))]))
        self.last_idxs = idxs = idxs[:n]
        self.last_probs = probs = self.tree[idxs]
        env_idxs, step_idxs = np.divmod(idxs - self.shift, self.P)
        return env_idxs, step_idxs, probs

    def update_last(self, new_values):
        """:74-75"""
        self.add(self.last_idxs, np.asarray(new_values, F64) - self.last_probs)


def importance_weights(probs, beta):
    """prioritized.py:31-35"""
    w = (1. / np.asarray(probs, F64)) ** beta
    return w / max(w)
