#!/usr/bin/env python
"""
Golden vectors for SURVEY 8(f1): the reference's frame-dedup replay buffer, n-step return
writer, uniform index sampling and parted sum tree.  Runs ONLY in the build container
(needs /root/reference); writes tests/golden/g11_replay.npz and g12_sumtree.npz.

Every recorded array is an input or output of the reference's OWN classes
(accel_rl.algos.dqn.replay_buffers.{frame,uniform,prioritized,sum_tree}), imported as they
are (pure numpy); nothing is computed by this repo's oracle or product.

G11 replay   UniformReplayBuffer: 3 envs x 20 states (F = 4 frames of 6x5 px, reward horizon 3,
             sampling horizon 5), 14 append_data calls (wraps 3 times); after every call the whole
             per-env state, then seeded sample_idxs + extract_batch.  Also F = 2 with reward horizon 1, and horizon 5.
G12 sumtree  PartedSumTree(part 20 x 3 parts): advance / sample_n (with the consumed uniforms
             replayable from the seed) / update_last_samples, the f64 tree after every call; and
             PrioritizedReplayBuffer.sample_batch importance weights.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shims  # noqa: E402

ref_shims.install()

from accel_rl.algos.dqn.replay_buffers.uniform import UniformReplayBuffer  # noqa: E402
from accel_rl.algos.dqn.replay_buffers.prioritized import PrioritizedReplayBuffer  # noqa: E402
from accel_rl.algos.dqn.replay_buffers.sum_tree import PartedSumTree  # noqa: E402


class _Space(object):
    def __init__(self, value):
        self._value = value

    def sample(self):
        return self._value


class _Spec(object):
    def __init__(self, n_frames, h, w):
        self.observation_space = _Space(np.zeros((n_frames, h, w), np.uint8))
        self.action_space = _Space(np.uint8(0))


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print("wrote %-24s %8.1f KB" % (name + ".npz", os.path.getsize(path) / 1024.))


def make_batches(rs, n_batches, n_env, horizon, n_frames, h, w, p_done):
    """Sampler-shaped inputs (env-major).  Observations are consistent frame stacks of an
    incrementing per-env frame stream with blank history after a done, as AtariEnv produces."""
    obs = np.zeros((n_batches, n_env, horizon, n_frames, h, w), np.uint8)
    acts = rs.randint(0, 18, size=(n_batches, n_env, horizon)).astype(np.uint8)
    rews = rs.choice([-1., 0., 0., 1., 0.5], size=(n_batches, n_env, horizon)).astype(np.float32)
    dones = rs.rand(n_batches, n_env, horizon) < p_done
    stack = np.zeros((n_env, n_frames, h, w), np.uint8)
    for e in range(n_env):
        stack[e, -1] = rs.randint(1, 256, size=(h, w))
    for b in range(n_batches):
        for e in range(n_env):
            for t in range(horizon):
                obs[b, e, t] = stack[e]
                new = rs.randint(1, 256, size=(h, w)).astype(np.uint8)
                if dones[b, e, t]:
                    stack[e] = 0
                    stack[e, -1] = new
                else:
                    stack[e] = np.concatenate([stack[e, 1:], new[None]])
    return obs, acts, rews, dones


def samples_data(obs, acts, rews, dones, b):
    n_env = obs.shape[1]
    return dict(segs_view=[dict(observations=obs[b, e], actions=acts[b, e], rewards=rews[b, e],
                                dones=dones[b, e]) for e in range(n_env)])


def state_of(buf):
    return dict(frames=np.stack([e.frames for e in buf.env_bufs]), acts=np.stack([e.acts for e in buf.env_bufs]),
                n_blanks=np.stack([e.n_blanks for e in buf.env_bufs]),
                terminals=np.stack([e.terminals for e in buf.env_bufs]),
                rewards=np.stack([e.rewards for e in buf.env_bufs]),
                returns=np.stack([e.returns for e in buf.env_bufs]))


def g11():
    out = dict()
    for tag, n_frames, h_r, disc in (("f4", 4, 3, 0.99), ("f2", 2, 1, 0.9), ("f4h5", 4, 5, 0.97)):
        rs = np.random.RandomState(11 + n_frames + h_r)
        n_env, horizon, n_batches, h, w = 3, 5, 14, 6, 5
        obs, acts, rews, dones = make_batches(rs, n_batches, n_env, horizon, n_frames, h, w, 0.15)
        buf = UniformReplayBuffer(env_spec=_Spec(n_frames, h, w), size=60, reward_horizon=h_r,
                                  sampling_horizon=horizon, n_environments=n_env, discount=disc)
        assert buf.env_replay_size == 20
        out[tag + "_cfg"] = np.array([n_env, horizon, n_batches, n_frames, h, w, h_r, 20], np.int64)
        out[tag + "_discount"] = np.float64(disc)
        for k, v in (("obs", obs), ("acts", acts), ("rews", rews), ("dones", dones)):
            out["%s_in_%s" % (tag, k)] = v
        for b in range(n_batches):
            buf.append_data(samples_data(obs, acts, rews, dones, b))
            for k, v in state_of(buf).items():
                out["%s_b%02d_%s" % (tag, b, k)] = v.copy()
            out["%s_b%02d_idx_full" % (tag, b)] = np.array([buf.idx, int(buf._buffer_full)], np.int64)
            if buf._buffer_full or buf.idx - h_r > 0:
                np.random.seed(1000 + b)
                env_idxs, step_idxs = buf.sample_idxs(16)
                o, no, a, r, term = buf.extract_batch(env_idxs, step_idxs)
                out["%s_b%02d_env_idxs" % (tag, b)] = np.asarray(env_idxs, np.int64)
                out["%s_b%02d_step_idxs" % (tag, b)] = np.asarray(step_idxs, np.int64)
                for k, v in (("obs", o), ("next_obs", no), ("actions", a), ("returns", r), ("terminals", term)):
                    out["%s_b%02d_x_%s" % (tag, b, k)] = np.asarray(v)
    save("g11_replay", **out)


def g12():
    out = dict()
    rs = np.random.RandomState(12)
    tree = PartedSumTree(part_size=20, num_parts=3, zeros_forward=4, zeros_backward=3,
                         default_value=1.0 ** 0.6, n_advance=5)
    out["cfg"] = np.array([20, 3, 4, 3, 5, tree.tree_level, tree.tree_size, tree.t_l_shift], np.int64)
    out["default_value"] = np.float64(tree.default_value)
    out["tree_init"] = tree.tree.copy()
    step = 0
    for k in range(11):
        tree.advance()
        out["s%02d_advance_tree" % step] = tree.tree.copy()
        out["s%02d_cursor" % step] = np.int64(tree.step_cursor)
        step += 1
        if k >= 1:
            np.random.seed(2000 + k)
            env_idxs, step_idxs, probs = tree.sample_n(8)
            out["s%02d_sample_env" % step] = np.asarray(env_idxs, np.int64)
            out["s%02d_sample_step" % step] = np.asarray(step_idxs, np.int64)
            out["s%02d_sample_probs" % step] = np.asarray(probs, np.float64)
            out["s%02d_sample_seed" % step] = np.int64(2000 + k)
            step += 1
            new = (rs.rand(8) * 2 + 0.01) ** 0.6
            tree.update_last_samples(new)
            out["s%02d_update_values" % step] = new
            out["s%02d_update_tree" % step] = tree.tree.copy()
            step += 1
    out["n_steps"] = np.int64(step)
    # find() on explicit uniforms incl. edge values
    u = np.concatenate([rs.rand(40), [0.0, 1.0 - 1e-12, 0.5]])
    out["find_u"] = u.copy()
    out["find_idx"] = tree.find(u.copy()).astype(np.int64)
    out["find_tree"] = tree.tree.copy()

    # prioritized buffer end-to-end: importance weights + priority update
    n_env, horizon, n_frames, h, w = 3, 5, 4, 6, 5
    rs2 = np.random.RandomState(13)
    obs, acts, rews, dones = make_batches(rs2, 9, n_env, horizon, n_frames, h, w, 0.1)
    buf = PrioritizedReplayBuffer(alpha=0.6, beta_initial=0.4, default_priority=1., env_spec=_Spec(n_frames, h, w),
                                  size=60, reward_horizon=3, sampling_horizon=horizon, n_environments=n_env,
                                  discount=0.99)
    for k, v in (("obs", obs), ("acts", acts), ("rews", rews), ("dones", dones)):
        out["pri_in_%s" % k] = v
    for b in range(9):
        buf.append_data(samples_data(obs, acts, rews, dones, b))
        if b >= 1:
            np.random.seed(3000 + b)
            o, no, a, r, term, isw = buf.sample_batch(6)
            out["pri_b%d_obs" % b] = o
            out["pri_b%d_next_obs" % b] = no
            out["pri_b%d_returns" % b] = np.asarray(r)
            out["pri_b%d_is_weights" % b] = np.asarray(isw, np.float64)
            pri = rs2.rand(6) + 0.05
            buf.update_batch_priorities(pri)
            out["pri_b%d_new_priorities" % b] = pri
        out["pri_b%d_tree" % b] = buf.priority_tree.tree.copy()
    save("g12_sumtree", **out)


if __name__ == "__main__":
    g11()
    g12()
