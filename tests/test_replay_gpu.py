"""Device replay memory + sum tree (csrc/replay.hip, accel_rl_amd/algos/dqn/replay_buffers/) against
vectors recorded from the reference's own classes (g11 / g12) and against the oracle at
BASELINE config 5's frame size: every stored array after every append, the sampled indices,
the extracted batches and the f64 tree -- bit for bit."""
import os

import numpy as np
import pytest
import torch

from oracle import replay_port as R

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
HERE = os.path.dirname(os.path.abspath(__file__))
G11 = np.load(os.path.join(HERE, "golden", "g11_replay.npz"))
G12 = np.load(os.path.join(HERE, "golden", "g12_sumtree.npz"))


class _Space(object):
    def __init__(self, shape):
        self.shape = shape


class _Spec(object):
    def __init__(self, shape):
        self.observation_space = _Space(shape)


def _samples(obs, acts, rews, dones, b):
    """Sampler layout: env-major flat arrays on the device.  Frames are padded to a multiple of
    16 bytes (the golden frames are 6x5; the product's are 104x80)."""
    n_env, t = obs.shape[1], obs.shape[2]
    o = torch.from_numpy(obs[b].reshape((n_env * t,) + obs.shape[3:])).to(DEV)
    return dict(observations=o, actions=torch.from_numpy(acts[b].reshape(-1)).to(DEV),
                rewards=torch.from_numpy(rews[b].reshape(-1)).to(DEV),
                dones=torch.from_numpy(dones[b].reshape(-1)).to(DEV))


def _pad(frames):
    """6x5 golden frames -> 6x8 (48 bytes, multiple of 16); extra pixels are zero."""
    out = np.zeros(frames.shape[:-1] + (8,), np.uint8)
    out[..., :5] = frames
    return out


@pytest.mark.parametrize("tag", ["f4", "f2", "f4h5"])
def test_uniform_buffer_matches_reference(tag):
    from accel_rl_amd.algos.dqn.replay_buffers.uniform import UniformReplayBuffer
    n_env, horizon, n_batches, n_frames, h, w, h_r, s = [int(x) for x in G11[tag + "_cfg"]]
    obs = _pad(G11[tag + "_in_obs"])
    acts, rews, dones = (G11["%s_in_%s" % (tag, k)] for k in ("acts", "rews", "dones"))
    buf = UniformReplayBuffer(env_spec=_Spec((n_frames, h, 8)), size=60, reward_horizon=h_r,
                              sampling_horizon=horizon, n_environments=n_env,
                              discount=float(G11[tag + "_discount"]), device=DEV)
    assert buf.env_replay_size == s
    for b in range(n_batches):
        buf.append_data(_samples(obs, acts, rews, dones, b))
        want = {k: G11["%s_b%02d_%s" % (tag, b, k)] for k in ("frames", "acts", "n_blanks", "terminals", "rewards", "returns")}
        np.testing.assert_array_equal(buf.frames.cpu().numpy()[..., :5], want["frames"])
        assert not buf.frames.cpu().numpy()[..., 5:].any()
        np.testing.assert_array_equal(buf.acts.cpu().numpy(), want["acts"])
        np.testing.assert_array_equal(buf.n_blanks.cpu().numpy(), want["n_blanks"])
        np.testing.assert_array_equal(buf.terminals.cpu().numpy().astype(bool), want["terminals"])
        np.testing.assert_array_equal(buf.rewards.cpu().numpy(), want["rewards"])
        np.testing.assert_array_equal(buf.returns.cpu().numpy(), want["returns"])
        idx, full = G11["%s_b%02d_idx_full" % (tag, b)]
        assert (buf.idx, int(buf._buffer_full)) == (idx, full)
        key = "%s_b%02d_env_idxs" % (tag, b)
        if key in G11.files:
            np.random.seed(1000 + b)
            o, no, a, r, term = buf.sample_batch(16)
            np.testing.assert_array_equal(o.cpu().numpy()[..., :5], G11["%s_b%02d_x_obs" % (tag, b)])
            np.testing.assert_array_equal(no.cpu().numpy()[..., :5], G11["%s_b%02d_x_next_obs" % (tag, b)])
            np.testing.assert_array_equal(a.cpu().numpy(), G11["%s_b%02d_x_actions" % (tag, b)])
            np.testing.assert_array_equal(r.cpu().numpy(), G11["%s_b%02d_x_returns" % (tag, b)])
            np.testing.assert_array_equal(term.cpu().numpy(), G11["%s_b%02d_x_terminals" % (tag, b)])


def test_sum_tree_matches_reference():
    from accel_rl_amd.algos.dqn.replay_buffers.sum_tree import PartedSumTree
    part, parts, zf, zb, n_adv, level, size, shift = [int(x) for x in G12["cfg"]]
    tree = PartedSumTree(part, parts, zf, zb, float(G12["default_value"]), n_adv, device=DEV)
    assert (tree.tree_level, tree.tree_size, tree.t_l_shift) == (level, size, shift)
    np.testing.assert_array_equal(tree.tree.cpu().numpy(), G12["tree_init"])
    for s in range(int(G12["n_steps"])):
        p = "s%02d_" % s
        if p + "advance_tree" in G12.files:
            tree.advance()
            np.testing.assert_array_equal(tree.tree.cpu().numpy(), G12[p + "advance_tree"])
            assert tree.step_cursor == int(G12[p + "cursor"])
        elif p + "sample_env" in G12.files:
            np.random.seed(int(G12[p + "sample_seed"]))
            e, st, pr = tree.sample_n(8)
            np.testing.assert_array_equal(e, G12[p + "sample_env"])
            np.testing.assert_array_equal(st, G12[p + "sample_step"])
            np.testing.assert_array_equal(pr, G12[p + "sample_probs"])
        else:
            tree.update_last_samples(G12[p + "update_values"])
            np.testing.assert_array_equal(tree.tree.cpu().numpy(), G12[p + "update_tree"])
    tree.tree.copy_(torch.from_numpy(G12["find_tree"]))
    np.testing.assert_array_equal(tree.find(G12["find_u"]), G12["find_idx"])


def test_prioritized_buffer_matches_reference():
    from accel_rl_amd.algos.dqn.replay_buffers.prioritized import PrioritizedReplayBuffer
    obs = _pad(G12["pri_in_obs"])
    acts, rews, dones = (G12["pri_in_%s" % k] for k in ("acts", "rews", "dones"))
    buf = PrioritizedReplayBuffer(alpha=0.6, beta_initial=0.4, default_priority=1., env_spec=_Spec((4, 6, 8)),
                                  size=60, reward_horizon=3, sampling_horizon=5, n_environments=3,
                                  discount=0.99, device=DEV)
    for b in range(9):
        buf.append_data(_samples(obs, acts, rews, dones, b))
        if b >= 1:
            np.random.seed(3000 + b)
            o, no, a, r, term, isw = buf.sample_batch(6)
            np.testing.assert_array_equal(o.cpu().numpy()[..., :5], G12["pri_b%d_obs" % b])
            np.testing.assert_array_equal(no.cpu().numpy()[..., :5], G12["pri_b%d_next_obs" % b])
            np.testing.assert_array_equal(r.cpu().numpy(), G12["pri_b%d_returns" % b])
            np.testing.assert_array_equal(isw, G12["pri_b%d_is_weights" % b])
            buf.update_batch_priorities(G12["pri_b%d_new_priorities" % b])
        np.testing.assert_array_equal(buf.priority_tree.tree.cpu().numpy(), G12["pri_b%d_tree" % b])


@pytest.mark.parametrize("promo", ["nep50", "legacy"])
def test_config5_shapes_against_oracle(promo):
    """Seaquest-shaped frames (4 x 104 x 80), 64 envs, many wraps, both numpy promotions; a large
    random tree update exercises the chunked in-order add (n > 4096)."""
    from accel_rl_amd import _lib
    from accel_rl_amd.algos.dqn.replay_buffers.uniform import UniformReplayBuffer
    from accel_rl_amd.algos.dqn.replay_buffers.sum_tree import PartedSumTree
    rs = np.random.RandomState(5)
    n_env, t, f, h_r, size = 64, 4, 4, 3, 64 * 40
    buf = UniformReplayBuffer(env_spec=_Spec((f, 104, 80)), size=size, reward_horizon=h_r, sampling_horizon=t,
                              n_environments=n_env, discount=0.99, device=DEV,
                              promo=_lib.PROMO_NEP50 if promo == "nep50" else _lib.PROMO_LEGACY)
    port = R.ReplayPort(n_env, f, (104, 80), size, h_r, t, 0.99, promo=promo)
    for b in range(25):
        obs = rs.randint(0, 256, size=(n_env, t, f, 104, 80), dtype=np.uint8)
        acts = rs.randint(0, 18, size=(n_env, t)).astype(np.uint8)
        rews = rs.randn(n_env, t).astype(np.float32)
        dones = rs.rand(n_env, t) < 0.1
        port.append(obs, acts, rews, dones)
        buf.append_data(dict(observations=torch.from_numpy(obs.reshape(n_env * t, f, 104, 80)).to(DEV),
                             actions=torch.from_numpy(acts.reshape(-1)).to(DEV),
                             rewards=torch.from_numpy(rews.reshape(-1)).to(DEV),
                             dones=torch.from_numpy(dones.reshape(-1)).to(DEV)))
    for k in ("frames", "n_blanks", "acts", "rewards", "returns"):
        np.testing.assert_array_equal(getattr(buf, k).cpu().numpy(), getattr(port, k), err_msg=k)
    np.testing.assert_array_equal(buf.terminals.cpu().numpy().astype(bool), port.terminals)
    np.random.seed(9)
    e, s = buf.sample_idxs(512)
    got = buf.extract_batch(e, s)
    want = port.extract_batch(e, s)
    for g, w_ in zip(got, want):
        np.testing.assert_array_equal(g.cpu().numpy(), w_)
    # sum tree: 6000 updates with many repeated leaves, in-order accumulation across chunks
    tree = PartedSumTree(size // n_env, n_env, f, h_r, 1.0, t, device=DEV)
    ref = R.SumTreePort(size // n_env, n_env, f, h_r, 1.0, t)
    for _ in range(3):
        tree.advance()
        ref.advance()
    idxs = rs.randint(0, size, size=6000) + ref.shift
    diffs = rs.randn(6000)
    tree.reconstruct(idxs, diffs)
    ref.add(idxs, diffs)
    np.testing.assert_array_equal(tree.tree.cpu().numpy(), ref.tree)
    u = rs.rand(1000)
    np.testing.assert_array_equal(tree.find(u), ref.find(u))


@pytest.mark.parametrize("levels,m,n", [(5, 9, 8), (8, 33, 32), (12, 537, 512), (21, 33, 32), (10, 4096, 3900),
                                        (4, 40, 3)])
def test_sumtree_sample_kernel_matches_host_unique(levels, m, n):
    """arl_sumtree_sample = find + np.unique + [:n] + gather + divmod of sum_tree.py:77-86, and the count of
    distinct leaves -- including trees so small that most draws collide."""
    from accel_rl_amd import _lib
    rs = np.random.RandomState(levels * 1000 + m)
    n_leaves = 2 ** (levels - 1)
    leaves = rs.rand(n_leaves) * (rs.rand(n_leaves) < 0.7)             # zero-priority leaves are never found
    tree = np.zeros(2 ** levels - 1)
    tree[n_leaves - 1:] = leaves
    for i in range(n_leaves - 2, -1, -1):
        tree[i] = tree[2 * i + 1] + tree[2 * i + 2]
    t = torch.from_numpy(tree).to(DEV)
    u = rs.rand(m)
    found = torch.empty(m, dtype=torch.int32, device=DEV)
    _lib.sumtree_find(t, levels, torch.from_numpy(u).to(DEV), found)
    uniq = np.unique(found.cpu().numpy())
    part = 7 if levels > 4 else 3
    i32 = lambda: torch.full((n,), -7, dtype=torch.int32, device=DEV)      # noqa: E731
    idx, env, step = i32(), i32(), i32()
    probs = torch.full((n,), -1., dtype=torch.float64, device=DEV)
    count = torch.zeros(1, dtype=torch.int32, device=DEV)
    _lib.sumtree_sample(t, levels, torch.from_numpy(u).to(DEV), n, part, idx, env, step, probs, count)
    c = int(count.item())
    assert c == len(uniq)
    k = min(c, n)
    np.testing.assert_array_equal(idx.cpu().numpy()[:k], uniq[:k])
    np.testing.assert_array_equal(probs.cpu().numpy()[:k], tree[uniq[:k]])
    e, s = np.divmod(uniq[:k] - (n_leaves - 1), part)
    np.testing.assert_array_equal(env.cpu().numpy()[:k], e)
    np.testing.assert_array_equal(step.cpu().numpy()[:k], s)
    assert (idx.cpu().numpy()[k:] == uniq[0]).all()                  # the slots past the distinct leaves repeat the first


def test_prioritized_device_path_tracks_the_reference_path():
    """sample_batch(device_weights=True) + update_batch_priorities(device f32) -- the path the DQN algorithms
    take -- against the host path (itself pinned bit for bit to the reference, above) on the same seeds:
    identical leaves and batches (integer work: exact), weights to f32 rounding, tree to 1e-7 (f32 pow).
    The golden tree is small, so the top-up fallback is exercised as well."""
    from accel_rl_amd.algos.dqn.replay_buffers.prioritized import PrioritizedReplayBuffer
    obs = _pad(G12["pri_in_obs"])
    acts, rews, dones = (G12["pri_in_%s" % k] for k in ("acts", "rews", "dones"))
    mk = lambda: PrioritizedReplayBuffer(alpha=0.6, beta_initial=0.4, default_priority=1., env_spec=_Spec((4, 6, 8)),   # noqa: E731
                                         size=60, reward_horizon=3, sampling_horizon=5, n_environments=3,
                                         discount=0.99, device=DEV)
    host, dev = mk(), mk()
    fast = slow = 0
    rs = np.random.RandomState(0)
    for b in range(9):
        for buf in (host, dev):
            buf.append_data(_samples(obs, acts, rews, dones, b))
        if b >= 1:
            for batch in (6, 3, 6):
                seed = 3000 + 10 * b + batch
                np.random.seed(seed)
                want = host.sample_batch(batch)
                after = np.random.randint(0, 2 ** 31 - 1)
                np.random.seed(seed)
                got = dev.sample_batch(batch, device_weights=True)
                assert np.random.randint(0, 2 ** 31 - 1) == after           # same host RNG consumption
                st = dev.priority_tree._dev_sample
                if int(st["count_host"]) >= batch:
                    fast += 1
                else:
                    slow += 1
                for w, g in zip(want[:5], got[:5]):
                    assert torch.equal(w, g)
                assert got[5].dtype == torch.float32 and got[5].is_cuda
                np.testing.assert_allclose(got[5].cpu().numpy(), want[5].astype(np.float32), rtol=3e-7, atol=0)
                np.testing.assert_array_equal(dev.priority_tree.last_tree_idxs.cpu().numpy(),
                                              host.priority_tree.last_tree_idxs.cpu().numpy())
                pri = (rs.rand(batch) * 2 + 0.01).astype(np.float32)
                host.update_batch_priorities(pri)
                dev.update_batch_priorities(torch.from_numpy(pri).to(DEV))
                np.testing.assert_allclose(dev.priority_tree.tree.cpu().numpy(), host.priority_tree.tree.cpu().numpy(),
                                           rtol=2e-7, atol=1e-12)
    assert fast > 0 and slow > 0, (fast, slow)


def test_config5_store_at_its_stated_size():
    """BASELINE config 5 as stated: the 1 000 448-transition store of 256 environments -- 8.33 GB of
    frames, i.e. byte offsets past 2^32 from environment 131 on -- appended to until every environment's ring has wrapped,
    then `extract_batch` against the oracle port at states on both sides of the 2^32-byte boundary and of the wrap, and
    one sampling round of the 21-level sum tree through the device path against the host path.
    The port mirrors four of the 256 environments (every environment is independent of the others:
    accel_rl/algos/dqn/replay_buffers/frame.py:23-90 keeps one object per environment; sum tree: sum_tree.py:12-98)."""
    from accel_rl_amd.algos.dqn.replay_buffers.prioritized import PrioritizedReplayBuffer
    free, _ = torch.cuda.mem_get_info()
    if free < 12 << 30:
        pytest.skip("needs 12 GB of free HBM")
    n_env, t, f, h_r, size = 256, 4, 4, 3, 1000448
    mk = lambda: PrioritizedReplayBuffer(alpha=0.6, beta_initial=0.4, default_priority=1., env_spec=_Spec((f, 104, 80)),   # noqa: E731
                                         size=size, reward_horizon=h_r, sampling_horizon=t, n_environments=n_env,
                                         discount=0.99, device=DEV)
    buf = mk()
    S = buf.env_replay_size
    assert S == 3908 and buf.frames.numel() == n_env * (S + f - 1) * 8320 and buf.frames.numel() > 2 ** 33 - 2 ** 30
    frame_bytes = (S + f - 1) * 8320
    e_lo = (2 ** 32) // frame_bytes                          # the environment whose ring the 2^32-byte boundary cuts
    assert e_lo == 131
    s_cut = (2 ** 32 - e_lo * frame_bytes) // 8320           # ... and the ring slot it falls into
    envs = [0, e_lo, e_lo + 1, n_env - 1]
    port = R.ReplayPort(len(envs), f, (104, 80), S * len(envs), h_r, t, 0.99)
    assert port.S == S
    g = torch.Generator(device=DEV).manual_seed(7)
    sel = torch.tensor(envs, device=DEV)
    n_app = S // t + 23                                      # every ring wraps; the cursor ends at 92
    for b in range(n_app):
        obs = torch.randint(0, 256, (n_env, t, f, 104, 80), dtype=torch.uint8, device=DEV, generator=g)
        acts = torch.randint(0, 18, (n_env, t), dtype=torch.uint8, device=DEV, generator=g)
        rews = torch.randn((n_env, t), device=DEV, generator=g)
        dones = torch.rand((n_env, t), device=DEV, generator=g) < 0.02
        buf.append_data(dict(observations=obs.reshape(n_env * t, f, 104, 80), actions=acts.reshape(-1),
                             rewards=rews.reshape(-1), dones=dones.reshape(-1)))
        port.append(obs[sel].cpu().numpy(), acts[sel].cpu().numpy(), rews[sel].cpu().numpy(), dones[sel].cpu().numpy())
    assert buf.idx == port.idx == (n_app * t) % S and port.full
    for k in ("n_blanks", "acts", "rewards", "returns"):
        np.testing.assert_array_equal(getattr(buf, k)[sel].cpu().numpy(), getattr(port, k), err_msg=k)
    np.testing.assert_array_equal(buf.terminals[sel].cpu().numpy().astype(bool), port.terminals)
    # states around the wrap of the ring, around the write cursor, and around the 2^32-byte boundary (environment e_lo)
    steps = list(range(0, 12)) + list(range(S - 12, S)) + list(range(port.idx - 8, port.idx + 8)) + \
        list(range(s_cut - 10, s_cut + 6)) + [1000, 2000, 3000]
    pe = np.repeat(np.arange(len(envs)), len(steps))
    ps = np.tile(np.array(steps), len(envs))
    got = buf.extract_batch(np.array(envs)[pe], ps)
    want = port.extract_batch(pe, ps)
    for gg, ww in zip(got, want):
        np.testing.assert_array_equal(gg.cpu().numpy(), ww)
    cut_rows = (pe == 1) & (ps + f - 1 >= s_cut) & (ps <= s_cut)
    assert cut_rows.sum() >= f                               # observations that straddle the 2^32-byte boundary
    assert got[0].cpu().numpy()[cut_rows].any()
    # ---- the 21-level tree: one sampling round, device path against the host path from the same state and seed
    tree = buf.priority_tree
    assert tree.tree_level == 21 and tree.tree.numel() == 2 ** 21 - 1 and tree.t_l_shift == 2 ** 20 - 1
    pri = torch.rand(size, dtype=torch.float64, device=DEV, generator=g) + 0.01
    leaves = tree.tree[tree.t_l_shift:tree.t_l_shift + size]
    live = leaves > 0                                        # the zeroed window around the cursor stays zero
    leaves[live] = pri[live]                                 # random priorities on every live state, then the sums above them
    for lvl in range(19, -1, -1):
        lo, hi = 2 ** lvl - 1, 2 ** (lvl + 1) - 1
        tree.tree[lo:hi] = tree.tree[2 * lo + 1:2 * hi + 1:2] + tree.tree[2 * lo + 2:2 * hi + 2:2]
    np.random.seed(11)
    want = buf.sample_batch(512)
    want = [w.clone() if isinstance(w, torch.Tensor) else np.array(w) for w in want]   # (outputs may be reused buffers)
    after = np.random.randint(0, 2 ** 31 - 1)
    want_idx = tree.last_tree_idxs.cpu().numpy().copy()
    np.random.seed(11)
    got = buf.sample_batch(512, device_weights=True)
    assert np.random.randint(0, 2 ** 31 - 1) == after        # same consumption of the host RNG
    np.testing.assert_array_equal(tree.last_tree_idxs.cpu().numpy(), want_idx)
    for w, gg in zip(want[:5], got[:5]):
        assert torch.equal(w, gg)
    np.testing.assert_allclose(got[5].cpu().numpy(), np.asarray(want[5]).astype(np.float32), rtol=3e-7, atol=0)
    e_s = np.divmod(want_idx - tree.t_l_shift, S)
    assert (e_s[0] >= e_lo).any() and (e_s[0] < e_lo).any()  # the batch drew from both sides of the boundary
