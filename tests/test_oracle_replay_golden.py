"""The oracle's replay-buffer / sum-tree restatement (oracle/replay_port.py) against vectors
recorded from the reference's own classes (tests/golden/gen_golden_replay.py): every stored
array after every append, sampled indices, extracted batches, the f64 tree after every
operation -- bit for bit.  Runs on CPU."""
import os

import numpy as np
import pytest

from oracle import replay_port as R

HERE = os.path.dirname(os.path.abspath(__file__))
G11 = np.load(os.path.join(HERE, "golden", "g11_replay.npz"))
G12 = np.load(os.path.join(HERE, "golden", "g12_sumtree.npz"))


def build(tag, g=G11, promo="nep50"):
    n_env, horizon, n_batches, n_frames, h, w, h_r, s = [int(x) for x in g[tag + "_cfg"]]
    port = R.ReplayPort(n_env, n_frames, (h, w), 60, h_r, horizon, float(g[tag + "_discount"]), promo=promo)
    assert port.S == s
    return port, n_batches


@pytest.mark.parametrize("tag", ["f4", "f2", "f4h5"])
def test_replay_state_sampling_extraction(tag):
    port, n_batches = build(tag)
    obs, acts, rews, dones = (G11["%s_in_%s" % (tag, k)] for k in ("obs", "acts", "rews", "dones"))
    sampled = 0
    for b in range(n_batches):
        port.append(obs[b], acts[b], rews[b], dones[b])
        for k in ("frames", "acts", "n_blanks", "terminals", "rewards", "returns"):
            np.testing.assert_array_equal(getattr(port, k), G11["%s_b%02d_%s" % (tag, b, k)], err_msg="%s b%d %s" % (tag, b, k))
        idx, full = G11["%s_b%02d_idx_full" % (tag, b)]
        assert (port.idx, int(port.full)) == (idx, full)
        key = "%s_b%02d_env_idxs" % (tag, b)
        if key in G11.files:
            np.random.seed(1000 + b)
            e, s = port.sample_idxs(16)
            np.testing.assert_array_equal(e, G11[key])
            np.testing.assert_array_equal(s, G11["%s_b%02d_step_idxs" % (tag, b)])
            got = port.extract_batch(e, s)
            for k, v in zip(("obs", "next_obs", "actions", "returns", "terminals"), got):
                np.testing.assert_array_equal(v, G11["%s_b%02d_x_%s" % (tag, b, k)], err_msg=k)
            sampled += 1
    assert sampled >= 10


def test_legacy_promotion_is_close():
    a, n = build("f4h5")
    b, _ = build("f4h5", promo="legacy")
    obs, acts, rews, dones = (G11["f4h5_in_%s" % k] for k in ("obs", "acts", "rews", "dones"))
    for i in range(n):
        a.append(obs[i], acts[i], rews[i], dones[i])
        b.append(obs[i], acts[i], rews[i], dones[i])
    assert np.allclose(a.returns, b.returns, rtol=1e-6, atol=1e-6) and np.array_equal(a.terminals, b.terminals)


def test_sum_tree_sequence():
    part, parts, zf, zb, n_adv, level, size, shift = [int(x) for x in G12["cfg"]]
    tree = R.SumTreePort(part, parts, zf, zb, float(G12["default_value"]), n_adv)
    assert (tree.level, tree.tree.size, tree.shift) == (level, size, shift)
    np.testing.assert_array_equal(tree.tree, G12["tree_init"])
    n_ops = 0
    for s in range(int(G12["n_steps"])):
        p = "s%02d_" % s
        if p + "advance_tree" in G12.files:
            tree.advance()
            np.testing.assert_array_equal(tree.tree, G12[p + "advance_tree"])
            assert tree.cursor == int(G12[p + "cursor"])
        elif p + "sample_env" in G12.files:
            np.random.seed(int(G12[p + "sample_seed"]))
            e, st, pr = tree.sample_n(8)
            np.testing.assert_array_equal(e, G12[p + "sample_env"])
            np.testing.assert_array_equal(st, G12[p + "sample_step"])
            np.testing.assert_array_equal(pr, G12[p + "sample_probs"])
        else:
            tree.update_last(G12[p + "update_values"])
            np.testing.assert_array_equal(tree.tree, G12[p + "update_tree"])
        n_ops += 1
    assert n_ops == int(G12["n_steps"]) > 25
    tree.tree[:] = G12["find_tree"]
    np.testing.assert_array_equal(tree.find(G12["find_u"]), G12["find_idx"])


def test_prioritized_buffer_end_to_end():
    """PrioritizedReplayBuffer: append + tree advance, sample_batch (obs, returns, importance weights),
    update_batch_priorities (prioritized.py:8-38)."""
    obs, acts, rews, dones = (G12["pri_in_%s" % k] for k in ("obs", "acts", "rews", "dones"))
    port = R.ReplayPort(3, 4, (6, 5), 60, 3, 5, 0.99)
    tree = R.SumTreePort(port.S, 3, 4, 3, 1.0 ** 0.6, 5)
    for b in range(9):
        port.append(obs[b], acts[b], rews[b], dones[b])
        tree.advance()
        if b >= 1:
            np.random.seed(3000 + b)
            e, s, probs = tree.sample_n(6)
            got = port.extract_batch(e, s)
            np.testing.assert_array_equal(got[0], G12["pri_b%d_obs" % b])
            np.testing.assert_array_equal(got[1], G12["pri_b%d_next_obs" % b])
            np.testing.assert_array_equal(got[3], G12["pri_b%d_returns" % b])
            np.testing.assert_array_equal(R.importance_weights(probs, 0.4), G12["pri_b%d_is_weights" % b])
            tree.update_last(G12["pri_b%d_new_priorities" % b] ** 0.6)
        np.testing.assert_array_equal(tree.tree, G12["pri_b%d_tree" % b])
