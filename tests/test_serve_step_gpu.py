"""GPU parity of arl_env_step_served -- the hidden layer's fold, the output layers, the action draw, the env step and the
next observation's first convolution in ONE launch per step (csrc/serve_step.hip) --

  (1) against the separate launches it replaces (arl_conv2d_fwd's fold + arl_pg_head_infer + arl_env_step +
      arl_conv2d_u8_fwd): every array of the rollout buffer, the env state and the trajectory records bit for bit, over
      batches with life losses, game-overs, over-length resets and start no-op draws, eager and as a hipGraph;
  (2) against the oracle's sequential sampler port (oracle/ref_port.py CpuSamplerPort: AtariEnv, collectors, TrajInfo,
      weighted_sample_n restated from the reference), which is served the SAME network by the separate device kernels.

Reference: accel_rl/sampler/act_server/alternating/overlap/sampler.py:120-151, overlap/worker.py:37-59,
policies/pg/atari_cnn_policy.py:63-67, rllab/misc/special.py:22-27.  Everything compared is bit-exact.
"""
import numpy as np
import pytest
import torch

from oracle import ref_port as P

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def make_sampler(game, n_parallel, envs_per, horizon, seed, max_path_length, env_kwargs, policy, use_graph, served):
    from accel_rl_amd.envs.synthetic_atari import SynthAtariEnv
    from accel_rl_amd.sampler.gpu_sampler import GpuVecSampler
    from accel_rl_amd.util import logger
    logger.set_quiet(True)

    class Sampler(GpuVecSampler):
        _serve_in_step = served

    env_args = dict(env_kwargs)
    env_args["game"] = game
    smp = Sampler(EnvCls=SynthAtariEnv, env_args=env_args, horizon=horizon, n_parallel=n_parallel, envs_per=envs_per,
                  mid_batch_reset=True, max_path_length=max_path_length, max_decorrelation_steps=0, device=DEV,
                  use_graph=use_graph)
    np.random.seed(seed)
    env_spec, *_ = smp.initialize(seed=seed + 1, affinities=dict(), discount=0.99, need_extra_obs=True)
    if getattr(policy, "env_spec", None) is None:
        np.random.seed(seed + 1000)
        policy.initialize(env_spec, device=DEV)
    np.random.seed(seed + 2000)                         # policy_init draws the reference's example observation / action
    smp.policy_init(policy)
    return smp


def host(buf):
    out = dict(observations=buf.observations, extra_observations=buf.extra_observations, actions=buf.actions,
               rewards=buf.rewards, dones=buf.dones, prob=buf.agent_infos["prob"], value=buf.agent_infos["value"])
    out.update(("env_" + k, v) for k, v in buf.env_infos.items())
    return {k: v.cpu().numpy().copy() for k, v in out.items()}


def env_state(smp):
    keys = ("tick", "emu_lives", "env_lives", "over", "traj_len", "traj_nonzero", "traj_ret", "traj_raw", "traj_disc",
            "traj_curdisc", "noop_cursor", "next_reset", "launch_count", "reset_flag", "frame_a", "frame_b", "frame_mode")
    return {k: smp._st[k].cpu().numpy().copy() for k in keys}


def traj_tuples(infos):
    return sorted((ti._env, ti.Length, ti.Return, ti.RawReturn, ti.NonzeroRewards, ti.DiscountedReturn) for ti in infos)


@pytest.mark.parametrize("use_graph", [False, True])
@pytest.mark.parametrize("game,spec,n_parallel,envs_per,max_len", [
    ("breakout", 1, 16, 8, 9),        # BASELINE config 2's sampler: 256 envs, spec-1 CNN (conv 1 inside the step launch)
    ("breakout", 1, 64, 8, 9),        # 1024 envs: the upper end of the size rule (four workgroups per CU in turn)
    ("seaquest", 1, 3, 5, 7),         # 18 actions, odd stream sizes, episodes end every other batch
    ("pong", 0, 4, 2, 11),            # spec 0: 16 filters of 8 x 8 (half of conv 1's MFMA tile idle), hid 256
    ("qbert", 1, 2, 2, 66),           # long episodes: life losses before any over-length reset
])
def test_served_step_is_bit_identical_to_the_separate_launches(game, spec, n_parallel, envs_per, max_len, use_graph):
    from accel_rl_amd.policies.atari_cnn_policy import AtariCnnPolicy
    from accel_rl_amd.policies.atari_cnn_specs import cnn_specs
    horizon, seed = 5, 31
    n_batches = 16 if max_len > 20 else 6
    policy = AtariCnnPolicy(**cnn_specs[spec])
    kw = dict(max_start_noops=30)
    a = make_sampler(game, n_parallel, envs_per, horizon, seed, max_len, kw, policy, use_graph, served=False)
    b = make_sampler(game, n_parallel, envs_per, horizon, seed, max_len, kw, policy, use_graph, served=True)
    assert not a._serve_fused and b._serve_fused and a._single_write and b._single_write
    # sharpen the policy: a freshly initialised head is almost uniform, which would leave the softmax untested
    with torch.no_grad():
        policy.flat_params[policy._offsets[policy._k_head]:].mul_(40.0)
    state = np.random.get_state()
    n_done, n_traj = 0, 0
    for k in range(n_batches):
        np.random.set_state(state)
        buf_a, infos_a = a.obtain_samples(k)
        got_a, traj_a = host(buf_a), traj_tuples(infos_a)
        np.random.set_state(state)
        buf_b, infos_b = b.obtain_samples(k)
        got_b, traj_b = host(buf_b), traj_tuples(infos_b)
        state = np.random.get_state()
        torch.cuda.synchronize()
        for key in got_a:
            np.testing.assert_array_equal(got_b[key], got_a[key], err_msg="%s %s batch %d" % (game, key, k))
        sa, sb = env_state(a), env_state(b)
        for key in sa:
            np.testing.assert_array_equal(sb[key], sa[key], err_msg="state %s batch %d" % (key, k))
        assert traj_a == traj_b
        n_done += int(got_a["dones"].sum())
        n_traj += len(traj_a)
        if k == 2:                       # the parameters move between batches (as under a learner): nothing may be stale
            with torch.no_grad():
                policy.flat_params.mul_(1.01)
    assert n_done > 0 and n_traj > 0
    assert len(np.unique(got_a["actions"])) > 1
    a.shutdown()
    b.shutdown()


def test_served_step_through_the_noop_ring_refills():
    """350 batches of 2-step episodes: every worker stream consumes more start no-op draws than its device ring holds
    (4096), so the ring is topped up from the streams' RandomStates several times and the cursors wrap -- the served
    launch's forecast / rank / cursor logic against the separate launches, every array bit for bit (compared on the
    device), no forecast mismatch."""
    from accel_rl_amd.policies.atari_cnn_policy import AtariCnnPolicy
    from accel_rl_amd.policies.atari_cnn_specs import cnn_specs
    from accel_rl_amd.sampler import gpu_sampler
    policy = AtariCnnPolicy(**cnn_specs[1])
    kw = dict(max_start_noops=30)
    a = make_sampler("breakout", 1, 8, 5, 41, 1, kw, policy, True, served=False)
    b = make_sampler("breakout", 1, 8, 5, 41, 1, kw, policy, True, served=True)
    with torch.no_grad():
        policy.flat_params[policy._offsets[policy._k_head]:].mul_(40.0)
    state = np.random.get_state()
    n_traj = 0
    for k in range(350):
        np.random.set_state(state)
        buf_a, infos_a = a.obtain_samples(k)
        ta = traj_tuples(infos_a)
        np.random.set_state(state)
        buf_b, infos_b = b.obtain_samples(k)
        tb = traj_tuples(infos_b)
        state = np.random.get_state()
        assert ta == tb, k
        n_traj += len(ta)
        for x, y in ((buf_a.observations, buf_b.observations), (buf_a.actions, buf_b.actions), (buf_a.rewards, buf_b.rewards),
                     (buf_a.dones, buf_b.dones), (buf_a.agent_infos["prob"], buf_b.agent_infos["prob"]),
                     (buf_a.agent_infos["value"], buf_b.agent_infos["value"]),
                     (buf_a.extra_observations, buf_b.extra_observations), (a._st.noop_cursor, b._st.noop_cursor),
                     (a._st.tick, b._st.tick), (a._st.next_reset, b._st.next_reset)):
            assert torch.equal(x, y), k
    assert int(a._st.noop_cursor.max().item()) > gpu_sampler.NOOP_RING          # the ring went round
    assert n_traj > 16 * 500 and int(b._st.epoch[2].item()) == 0
    a.shutdown()
    b.shutdown()


class EchoPolicy(object):
    """Host policy for the oracle's sampler port: serves a group's observations with the DEVICE network through the
    separate kernels (policy.prob_value on a full-size batch with the group's rows at their env positions, so that
    every launch has the served sampler's shapes and therefore its summation orders), samples on the host with the
    oracle's weighted_sample_n."""

    def __init__(self, policy, n_env):
        self.policy, self.n_env, self.calls = policy, n_env, 0

    def get_actions(self, obs):
        half = self.n_env // 2
        group = self.calls % 2                      # the port alternates its two groups (sampler.py:129-145)
        self.calls += 1
        assert obs.shape[0] == half
        full = torch.zeros((self.n_env,) + obs.shape[1:], dtype=torch.uint8, device=DEV)
        full[group * half:(group + 1) * half] = torch.from_numpy(np.ascontiguousarray(obs)).to(DEV)
        prob, value = self.policy.prob_value(full)
        prob = prob[group * half:(group + 1) * half].cpu().numpy()
        value = value[group * half:(group + 1) * half].cpu().numpy()
        return P.sample_actions(prob, np.random.rand(half)), dict(prob=prob, value=value)


@pytest.mark.parametrize("game,n_parallel,envs_per", [("breakout", 16, 8), ("seaquest", 2, 3)])
def test_served_step_matches_the_oracle_sampler_port(game, n_parallel, envs_per):
    from accel_rl_amd.policies.atari_cnn_policy import AtariCnnPolicy
    from accel_rl_amd.policies.atari_cnn_specs import cnn_specs
    horizon, seed, max_len, n_batches = 5, 17, 9, 4
    kw = dict(max_start_noops=30)
    policy = AtariCnnPolicy(**cnn_specs[1])
    smp = make_sampler(game, n_parallel, envs_per, horizon, seed, max_len, kw, policy, True, served=True)
    assert smp._serve_fused
    with torch.no_grad():
        policy.flat_params[policy._offsets[policy._k_head]:].mul_(40.0)
    n_env = 2 * n_parallel * envs_per
    ora = P.CpuSamplerPort(game, horizon, n_parallel, envs_per, max_path_length=max_len, mid_batch_reset=True, env_kwargs=kw)
    np.random.seed(seed)
    ora.initialize(seed + 1, discount=0.99)
    echo = EchoPolicy(policy, n_env)
    state = np.random.get_state()
    n_completed = 0
    for b in range(n_batches):
        np.random.set_state(state)
        buf, infos = smp.obtain_samples(b)
        got = host(buf)
        got_t = sorted((ti.Length, ti.Return, ti.RawReturn, ti.NonzeroRewards, ti.DiscountedReturn) for ti in infos)
        np.random.set_state(state)
        want, completed = ora.obtain_samples(echo)
        state = np.random.get_state()
        for key, wkey in (("actions", "actions"), ("rewards", "rewards"), ("dones", "dones"), ("env_raw_reward", "raw_reward"),
                          ("env_need_reset", "need_reset"), ("prob", "prob"), ("value", "value"),
                          ("observations", "observations"), ("extra_observations", "extra_observations")):
            np.testing.assert_array_equal(got[key].astype(want[wkey].dtype), want[wkey], err_msg="%s batch %d" % (key, b))
        assert got_t == sorted(ti.as_tuple() for ti in completed)
        n_completed += len(got_t)
    assert n_completed > 0
    smp.shutdown()


def test_forward_parts_describe_the_fold():
    """arl_conv2d_fwd_parts: the partial sums it leaves, folded by arl_fold_many's rule with bias and rectifier, are
    arl_conv2d_fwd's output bit for bit (dense layer at the rollout's shape: 256 x 6912 -> 512, 24 splits)."""
    from accel_rl_amd import _lib
    torch.manual_seed(3)
    b, fan_in, hid = 256, 6912, 512
    x = torch.randn(b, fan_in, device=DEV).relu_()
    w = torch.randn(hid, fan_in, device=DEV) * 0.02
    bias = torch.randn(hid, device=DEV)
    geom = _lib.dense_geom(b, fan_in, hid)
    ws = _lib.conv_workspace(DEV)
    y = torch.empty(b, hid, device=DEV)
    _lib.conv2d_fwd(x, w, bias, y, geom, True, ws)
    y2 = torch.full((b, hid), -7.0, device=DEV)
    item = _lib.conv2d_fwd_parts(x, w, bias, y2, geom, True, ws)
    assert item.splits == 24 and item.total == b * hid and item.part == ws.data_ptr()
    part = ws.view(torch.float32)[:item.splits * item.total].view(item.splits, b, hid)
    # fold_sum (mfma_conv.hip): 16 threads share an output, thread zg sums splits zg, zg + 16, ...; then in index order
    zs = [part[zg] + part[zg + 16] if zg + 16 < item.splits else part[zg].clone() for zg in range(16)]
    s = zs[0]
    for k in range(1, 16):
        s = s + zs[k]
    want = (s + bias).relu_()
    torch.testing.assert_close(want, y, rtol=0, atol=0)
    # a launch that does not split says so and leaves the finished output
    g_small = _lib.dense_geom(4096, 64, 64)
    xs, w_s, ys = torch.randn(4096, 64, device=DEV), torch.randn(64, 64, device=DEV), torch.empty(4096, 64, device=DEV)
    it = _lib.conv2d_fwd_parts(xs, w_s, None, ys, g_small, False, ws)
    ys2 = torch.empty_like(ys)
    _lib.conv2d_fwd(xs, w_s, None, ys2, g_small, False, ws)
    assert it.splits == 0 and it.part == ys.data_ptr()
    torch.testing.assert_close(ys, ys2, rtol=0, atol=0)


def test_rollout_begin_with_conv1_is_the_two_launches():
    """arl_rollout_begin_conv1 = arl_rollout_begin + arl_conv2d_u8_fwd of the copied rows, bit for bit."""
    from accel_rl_amd import _lib
    from accel_rl_amd.policies.atari_cnn_policy import AtariCnnPolicy
    from accel_rl_amd.policies.atari_cnn_specs import cnn_specs
    policy = AtariCnnPolicy(**cnn_specs[1])
    smp = make_sampler("breakout", 3, 7, 5, 5, 9, dict(max_start_noops=30), policy, False, served=True)
    n = 42
    smp.obtain_samples(0)                                   # step_obs: real frames
    torch.cuda.synchronize()
    buf, st = smp.samples_buf, smp._st
    conv1, y1 = policy.serve_conv1(smp._game, n)
    assert conv1 is not None
    buf.observations.fill_(7)
    st.done_count.fill_(5)
    _lib.rollout_begin(smp._game, smp._state, smp._rollout)
    conv_g, _ = policy._layer_geoms(n)
    want_y = torch.empty_like(y1)
    _lib.conv2d_u8_fwd(buf.observations, smp._step_rows[0], policy._scale, policy._w[0], policy._w[1], want_y, conv_g[0], True)
    want_obs = buf.observations.clone()
    buf.observations.fill_(9)
    st.done_count.fill_(5)
    y1.fill_(-1.0)
    _lib.rollout_begin_conv1(smp._game, smp._state, smp._rollout, conv1)
    torch.cuda.synchronize()
    rows = smp._step_rows[0].long()
    assert torch.equal(buf.observations[rows], want_obs[rows]) and torch.equal(buf.observations[rows], smp.step_obs)
    assert int(st.done_count.item()) == 0
    assert torch.equal(y1, want_y) and float(y1.abs().sum()) > 0
    smp.shutdown()


def test_served_step_argument_errors():
    from accel_rl_amd import _lib
    from accel_rl_amd.policies.atari_cnn_policy import AtariCnnPolicy
    from accel_rl_amd.policies.atari_cnn_specs import cnn_specs
    policy = AtariCnnPolicy(**cnn_specs[1])
    smp = make_sampler("breakout", 2, 2, 5, 5, 9, dict(max_start_noops=30), policy, False, served=True)
    n = 8
    head, conv1, y1 = policy.serve_forward(smp._game, smp.samples_buf.observations, smp._step_rows[0], None)
    assert conv1 is not None and y1 is not None
    u = torch.rand(n, dtype=torch.float64, device=DEV)
    bad = _lib.ArlServeHead.from_buffer_copy(head)
    bad.hid = 510
    with pytest.raises(RuntimeError, match="multiple of 4"):
        _lib.env_step_served(smp._game, smp._state, smp._rollout, bad, None, u, 0, 9, 0.99, 30)
    bad = _lib.ArlServeHead.from_buffer_copy(head)
    bad.hidden.total = 17
    with pytest.raises(RuntimeError, match="n_env rows"):
        _lib.env_step_served(smp._game, smp._state, smp._rollout, bad, None, u, 0, 9, 0.99, 30)
    with pytest.raises(RuntimeError, match="horizon"):
        _lib.env_step_served(smp._game, smp._state, smp._rollout, head, None, u, 5, 9, 0.99, 30)
    g = _lib.conv_geom(n, 104, 80, 4, 24, 8, 8, 4, 0, 0)              # 24 filters: not served (32 and 16 are)
    assert _lib.serve_conv1_supported(smp._game, _lib.conv_geom(n, 104, 80, 4, 16, 8, 8, 4, 0, 0))
    assert not _lib.serve_conv1_supported(smp._game, g)
    c = _lib.ArlServeConv1.from_buffer_copy(conv1)
    import ctypes
    c.geom = ctypes.pointer(g)
    with pytest.raises(RuntimeError, match="arl_serve_conv1_supported"):
        _lib.env_step_served(smp._game, smp._state, smp._rollout, head, c, u, 0, 9, 0.99, 30)
    smp.shutdown()
