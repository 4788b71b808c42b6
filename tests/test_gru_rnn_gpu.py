"""GRU and plain-RNN policies: csrc/gru.hip, AtariGruPolicy, AtariRnnPolicy against a
plain-PyTorch restatement of the reference's GruLayer / RecurrentLayer steps
(policies/layers.py:163-168, 80-82; pg/networks/pg_cnn_gru.py, pg_cnn_rnn.py): h0 = 0, state key
hprev_0, BPTT over each environment's segment from its stored initial state.  fp32; tolerances
stated per check."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TINY = 1e-8


def gru_cell(gx, gh, h_prev):
    h = h_prev.shape[1]
    r = torch.sigmoid(gx[:, :h] + gh[:, :h])
    u = torch.sigmoid(gx[:, h:2 * h] + gh[:, h:2 * h])
    c = torch.tanh(gx[:, 2 * h:] + r * gh[:, 2 * h:])
    return (1 - u) * h_prev + u * c


def rnn_cell(gx, gh, h_prev):
    return torch.tanh(gx + gh)


@pytest.mark.parametrize("batch,hidden,t_len", [(8, 64, 5), (33, 256, 3), (256, 512, 5)])
def test_gru_cell_forward_backward_on_time_slices(batch, hidden, t_len):
    from accel_rl_amd import _lib
    gen = torch.Generator(device=DEV).manual_seed(batch + hidden)
    rnd = lambda *s: torch.randn(*s, device=DEV, generator=gen)                          # noqa: E731
    nan = lambda *s: torch.full(s, float("nan"), device=DEV)                             # noqa: E731
    gx_all, gh, hp_all = rnd(batch * t_len, 3 * hidden), rnd(batch, 3 * hidden), rnd(batch * t_len, hidden)
    sl = lambda a, t: a.view(batch, t_len, -1)[:, t]                                      # noqa: E731
    t = t_len - 2
    h_all, saved_all = nan(batch * t_len, hidden), nan(batch * t_len, 4 * hidden)
    _lib.gru_cell_fwd(sl(gx_all, t), gh, sl(hp_all, t), sl(h_all, t), sl(saved_all, t))
    gx_r, gh_r = sl(gx_all, t).clone().requires_grad_(), gh.clone().requires_grad_()
    hp_r = sl(hp_all, t).clone().requires_grad_()
    h_ref = gru_cell(gx_r, gh_r, hp_r)
    assert torch.allclose(sl(h_all, t), h_ref, rtol=1e-5, atol=1e-6)
    assert torch.isnan(sl(h_all, 0)).all() and torch.isnan(sl(saved_all, 0)).all()       # other time slices untouched
    dh_all, dh_rec, dh_dir = rnd(batch * t_len, hidden), rnd(batch, hidden), rnd(batch, hidden)
    dgx_all, dgh_all = nan(batch * t_len, 3 * hidden), nan(batch * t_len, 3 * hidden)
    dh_prev = torch.empty(batch, hidden, device=DEV)
    _lib.gru_cell_bwd(sl(dh_all, t), dh_rec, dh_dir, sl(saved_all, t), sl(hp_all, t), sl(dgx_all, t), sl(dgh_all, t),
                      dh_prev)
    dh = sl(dh_all, t) + dh_rec + dh_dir
    (h_ref * dh).sum().backward()
    assert torch.allclose(sl(dgx_all, t), gx_r.grad, rtol=1e-4, atol=1e-6)
    assert torch.allclose(sl(dgh_all, t), gh_r.grad, rtol=1e-4, atol=1e-6)
    u = sl(saved_all, t)[:, hidden:2 * hidden]
    assert torch.allclose(dh_prev, dh * (1 - u), rtol=1e-5, atol=1e-6)                    # direct part only
    # without the optional inputs
    _lib.gru_cell_bwd(None, dh_rec, None, sl(saved_all, t), sl(hp_all, t), sl(dgx_all, t), sl(dgh_all, t), dh_prev)
    assert torch.allclose(dh_prev, dh_rec * (1 - u), rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("batch,hidden,t_len", [(8, 64, 5), (33, 256, 3)])
def test_rnn_cell_forward_backward_on_time_slices(batch, hidden, t_len):
    from accel_rl_amd import _lib
    gen = torch.Generator(device=DEV).manual_seed(batch)
    rnd = lambda *s: torch.randn(*s, device=DEV, generator=gen)                          # noqa: E731
    gx_all, gh = rnd(batch * t_len, hidden), rnd(batch, hidden)
    sl = lambda a, t: a.view(batch, t_len, -1)[:, t]                                      # noqa: E731
    t = 1
    h_all = torch.full((batch * t_len, hidden), float("nan"), device=DEV)
    _lib.rnn_cell_fwd(sl(gx_all, t), gh, sl(h_all, t))
    want = torch.tanh(sl(gx_all, t) + gh)
    assert torch.allclose(sl(h_all, t), want, rtol=1e-5, atol=1e-6) and torch.isnan(sl(h_all, 0)).all()
    dh_all, dh_rec = rnd(batch * t_len, hidden), rnd(batch, hidden)
    dpre_all = torch.full_like(h_all, float("nan"))
    _lib.rnn_cell_bwd(sl(dh_all, t), dh_rec, sl(h_all, t), sl(dpre_all, t))
    assert torch.allclose(sl(dpre_all, t), (sl(dh_all, t) + dh_rec) * (1 - want * want), rtol=1e-4, atol=1e-6)
    _lib.rnn_cell_bwd(sl(dh_all, t), None, sl(h_all, t), sl(dpre_all, t))
    assert torch.allclose(sl(dpre_all, t), sl(dh_all, t) * (1 - want * want), rtol=1e-4, atol=1e-6)


def _policy_cls(kind):
    if kind == "gru":
        from accel_rl_amd.policies.atari_gru_policy import AtariGruPolicy
        return AtariGruPolicy
    from accel_rl_amd.policies.atari_rnn_policy import AtariRnnPolicy
    return AtariRnnPolicy


def _make(kind, n_act=6, hidden=256):
    from accel_rl_amd.policies.atari_cnn_specs import cnn_specs
    from accel_rl_amd.spaces import Discrete, UintBox, EnvSpec
    from accel_rl_amd.util.seed import set_seed
    set_seed(8)
    spec = dict(cnn_specs[0], hidden_sizes=[hidden])
    policy = _policy_cls(kind)(**spec)
    policy.initialize(EnvSpec(UintBox((4, 104, 80)), Discrete(n_act)), device=DEV)
    return policy, spec


def _ref_params(policy):
    flat = policy.get_param_values()
    out, pos = [], 0
    for shape in policy._ref_shapes:
        n = int(np.prod(shape))
        out.append(torch.from_numpy(flat[pos:pos + n].reshape(shape).copy()).to(DEV).requires_grad_())
        pos += n
    assert pos == flat.size == policy.n_params
    return out


def _ref_features(rp, spec, x):
    k = 0
    for i in range(len(spec["conv_filters"])):
        x = F.relu(F.conv2d(x, rp[k].flip(2, 3), rp[k + 1], stride=spec["conv_strides"][i], padding=tuple(spec["conv_pads"][i])))
        k += 2
    return x.flatten(1), k


def _ref_step(kind, rp, k, xf, h):
    """One step in the reference's own parameter layout; returns (h', index of W_pi)."""
    if kind == "rnn":
        return torch.tanh(xf @ rp[k] + h @ rp[k + 1] + rp[k + 2]), k + 3
    k += 3                                              # W_xh, W_hh, b: registered, never read
    r = torch.sigmoid(xf @ rp[k] + h @ rp[k + 1] + rp[k + 2])
    u = torch.sigmoid(xf @ rp[k + 3] + h @ rp[k + 4] + rp[k + 5])
    c = torch.tanh(xf @ rp[k + 6] + r * (h @ rp[k + 7]) + rp[k + 8])
    return (1 - u) * h + u * c, k + 9


@pytest.mark.parametrize("kind", ["gru", "rnn"])
def test_rollout_step_state_and_reference_layout(kind):
    policy, spec = _make(kind)
    assert policy.recurrent and policy.state_info_keys == ["hprev_0"]
    want_shapes = [(3456, 256), (256, 256), (256,)] * (4 if kind == "gru" else 1)
    assert [tuple(s) for s in policy._ref_shapes[4:4 + len(want_shapes)]] == want_shapes
    flat = policy.get_param_values()
    rs = np.random.RandomState(1)
    other = rs.randn(flat.size).astype(np.float32)      # every slot distinct: the layout maps are bijections
    policy.set_param_values(other)
    np.testing.assert_array_equal(policy.get_param_values(), other)
    policy.set_param_values(flat)
    rp = _ref_params(policy)
    n = 12
    policy.reset(n_batch=n)
    h = torch.zeros(n, 256, device=DEV)
    for step in range(3):
        obs = torch.from_numpy(rs.randint(0, 256, size=(n, 4, 104, 80), dtype=np.uint8)).to(DEV)
        pv0 = policy.prob_value(obs)                          # does not advance
        prob, value, hp = policy.act_step(obs)
        assert torch.equal(pv0[0], prob) and torch.equal(pv0[1], value)
        assert torch.allclose(hp, h, atol=1e-6)
        with torch.no_grad():
            xf, k = _ref_features(rp, spec, obs.float() * np.float32(1. / 255))
            h, kp = _ref_step(kind, rp, k, xf, h)
            want_p = torch.softmax(h @ rp[kp] + rp[kp + 1], 1)
            want_v = (h @ rp[kp + 2] + rp[kp + 3]).reshape(-1)
        assert torch.allclose(prob, want_p, rtol=1e-4, atol=1e-6) and torch.allclose(value, want_v, rtol=1e-4, atol=1e-5)
        if step == 1:
            mask = torch.zeros(n, dtype=torch.uint8, device=DEV)
            mask[[3, 7]] = 1
            policy.reset_rows(mask)
            h[[3, 7]] = 0
    assert torch.allclose(policy.get_prev_hiddens()[0], h, rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("kind,masked", [("gru", False), ("gru", True), ("rnn", False), ("rnn", True)])
def test_bptt_gradients_match_autograd(kind, masked):
    """A2C loss over [8 trajectories x 5 steps] from stored initial states: every parameter gradient
    in the reference's layout vs autograd through the plain-torch network (the GRU's three unread
    tensors get exactly zero)."""
    policy, spec = _make(kind)
    rs = np.random.RandomState(4)
    nb, t_len, hh = 8, 5, 256
    rows = nb * t_len
    obs = torch.from_numpy(rs.randint(0, 256, size=(rows, 4, 104, 80), dtype=np.uint8)).to(DEV)
    act = torch.from_numpy(rs.randint(0, 6, size=rows).astype(np.uint8)).to(DEV)
    adv = torch.from_numpy(rs.randn(rows).astype(np.float32)).to(DEV)
    ret = torch.from_numpy(rs.randn(rows).astype(np.float32)).to(DEV)
    hprev = torch.from_numpy((rs.randn(rows, hh) * 0.3).astype(np.float32)).to(DEV)
    valids = torch.from_numpy((rs.rand(rows) < 0.8).astype(np.int8)).to(DEV) if masked else None
    inv = (1. / valids.sum(dtype=torch.float32)).reshape(1) if masked else None
    lr_mult = torch.ones(1, device=DEV)
    mb = dict(observations=obs, idx=None, actions=act, advantages=adv, returns=ret, valids=valids,
              hprev_0=hprev, horizon=t_len)
    loss4 = policy.loss_and_grads(mb, 0, 0., 0.25, 0.01, lr_mult, inv).clone()
    got = policy.bucket_to_reference(policy.flat_grads)
    rp = _ref_params(policy)
    xf, k = _ref_features(rp, spec, obs.float() * np.float32(1. / 255))
    xf = xf.view(nb, t_len, -1)
    h = hprev.view(nb, t_len, hh)[:, 0]
    hs = []
    for t in range(t_len):
        h, kp = _ref_step(kind, rp, k, xf[:, t], h)
        hs.append(h)
    h_all = torch.stack(hs, dim=1).reshape(rows, hh)
    prob = torch.softmax(h_all @ rp[kp] + rp[kp + 1], 1)
    value = (h_all @ rp[kp + 2] + rp[kp + 3]).reshape(-1)
    w = (valids.float() * inv) if masked else torch.full((rows,), 1. / rows, device=DEV)
    pa = prob[torch.arange(rows), act.long()]
    pi = -torch.sum(w * torch.log(pa + TINY) * adv)
    vl = 0.25 * torch.sum(w * (value - ret) ** 2)
    el = -0.01 * torch.sum(w * -torch.sum(prob * torch.log(prob + TINY), dim=1))
    grads = torch.autograd.grad(pi + vl + el, rp, allow_unused=True)
    grads = [torch.zeros_like(p) if g is None else g for g, p in zip(grads, rp)]
    want = np.concatenate([g.detach().cpu().numpy().reshape(-1) for g in grads])
    assert torch.allclose(loss4[:3], torch.stack([pi, vl, el]).detach(), rtol=1e-4, atol=1e-6)
    scale = np.abs(want).max()
    assert np.allclose(got, want, rtol=2e-3, atol=2e-5 * max(scale, 1e-3)), (np.abs(got - want).max(), scale)
    if kind == "gru":
        n_dead = 3456 * 256 + 256 * 256 + 256
        pos = sum(int(np.prod(s)) for s in policy._ref_shapes[:4])
        assert not got[pos:pos + n_dead].any()


@pytest.mark.parametrize("kind", ["gru", "rnn"])
def test_a2c_trains_through_the_sampler(kind):
    """example_train_a2c.py with the gru / rnn policy: mid_batch_reset=False, valids, hidden state
    stored per step, one rmsprop step per batch; seeded runs agree bit for bit."""
    from accel_rl_amd.algos.pg.a2c import A2C
    from accel_rl_amd.envs.synthetic_atari import SynthAtariEnv
    from accel_rl_amd.policies.atari_cnn_specs import cnn_specs
    from accel_rl_amd.runners.accel_rl import AccelRL
    from accel_rl_amd.sampler.gpu_sampler import GpuVecSampler
    from accel_rl_amd.util import logger
    logger.set_quiet(True)
    finals = []
    for _ in range(2):
        sampler = GpuVecSampler(EnvCls=SynthAtariEnv, env_args=dict(game="pong"), horizon=5, n_parallel=4, envs_per=4,
                                max_path_length=23, mid_batch_reset=False, max_decorrelation_steps=0, device=DEV)
        policy = _policy_cls(kind)(**dict(cnn_specs[0], hidden_sizes=[256]))
        runner = AccelRL(algo=A2C(), policy=policy, sampler=sampler, n_steps=160 * 12, seed=2, log_interval_steps=640)
        runner.train()
        tab = runner.last_tabular
        assert np.isfinite(tab["GradNormAverage"]) and tab["CumCompletedTrajs"] > 0 and tab["LengthAverage"] == 24
        hp = sampler.samples_buf.agent_infos["hprev_0"].view(32, 5, 256)
        assert hp.abs().sum() > 0 and torch.isfinite(hp).all()
        assert list(sampler.samples_buf.agent_infos.keys()).count("cprev_0") == 0
        finals.append(policy.get_param_values())
    np.testing.assert_array_equal(finals[0], finals[1])
