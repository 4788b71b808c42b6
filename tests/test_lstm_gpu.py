"""Recurrent policy: csrc/lstm.hip and AtariLstmPolicy against a plain-PyTorch restatement
of the reference's FastLstmLayer / PgCnnLstm (policies/layers.py:292-385, pg/networks/pg_cnn_lstm.py):
gate order f, i, c~, o; h0 = c0 = 0; BPTT over each environment's segment from its stored initial
state.  fp32; tolerances stated per check."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TINY = 1e-8


def ref_cell(gx, gh, c_prev):
    h = c_prev.shape[1]
    pre = gx + gh
    f, i, g, o = (pre[:, k * h:(k + 1) * h] for k in range(4))
    f, i, g, o = torch.sigmoid(f), torch.sigmoid(i), torch.tanh(g), torch.sigmoid(o)
    c = f * c_prev + i * g
    return o * torch.tanh(c), c


@pytest.mark.parametrize("batch,hidden,t_len", [(8, 64, 5), (33, 256, 3), (256, 512, 5)])
def test_cell_forward_backward_on_time_slices(batch, hidden, t_len):
    from accel_rl_amd import _lib
    gen = torch.Generator(device=DEV).manual_seed(batch + hidden)
    rnd = lambda *s: torch.randn(*s, device=DEV, generator=gen)                          # noqa: E731
    gx_all, gh = rnd(batch * t_len, 4 * hidden), rnd(batch, 4 * hidden)
    c_prev_all = rnd(batch * t_len, hidden)
    sl = lambda a, t: a.view(batch, t_len, -1)[:, t]                                      # noqa: E731
    t = t_len - 2
    h_all = torch.full((batch * t_len, hidden), float("nan"), device=DEV)
    c_all, gates_all = torch.full_like(h_all, float("nan")), torch.full((batch * t_len, 4 * hidden), float("nan"), device=DEV)
    _lib.lstm_cell_fwd(sl(gx_all, t), gh, sl(c_prev_all, t), sl(h_all, t), sl(c_all, t), sl(gates_all, t))
    gx_r, gh_r = sl(gx_all, t).clone().requires_grad_(), gh.clone().requires_grad_()
    cp_r = sl(c_prev_all, t).clone().requires_grad_()
    h_ref, c_ref = ref_cell(gx_r, gh_r, cp_r)
    assert torch.allclose(sl(h_all, t), h_ref, rtol=1e-5, atol=1e-6) and torch.allclose(sl(c_all, t), c_ref, rtol=1e-5, atol=1e-6)
    assert torch.isnan(sl(h_all, 0)).all()                                                # other time slices untouched
    dh_all, dh_rec, dc_next = rnd(batch * t_len, hidden), rnd(batch, hidden), rnd(batch, hidden)
    dgates_all = torch.full_like(gates_all, float("nan"))
    dc_prev = torch.empty(batch, hidden, device=DEV)
    _lib.lstm_cell_bwd(sl(dh_all, t), dh_rec, dc_next, sl(gates_all, t), sl(c_prev_all, t), sl(c_all, t), sl(dgates_all, t), dc_prev)
    (h_ref * (sl(dh_all, t) + dh_rec)).sum().add((c_ref * dc_next).sum()).backward()
    assert torch.allclose(sl(dgates_all, t), gx_r.grad, rtol=1e-4, atol=1e-6)
    assert torch.allclose(dc_prev, cp_r.grad, rtol=1e-4, atol=1e-6)


def _make(n_act=6, hidden=256):
    from accel_rl_amd.policies.atari_cnn_specs import cnn_specs
    from accel_rl_amd.policies.atari_lstm_policy import AtariLstmPolicy
    from accel_rl_amd.spaces import Discrete, UintBox, EnvSpec
    from accel_rl_amd.util.seed import set_seed
    set_seed(8)
    spec = dict(cnn_specs[0], hidden_sizes=[hidden])
    policy = AtariLstmPolicy(**spec)
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


def test_rollout_step_state_and_reference_layout():
    policy, spec = _make()
    assert policy.recurrent and policy.state_info_keys == ["hprev_0", "cprev_0"]
    assert [tuple(s) for s in policy._ref_shapes[4:7]] == [(3456, 1024), (256, 1024), (1024,)]      # W_x, W_h, b
    flat = policy.get_param_values()
    policy.set_param_values(flat * 2)
    np.testing.assert_array_equal(policy.get_param_values(), flat * 2)
    policy.set_param_values(flat)
    rp = _ref_params(policy)
    rs = np.random.RandomState(0)
    n = 12
    policy.reset(n_batch=n)
    h, c = torch.zeros(n, 256, device=DEV), torch.zeros(n, 256, device=DEV)
    for step in range(3):
        obs = torch.from_numpy(rs.randint(0, 256, size=(n, 4, 104, 80), dtype=np.uint8)).to(DEV)
        pv0 = policy.prob_value(obs)                          # does not advance
        prob, value, hp, cp = policy.act_step(obs)
        assert torch.equal(pv0[0], prob) and torch.equal(pv0[1], value)
        assert torch.allclose(hp, h, atol=1e-6) and torch.allclose(cp, c, atol=1e-6)
        with torch.no_grad():
            xf, k = _ref_features(rp, spec, obs.float() * np.float32(1. / 255))
            h, c = ref_cell(xf @ rp[k] + rp[k + 2], h @ rp[k + 1], c)
            want_p = torch.softmax(h @ rp[k + 3] + rp[k + 4], 1)
            want_v = (h @ rp[k + 5] + rp[k + 6]).reshape(-1)
        assert torch.allclose(prob, want_p, rtol=1e-4, atol=1e-6) and torch.allclose(value, want_v, rtol=1e-4, atol=1e-5)
        if step == 1:                                          # reset envs 3 and 7 (h0 = c0 = 0)
            mask = torch.zeros(n, dtype=torch.uint8, device=DEV)
            mask[[3, 7]] = 1
            policy.reset_rows(mask)
            h[[3, 7]] = 0
            c[[3, 7]] = 0
    assert torch.allclose(policy._state[0], h, rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("masked", [False, True])
def test_bptt_gradients_match_autograd(masked):
    """A2C loss over [8 trajectories x 5 steps] from stored initial states: every parameter gradient
    in the reference's layout vs autograd through the plain-torch network."""
    policy, spec = _make()
    rs = np.random.RandomState(4)
    nb, t_len, hh = 8, 5, 256
    rows = nb * t_len
    obs = torch.from_numpy(rs.randint(0, 256, size=(rows, 4, 104, 80), dtype=np.uint8)).to(DEV)
    act = torch.from_numpy(rs.randint(0, 6, size=rows).astype(np.uint8)).to(DEV)
    adv = torch.from_numpy(rs.randn(rows).astype(np.float32)).to(DEV)
    ret = torch.from_numpy(rs.randn(rows).astype(np.float32)).to(DEV)
    hprev = torch.from_numpy((rs.randn(rows, hh) * 0.3).astype(np.float32)).to(DEV)
    cprev = torch.from_numpy((rs.randn(rows, hh) * 0.3).astype(np.float32)).to(DEV)
    valids = torch.from_numpy((rs.rand(rows) < 0.8).astype(np.int8)).to(DEV) if masked else None
    inv = (1. / valids.sum(dtype=torch.float32)).reshape(1) if masked else None
    lr_mult = torch.ones(1, device=DEV)
    mb = dict(observations=obs, idx=None, actions=act, advantages=adv, returns=ret, valids=valids,
              hprev_0=hprev, cprev_0=cprev, horizon=t_len)
    loss4 = policy.loss_and_grads(mb, 0, 0., 0.25, 0.01, lr_mult, inv).clone()
    got = policy.bucket_to_reference(policy.flat_grads)
    rp = _ref_params(policy)
    xf, k = _ref_features(rp, spec, obs.float() * np.float32(1. / 255))
    gx = (xf @ rp[k] + rp[k + 2]).view(nb, t_len, -1)
    h, c = hprev.view(nb, t_len, hh)[:, 0], cprev.view(nb, t_len, hh)[:, 0]
    hs = []
    for t in range(t_len):
        h, c = ref_cell(gx[:, t], h @ rp[k + 1], c)
        hs.append(h)
    h_all = torch.stack(hs, dim=1).reshape(rows, hh)
    prob = torch.softmax(h_all @ rp[k + 3] + rp[k + 4], 1)
    value = (h_all @ rp[k + 5] + rp[k + 6]).reshape(-1)
    w = (valids.float() * inv) if masked else torch.full((rows,), 1. / rows, device=DEV)
    pa = prob[torch.arange(rows), act.long()]
    pi = -torch.sum(w * torch.log(pa + TINY) * adv)
    vl = 0.25 * torch.sum(w * (value - ret) ** 2)
    el = -0.01 * torch.sum(w * -torch.sum(prob * torch.log(prob + TINY), dim=1))
    grads = torch.autograd.grad(pi + vl + el, rp)
    want = np.concatenate([g.detach().cpu().numpy().reshape(-1) for g in grads])
    assert torch.allclose(loss4[:3], torch.stack([pi, vl, el]).detach(), rtol=1e-4, atol=1e-6)
    scale = np.abs(want).max()
    assert np.allclose(got, want, rtol=2e-3, atol=2e-5 * max(scale, 1e-3)), (np.abs(got - want).max(), scale)


def test_a2c_lstm_trains_through_the_sampler():
    """example_train_a2c.py with the lstm policy: mid_batch_reset=False, valids, hidden state stored per
    step, one rmsprop step per batch; seeded runs agree bit for bit."""
    from accel_rl_amd.algos.pg.a2c import A2C
    from accel_rl_amd.algos.pg.ppo import PPO
    from accel_rl_amd.envs.synthetic_atari import SynthAtariEnv
    from accel_rl_amd.policies.atari_cnn_specs import cnn_specs
    from accel_rl_amd.policies.atari_lstm_policy import AtariLstmPolicy
    from accel_rl_amd.runners.accel_rl import AccelRL
    from accel_rl_amd.sampler.gpu_sampler import GpuVecSampler
    from accel_rl_amd.util import logger
    logger.set_quiet(True)
    finals = []
    for _ in range(2):
        sampler = GpuVecSampler(EnvCls=SynthAtariEnv, env_args=dict(game="pong"), horizon=5, n_parallel=4, envs_per=4,
                                max_path_length=23, mid_batch_reset=False, max_decorrelation_steps=0, device=DEV)
        policy = AtariLstmPolicy(**dict(cnn_specs[0], hidden_sizes=[256]))
        runner = AccelRL(algo=A2C(), policy=policy, sampler=sampler, n_steps=160 * 12, seed=2, log_interval_steps=640)
        runner.train()
        tab = runner.last_tabular
        assert np.isfinite(tab["GradNormAverage"]) and tab["CumCompletedTrajs"] > 0 and tab["LengthAverage"] == 24
        buf = sampler.samples_buf
        hp = buf.agent_infos["hprev_0"].view(32, 5, 256)
        assert hp.abs().sum() > 0 and torch.isfinite(hp).all()
        finals.append(policy.get_param_values())
    np.testing.assert_array_equal(finals[0], finals[1])
    with pytest.raises(NotImplementedError):                 # recurrent + mid-batch reset (aac_base.py:33-34)
        s2 = GpuVecSampler(EnvCls=SynthAtariEnv, env_args=dict(game="pong"), horizon=5, n_parallel=2, envs_per=2,
                           mid_batch_reset=True, max_decorrelation_steps=0, device=DEV)
        AccelRL(algo=A2C(), policy=AtariLstmPolicy(**dict(cnn_specs[0], hidden_sizes=[256])), sampler=s2,
                n_steps=400, seed=1).train()
    with pytest.raises(NotImplementedError):                 # row-minibatch optimizer with a recurrent policy
        s3 = GpuVecSampler(EnvCls=SynthAtariEnv, env_args=dict(game="pong"), horizon=5, n_parallel=2, envs_per=2,
                           mid_batch_reset=False, max_decorrelation_steps=0, device=DEV)
        AccelRL(algo=PPO(), policy=AtariLstmPolicy(**dict(cnn_specs[0], hidden_sizes=[256])), sampler=s3,
                n_steps=400, seed=1).train()
