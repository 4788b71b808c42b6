"""Plain (double) DQN on the device (csrc/dqn.hip: arl_dqn_act / arl_dqn_loss, accel_rl_amd/policies/dqn/
atari_dqn_policy.py, accel_rl_amd/algos/dqn/dqn.py) against a plain-PyTorch restatement of the reference's
Theano graph (accel_rl/algos/dqn/dqn.py:137-172, policies/dqn/atari_dqn_policy.py:76-106) and an end-to-end
training run with prioritized replay and offline evaluation.  Floating point: fp32, tolerances per check."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from test_catdqn_gpu import _ref_logits

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def ref_q_loss(q, tgt_q, pol_next_q, act, ret, term, isw, gamma_n, delta_clip):
    """dqn.py:146-165 in plain torch."""
    rows = torch.arange(len(act))
    if pol_next_q is not None:
        next_q = tgt_q[rows, torch.argmax(pol_next_q, dim=1)]
    else:
        next_q = tgt_q.max(dim=1).values
    y = ret + (1 - term.float()) * (gamma_n * next_q)
    d = y - q[rows, act.long()]
    losses = 0.5 * d ** 2
    if delta_clip is not None:
        losses = torch.where(d.abs() <= delta_clip, losses, delta_clip * (d.abs() - delta_clip / 2))
    if isw is not None:
        losses = isw * losses
    td = d.abs() if delta_clip is None else torch.clamp(d.abs(), 0, delta_clip)
    return losses.mean(), td


@pytest.mark.parametrize("n_act,batch,double,weighted,clip", [(18, 32, False, True, 1.), (4, 37, True, False, 1.),
                                                              (6, 512, False, False, None), (3, 5, True, True, 0.5),
                                                              (9, 1000, True, True, 2.)])
def test_q_loss_and_gradient_vs_autograd(n_act, batch, double, weighted, clip):
    from accel_rl_amd import _lib
    stride = 32
    gen = torch.Generator(device=DEV).manual_seed(n_act * 100 + batch)
    mk = lambda: torch.randn(batch, stride, device=DEV, generator=gen) * 2          # noqa: E731
    q, tgt, pol = mk(), mk(), (mk() if double else None)
    tgt[:, n_act:] = 1e9                                        # the padding must never be looked at
    act = torch.randint(0, n_act, (batch,), device=DEV, generator=gen).to(torch.uint8)
    ret = torch.randn(batch, device=DEV, generator=gen)
    term = (torch.rand(batch, device=DEV, generator=gen) < 0.3).to(torch.uint8)
    isw = torch.rand(batch, device=DEV, generator=gen) + 0.1 if weighted else None
    gamma_n = float(np.float32(0.99 ** 3))
    dq = torch.full_like(q, float("nan"))
    rows, td = torch.empty(batch, device=DEV), torch.empty(batch, device=DEV)
    _lib.dqn_loss(q, tgt, pol, act, ret, term, isw, n_act, gamma_n, clip, dq, rows, td)
    p = q[:, :n_act].clone().requires_grad_()
    loss, td_ref = ref_q_loss(p, tgt[:, :n_act], None if pol is None else pol[:, :n_act], act, ret, term, isw,
                              gamma_n, clip)
    loss.backward()
    assert torch.isfinite(dq).all() and not dq[:, n_act:].any()
    assert abs(rows.sum().item() - loss.item()) <= 1e-5 * max(1., abs(loss.item()))
    assert torch.allclose(td, td_ref.detach(), rtol=1e-5, atol=1e-6)
    assert torch.allclose(dq[:, :n_act], p.grad, rtol=1e-5, atol=1e-8)


def test_q_action_kernel_greedy_and_override():
    from accel_rl_amd import _lib
    gen = torch.Generator(device=DEV).manual_seed(4)
    b, a, s = 300, 18, 32
    q = torch.randn(b, s, device=DEV, generator=gen)
    q[:, a:] = 1e9
    q[7, :a] = 0.                                               # all-equal Q: first maximum wins (T.argmax)
    q[9, 3] = q[9, 11] = 50.
    ov = torch.full((b,), -1, dtype=torch.int32, device=DEV)
    ov[::5] = torch.randint(0, a, (len(ov[::5]),), device=DEV, generator=gen).to(torch.int32)
    onehot, greedy = torch.empty(b, a, device=DEV), torch.empty(b, dtype=torch.uint8, device=DEV)
    _lib.dqn_act(q, ov, a, onehot, greedy)
    want = torch.argmax(q[:, :a], dim=1)
    want[7], want[9] = 0, 3
    assert torch.equal(greedy.long(), want)
    chosen = torch.where(ov >= 0, ov.long(), greedy.long())
    assert torch.equal(onehot, F.one_hot(chosen, a).float())
    with pytest.raises(RuntimeError, match="q_stride"):
        _lib.dqn_act(q.view(-1)[:b * 16].view(b, 16), None, a, onehot)


def _make_policy(n_act=6, eps=0.3, **kw):
    from accel_rl_amd.policies.atari_cnn_specs import cnn_specs
    from accel_rl_amd.policies.dqn.atari_dqn_policy import AtariDqnPolicy
    from accel_rl_amd.spaces import Discrete, UintBox, EnvSpec
    from accel_rl_amd.util.seed import set_seed
    set_seed(5)
    spec = dict(cnn_specs[0])
    policy = AtariDqnPolicy(epsilon=eps, **spec, **kw)
    policy.initialize(EnvSpec(UintBox((4, 104, 80)), Discrete(n_act)), device=DEV)
    return policy, spec


def _ref_params(policy, flat_bucket):
    flat = policy.bucket_to_reference(flat_bucket)
    out, pos = [], 0
    for shape in policy._ref_shapes:
        n = int(np.prod(shape))
        out.append(torch.from_numpy(flat[pos:pos + n].reshape(shape).copy()).to(DEV).requires_grad_())
        pos += n
    return out


def test_policy_forward_layout_and_epsilon_greedy_stream():
    policy, spec = _make_policy()
    flat = policy.get_param_values()
    assert flat.size == policy.n_params and policy._ref_shapes[-2] == (256, 6) and policy._ref_shapes[-1] == (6,)
    policy.set_param_values(flat * 1.5)
    np.testing.assert_array_equal(policy.get_param_values(), flat * np.float32(1.5))
    policy.set_param_values(flat)
    rp = _ref_params(policy, policy.flat_params)
    rs = np.random.RandomState(1)
    obs = torch.from_numpy(rs.randint(0, 256, size=(24, 4, 104, 80), dtype=np.uint8)).to(DEV)
    with torch.no_grad():
        want = _ref_logits(rp, spec, obs.float() * np.float32(1. / 255))
    out, _, _ = policy._logits(policy._scaled(obs))
    assert out.shape == (24, 32) and not out[:, 6:].any()
    assert torch.allclose(policy.q(obs), want, rtol=1e-4, atol=1e-6)
    policy.flat_target.mul_(0.5)
    assert not torch.allclose(policy.target_q(obs), policy.q(obs))
    policy.update_target()
    assert torch.equal(policy.target_q(obs), policy.q(obs))
    # epsilon-greedy: whole-rollout draws == the reference's per-(step, group) loop on the same seed
    greedy = policy.greedy_actions(obs).cpu().numpy()
    np.testing.assert_array_equal(greedy, policy.q(obs).argmax(dim=1).cpu().numpy())
    np.random.seed(77)
    policy.host_draws(1, 24)
    policy.set_step(0)
    onehot, value = policy.prob_value(obs)
    served = onehot.argmax(dim=1).cpu().numpy()
    np.random.seed(77)
    want_acts = greedy.copy()
    for j in (0, 1):                                            # two alternating groups of 12
        acts = want_acts[j * 12:(j + 1) * 12]
        idx = np.where(np.random.rand(12) < 0.3)[0]
        acts[idx] = np.random.randint(low=0, high=6, size=len(idx), dtype=np.uint8)
    np.testing.assert_array_equal(served, want_acts)
    assert not value.any()


@pytest.mark.parametrize("double", [False, True])
def test_training_step_matches_autograd_through_plain_torch(double):
    """One DQN minibatch: gradients of every parameter in the reference's layout."""
    policy, spec = _make_policy()
    rs = np.random.RandomState(3)
    b = 32
    obs = torch.from_numpy(rs.randint(0, 256, size=(b, 4, 104, 80), dtype=np.uint8)).to(DEV)
    nxt = torch.from_numpy(rs.randint(0, 256, size=(b, 4, 104, 80), dtype=np.uint8)).to(DEV)
    act = torch.from_numpy(rs.randint(0, 6, size=b).astype(np.uint8)).to(DEV)
    ret = torch.from_numpy((rs.randn(b) * 0.02).astype(np.float32)).to(DEV)     # Q values start near 0.01-scale
    term = torch.from_numpy((rs.rand(b) < 0.2).astype(np.uint8)).to(DEV)
    isw = torch.from_numpy((rs.rand(b) + 0.2).astype(np.float32)).to(DEV)
    policy.flat_target.copy_(policy.flat_params * 0.9)          # a target net that differs
    gamma_n = float(np.float32(0.99))
    rows, td = policy.q_loss_and_grads(obs, nxt, act, ret, term, isw, gamma_n, 0.01, double_dqn=double)
    got = policy.bucket_to_reference(policy.flat_grads)
    rp, rt = _ref_params(policy, policy.flat_params), _ref_params(policy, policy.flat_target)
    scale = np.float32(1. / 255)
    q = _ref_logits(rp, spec, obs.float() * scale)
    with torch.no_grad():
        tgt = _ref_logits(rt, spec, nxt.float() * scale)
        pol = _ref_logits(rp, spec, nxt.float() * scale) if double else None
    loss, td_ref = ref_q_loss(q, tgt, pol, act, ret, term, isw, gamma_n, 0.01)
    grads = torch.autograd.grad(loss, rp)
    want = np.concatenate([g.detach().cpu().numpy().reshape(-1) for g in grads])
    assert (td_ref < 0.01).any() and (td_ref >= 0.01).any()     # both branches of the Huber loss in play
    assert abs(rows.sum().item() - loss.item()) <= 1e-4 * abs(loss.item())
    assert torch.allclose(td, td_ref, rtol=2e-3, atol=1e-6)
    assert np.allclose(got, want, rtol=2e-3, atol=2e-5 * max(np.abs(want).max(), 1e-3)), np.abs(got - want).max()


def test_dqn_trains_with_prioritized_replay_and_eval():
    """GpuVecEvalSampler -> device replay (prioritized) -> double-DQN Huber updates -> target sync, epsilon / beta
    schedules, AccelRLEval logging; two seeded runs agree bit for bit."""
    from accel_rl_amd.algos.dqn.dqn import DQN
    from accel_rl_amd.envs.synthetic_atari import SynthAtariEnv
    from accel_rl_amd.policies.atari_cnn_specs import cnn_specs
    from accel_rl_amd.policies.dqn.atari_dqn_policy import AtariDqnPolicy
    from accel_rl_amd.runners.accel_rl import AccelRLEval
    from accel_rl_amd.sampler.gpu_sampler_with_eval import GpuVecEvalSampler
    from accel_rl_amd.util import logger
    logger.set_quiet(True)
    finals = []
    for _ in range(2):
        sampler = GpuVecEvalSampler(eval_steps=8 * 40, eval_envs_per=1, EnvCls=SynthAtariEnv,
                                    env_args=dict(game="seaquest"), horizon=4, n_parallel=4, envs_per=2,
                                    max_path_length=25, max_decorrelation_steps=0, device=DEV)
        algo = DQN(batch_size=32, min_steps_learn=64 * 4, replay_size=64 * 60, training_intensity=8,
                   target_update_steps=64 * 3, reward_horizon=3, prioritized_replay=True, double_dqn=True,
                   eps_greedy_args=dict(anneal_steps=64 * 10))
        policy = AtariDqnPolicy(**cnn_specs[0])
        runner = AccelRLEval(algo=algo, policy=policy, sampler=sampler, n_steps=64 * 24, seed=9,
                             eval_interval_steps=64 * 8)
        runner.train()
        tab = runner.last_tabular
        for key in ("StepsInEval", "TrajsInEval", "LossAverage", "PriorityAverage", "ReturnAverage", "ParamsNorm"):
            assert key in tab, key
        assert np.isfinite(tab["LossAverage"]) and tab["LossAverage"] > 0 and tab["TrajsInEval"] > 0
        assert 0 < tab["PriorityAverage"] <= 1.0                # |TD error| clipped to delta_clip = 1
        assert algo._updates_per_optimize == 8 * 64 // 32 and abs(policy.get_epsilon() - 0.1) < 1e-9
        assert algo.replay_buffer.beta > 0.4
        finals.append(policy.get_param_values())
    np.testing.assert_array_equal(finals[0], finals[1])


def test_shared_last_bias_is_one_scalar_parameter():
    """`shared_last_bias=True` (dqn_cnn.py:67-82: output layer without a bias + a BiasLayer with shared_axes=(0, 1)): the
    reference-layout vector has ONE bias element after the output weights; forward, every gradient of a training step (the
    scalar's = the sum over the actions) against autograd through plain PyTorch; an adam step keeps the stored per-action
    entries equal."""
    policy, spec = _make_policy(shared_last_bias=True)
    plain, _ = _make_policy()
    assert policy._ref_shapes[-2] == (256, 6) and policy._ref_shapes[-1] == (1,) and policy.n_params == plain.n_params - 5
    flat = policy.get_param_values()
    flat[-1] = 0.37
    policy.set_param_values(flat)
    np.testing.assert_array_equal(policy.get_param_values(), flat)
    rs = np.random.RandomState(3)
    b = 32
    obs = torch.from_numpy(rs.randint(0, 256, size=(b, 4, 104, 80), dtype=np.uint8)).to(DEV)
    nxt = torch.from_numpy(rs.randint(0, 256, size=(b, 4, 104, 80), dtype=np.uint8)).to(DEV)
    act = torch.from_numpy(rs.randint(0, 6, size=b).astype(np.uint8)).to(DEV)
    ret = torch.from_numpy((rs.randn(b) * 0.02 + 0.37).astype(np.float32)).to(DEV)
    term = torch.from_numpy((rs.rand(b) < 0.2).astype(np.uint8)).to(DEV)
    isw = torch.from_numpy((rs.rand(b) + 0.2).astype(np.float32)).to(DEV)
    rp = _ref_params(policy, policy.flat_params)
    scale = np.float32(1. / 255)
    with torch.no_grad():
        want_q = _ref_logits(rp, spec, obs.float() * scale)
    assert torch.allclose(policy.q(obs), want_q, rtol=1e-4, atol=1e-6)
    policy.flat_target.copy_(policy.flat_params * 0.9)
    gamma_n = float(np.float32(0.99))
    rows, td = policy.q_loss_and_grads(obs, nxt, act, ret, term, isw, gamma_n, 0.01, double_dqn=True)
    got = policy.bucket_to_reference(policy.flat_grads)
    rt = _ref_params(policy, policy.flat_target)
    q = _ref_logits(rp, spec, obs.float() * scale)
    with torch.no_grad():
        tgt = _ref_logits(rt, spec, nxt.float() * scale)
        pol = _ref_logits(rp, spec, nxt.float() * scale)
    loss, _ = ref_q_loss(q, tgt, pol, act, ret, term, isw, gamma_n, 0.01)
    grads = torch.autograd.grad(loss, rp)
    want = np.concatenate([g.detach().cpu().numpy().reshape(-1) for g in grads])
    assert got.size == want.size and abs(want[-1]) > 0
    assert np.allclose(got, want, rtol=2e-3, atol=2e-5 * max(np.abs(want).max(), 1e-3)), np.abs(got - want).max()
    gb = policy.grads[policy._k_head + 1]
    assert torch.equal(gb[:6], gb[:1].expand(6)) and not gb[6:].any()       # every action's entry holds the sum
    with torch.no_grad():
        bias = policy.params[policy._k_head + 1]
        bias.sub_(0.1 * gb)                                                   # any elementwise update keeps them equal
        assert torch.equal(bias[:6], bias[:1].expand(6))
