"""Dueling networks on the device: the merge val + (adv - mean adv) inside csrc/dqn.hip's action / loss kernels
(plain DQN and C51), the stacked hidden layers / block-structured output matrix of QPolicyBase, the reference's
flat parameter order (value branch first), conv-gradient scaling, and an EpsRainbow run -- against plain-PyTorch
restatements of accel_rl/policies/dqn/networks/{dqn_cnn.py:89-112, catdqn_cnn.py:77-93},
policies/dqn/layers/dueling_merge_layer.py:32-35, optimizers/single/dqn_optimizer.py:34-36.  fp32; tolerances per check."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from test_catdqn_gpu import ref_cat_loss
from test_dqn_gpu import ref_q_loss

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def merge(val, adv):
    return val + (adv - adv.mean(dim=1, keepdim=True))


@pytest.mark.parametrize("n_act,batch,double,clip", [(18, 32, True, 1.), (4, 37, False, 1.), (6, 300, True, None)])
def test_dueling_q_loss_and_actions_vs_autograd(n_act, batch, double, clip):
    from accel_rl_amd import _lib
    stride = 32
    gen = torch.Generator(device=DEV).manual_seed(n_act + batch)
    mk = lambda: torch.randn(batch, stride, device=DEV, generator=gen) * 2          # noqa: E731
    q, tgt, pol = mk(), mk(), (mk() if double else None)
    act = torch.randint(0, n_act, (batch,), device=DEV, generator=gen).to(torch.uint8)
    ret = torch.randn(batch, device=DEV, generator=gen)
    term = (torch.rand(batch, device=DEV, generator=gen) < 0.3).to(torch.uint8)
    isw = torch.rand(batch, device=DEV, generator=gen) + 0.1
    gamma_n = float(np.float32(0.99 ** 3))
    dq = torch.full_like(q, float("nan"))
    rows, td = torch.empty(batch, device=DEV), torch.empty(batch, device=DEV)
    _lib.dqn_loss(q, tgt, pol, act, ret, term, isw, n_act, gamma_n, clip, dq, rows, td, dueling=True)
    p = q[:, :n_act + 1].clone().requires_grad_()
    full = lambda t: merge(t[:, n_act:n_act + 1], t[:, :n_act])                     # noqa: E731
    loss, td_ref = ref_q_loss(full(p), full(tgt), None if pol is None else full(pol), act, ret, term, isw, gamma_n, clip)
    loss.backward()
    assert torch.isfinite(dq).all() and not dq[:, n_act + 1:].any()
    assert abs(rows.sum().item() - loss.item()) <= 1e-5 * max(1., abs(loss.item()))
    assert torch.allclose(td, td_ref.detach(), rtol=1e-5, atol=1e-6)
    assert torch.allclose(dq[:, :n_act + 1], p.grad, rtol=1e-5, atol=1e-8)
    onehot, greedy = torch.empty(batch, n_act, device=DEV), torch.empty(batch, dtype=torch.uint8, device=DEV)
    _lib.dqn_act(q, None, n_act, onehot, greedy, dueling=True)
    qq = full(q)
    top2 = torch.topk(qq, 2, dim=1).values
    clear = (top2[:, 0] - top2[:, 1]) > 1e-5
    assert torch.equal(greedy.long()[clear], qq.argmax(dim=1)[clear])
    assert torch.equal(onehot, F.one_hot(greedy.long(), n_act).float())


@pytest.mark.parametrize("n_act,n_atoms,batch,double", [(18, 51, 32, True), (4, 51, 37, False), (3, 64, 5, True)])
def test_dueling_c51_loss_and_actions_vs_autograd(n_act, n_atoms, batch, double):
    from accel_rl_amd import _lib
    stride = (n_atoms + 3) // 4 * 4
    gen = torch.Generator(device=DEV).manual_seed(n_act * 100 + n_atoms)
    mk = lambda: torch.randn(batch, n_act + 1, stride, device=DEV, generator=gen) * 2          # noqa: E731
    pred, tgt, pol = mk(), mk(), (mk() if double else None)
    z = torch.linspace(-10, 10, n_atoms, device=DEV)
    act = torch.randint(0, n_act, (batch,), device=DEV, generator=gen).to(torch.uint8)
    ret = torch.randn(batch, device=DEV, generator=gen) * 6
    term = (torch.rand(batch, device=DEV, generator=gen) < 0.3).to(torch.uint8)
    isw = torch.rand(batch, device=DEV, generator=gen) + 0.1
    gamma_n = float(np.float32(0.99 ** 3))
    dl = torch.full_like(pred, float("nan"))
    rows, kl = torch.empty(batch, device=DEV), torch.empty(batch, device=DEV)
    _lib.catdqn_loss(pred, tgt, pol, z, act, ret, term, isw, n_act, n_atoms, -10., 10., gamma_n, dl, rows, kl,
                     dueling=True)
    full = lambda t: merge(t[:, n_act:, :n_atoms], t[:, :n_act, :n_atoms])          # noqa: E731
    p = pred[:, :, :n_atoms].clone().requires_grad_()
    loss, kl_ref = ref_cat_loss(merge(p[:, n_act:], p[:, :n_act]), full(tgt), None if pol is None else full(pol), z,
                                act, ret, term, isw, -10., 10., gamma_n)
    loss.backward()
    assert torch.isfinite(dl).all() and not dl[:, :, n_atoms:].any()
    assert abs(rows.sum().item() - loss.item()) <= 1e-5 * max(1., abs(loss.item()))
    assert torch.allclose(kl, kl_ref.detach(), rtol=2e-4, atol=2e-6)
    assert torch.allclose(dl[:, :, :n_atoms], p.grad, rtol=2e-4, atol=1e-7)
    onehot, greedy = torch.empty(batch, n_act, device=DEV), torch.empty(batch, dtype=torch.uint8, device=DEV)
    _lib.catdqn_act(pred, z, None, n_act, n_atoms, onehot, greedy, dueling=True)
    qq = (torch.softmax(full(pred), dim=2) * z).sum(dim=2)
    top2 = torch.topk(qq, 2, dim=1).values
    clear = (top2[:, 0] - top2[:, 1]) > 1e-5
    assert torch.equal(greedy.long()[clear], qq.argmax(dim=1)[clear])


def _ref_dueling_out(rp, spec, x, n_act, n_atoms=None):
    """Flat order of the reference: conv..., hidden_Val_0 (W, b), Val (W, b), hidden_0 (W, b), output (W, b)."""
    n_conv = len(spec["conv_filters"])
    k = 0
    for i in range(n_conv):
        x = F.relu(F.conv2d(x, rp[k].flip(2, 3), rp[k + 1], stride=spec["conv_strides"][i],
                            padding=tuple(spec["conv_pads"][i])))
        k += 2
    x = x.flatten(1)
    val = F.relu(x @ rp[k] + rp[k + 1]) @ rp[k + 2] + rp[k + 3]
    adv = F.relu(x @ rp[k + 4] + rp[k + 5]) @ rp[k + 6] + rp[k + 7]
    if n_atoms is None:
        return merge(val, adv)
    return merge(val.view(-1, 1, n_atoms), adv.view(-1, n_act, n_atoms))


def _ref_params(policy, flat_bucket):
    flat = policy.bucket_to_reference(flat_bucket)
    out, pos = [], 0
    for shape in policy._ref_shapes:
        n = int(np.prod(shape))
        out.append(torch.from_numpy(flat[pos:pos + n].reshape(shape).copy()).to(DEV).requires_grad_())
        pos += n
    return out


def _make(kind, n_act=6):
    from accel_rl_amd.policies.atari_cnn_specs import cnn_specs
    from accel_rl_amd.policies.dqn.atari_cat_dqn_policy import AtariCatDqnPolicy
    from accel_rl_amd.policies.dqn.atari_dqn_policy import AtariDqnPolicy
    from accel_rl_amd.spaces import Discrete, UintBox, EnvSpec
    from accel_rl_amd.util.seed import set_seed
    set_seed(5)
    spec = dict(cnn_specs[0])
    policy = (AtariDqnPolicy if kind == "dqn" else AtariCatDqnPolicy)(epsilon=0.3, dueling=True, **spec)
    policy.initialize(EnvSpec(UintBox((4, 104, 80)), Discrete(n_act)), device=DEV)
    if kind == "c51":
        policy.incorporate_z(np.linspace(-10, 10, 51, dtype=np.float32))
    return policy, spec


@pytest.mark.parametrize("kind", ["dqn", "c51"])
def test_dueling_policy_layout_forward_and_training_step(kind):
    policy, spec = _make(kind)
    out_units, val_units = (6, 1) if kind == "dqn" else (6 * 51, 51)
    fan = policy._ref_shapes[-8][0]
    assert [tuple(s) for s in policy._ref_shapes[-8:]] == [(fan, 256), (256,), (256, val_units), (val_units,),
                                                           (fan, 256), (256,), (256, out_units), (out_units,)]
    assert policy.param_short_names[-8:] == ["FCVal0W", "FCVal0b", "ValW", "Valb", "FC0W", "FC0b", "OutputW", "Outputb"]
    flat = policy.get_param_values()
    assert flat.size == policy.n_params == sum(int(np.prod(s)) for s in policy._ref_shapes)
    rs = np.random.RandomState(11)
    flat = flat + (rs.randn(flat.size) * 0.01).astype(np.float32)       # non-zero biases: layout errors would show
    policy.set_param_values(flat)
    np.testing.assert_array_equal(policy.get_param_values(), flat)
    k = policy._k_head
    assert not (policy.params[k].detach() * (1 - policy._duel_mask)).any()     # absent blocks stay exactly zero
    b = 32
    obs = torch.from_numpy(rs.randint(0, 256, size=(b, 4, 104, 80), dtype=np.uint8)).to(DEV)
    nxt = torch.from_numpy(rs.randint(0, 256, size=(b, 4, 104, 80), dtype=np.uint8)).to(DEV)
    act = torch.from_numpy(rs.randint(0, 6, size=b).astype(np.uint8)).to(DEV)
    term = torch.from_numpy((rs.rand(b) < 0.2).astype(np.uint8)).to(DEV)
    isw = torch.from_numpy((rs.rand(b) + 0.2).astype(np.float32)).to(DEV)
    policy.flat_target.copy_(policy.flat_params * 0.9)
    gamma_n = float(np.float32(0.99))
    scale = np.float32(1. / 255)
    rp, rt = _ref_params(policy, policy.flat_params), _ref_params(policy, policy.flat_target)
    n_atoms = None if kind == "dqn" else 51
    pred = _ref_dueling_out(rp, spec, obs.float() * scale, 6, n_atoms)
    with torch.no_grad():
        tgt = _ref_dueling_out(rt, spec, nxt.float() * scale, 6, n_atoms)
        pol = _ref_dueling_out(rp, spec, nxt.float() * scale, 6, n_atoms)
    if kind == "dqn":
        assert torch.allclose(policy.q(obs), pred.detach(), rtol=1e-4, atol=1e-5)
        assert torch.allclose(policy.target_q(nxt), tgt, rtol=1e-4, atol=1e-5)
        ret = torch.from_numpy((rs.randn(b) * 0.05).astype(np.float32)).to(DEV)
        rows, pri = policy.q_loss_and_grads(obs, nxt, act, ret, term, isw, gamma_n, 0.02, double_dqn=True)
        loss, pri_ref = ref_q_loss(pred, tgt, pol, act, ret, term, isw, gamma_n, 0.02)
    else:
        ret = torch.from_numpy(rs.randn(b).astype(np.float32)).to(DEV)
        rows, pri = policy.cat_loss_and_grads(obs, nxt, act, ret, term, isw, -10., 10., gamma_n, double_dqn=True)
        z = torch.linspace(-10, 10, 51, device=DEV)
        loss, pri_ref = ref_cat_loss(pred, tgt, pol, z, act, ret, term, isw, -10., 10., gamma_n)
    got = policy.bucket_to_reference(policy.flat_grads)
    grads = torch.autograd.grad(loss, rp)
    want = np.concatenate([g.detach().cpu().numpy().reshape(-1) for g in grads])
    assert abs(rows.sum().item() - loss.item()) <= 1e-4 * abs(loss.item())
    assert torch.allclose(pri, pri_ref.detach(), rtol=2e-3, atol=1e-5)
    assert np.allclose(got, want, rtol=2e-3, atol=2e-5 * max(np.abs(want).max(), 1e-3)), np.abs(got - want).max()
    assert not (policy.grads[k] * (1 - policy._duel_mask)).any()
    greedy = policy.greedy_actions(obs).cpu().numpy()
    qq = pred.detach() if kind == "dqn" else (torch.softmax(pred.detach(), dim=2) * z).sum(dim=2)
    top2 = torch.topk(qq, 2, dim=1).values
    clear = ((top2[:, 0] - top2[:, 1]) > 1e-5).cpu().numpy()
    np.testing.assert_array_equal(greedy[clear], qq.argmax(dim=1).cpu().numpy()[clear])


def test_conv_gradient_scaling_in_the_optimizer():
    """DqnOptimizer(scale_conv_grads=True): conv gradients x 2^-1/2 before the update; plain SGD-like check via two
    runs of one rmsprop step from identical states."""
    from accel_rl_amd.algos.dqn.dqn import DQN
    from accel_rl_amd.spaces import Discrete, UintBox, EnvSpec
    rs = np.random.RandomState(2)
    b = 32
    batch = (torch.from_numpy(rs.randint(0, 256, size=(b, 4, 104, 80), dtype=np.uint8)).to(DEV),
             torch.from_numpy(rs.randint(0, 256, size=(b, 4, 104, 80), dtype=np.uint8)).to(DEV),
             torch.from_numpy(rs.randint(0, 6, size=b).astype(np.uint8)).to(DEV),
             torch.from_numpy(rs.randn(b).astype(np.float32)).to(DEV),
             torch.from_numpy((rs.rand(b) < 0.2).astype(np.uint8)).to(DEV))
    grads = []
    for scale in (False, True):
        policy, _ = _make("dqn")
        algo = DQN(dueling_dqn=True, double_dqn=True, replay_size=64, batch_size=b,
                   optimizer_args=dict(scale_conv_grads=scale, use_graph=False, grad_norm_clip=None))
        algo.initialize(policy, EnvSpec(UintBox((4, 104, 80)), Discrete(6)), sample_size=64, horizon=4,
                        mid_batch_reset=True)
        algo.optimizer.optimize(batch)
        grads.append(policy.flat_grads.clone())
    split = policy.grad_split_offset
    assert torch.equal(grads[1][split:], grads[0][split:])
    assert torch.allclose(grads[1][:split], grads[0][:split] * float(np.float32(2 ** -0.5)), rtol=1e-6, atol=0)
    assert grads[0][:split].abs().max() > 0


def test_eps_rainbow_trains_and_reproduces():
    from accel_rl_amd.algos.dqn.eps_rainbow import EpsRainbow
    from accel_rl_amd.envs.synthetic_atari import SynthAtariEnv
    from accel_rl_amd.policies.atari_cnn_specs import cnn_specs
    from accel_rl_amd.policies.dqn.atari_cat_dqn_policy import AtariCatDqnPolicy
    from accel_rl_amd.runners.accel_rl import AccelRLEval
    from accel_rl_amd.sampler.gpu_sampler_with_eval import GpuVecEvalSampler
    from accel_rl_amd.util import logger
    logger.set_quiet(True)
    a = EpsRainbow()
    assert (a.reward_horizon, a.double_dqn, a.dueling_dqn, a.prioritized_replay, a.target_update_steps,
            a.min_steps_learn) == (3, True, True, True, 8000, 20000)
    o = a.optimizer
    assert (o._learning_rate, o._grad_norm_clip, o._scale_conv_grads, o._update_method.name) == (6.25e-5, 10, True, "adam")
    finals = []
    for _ in range(2):
        sampler = GpuVecEvalSampler(eval_steps=8 * 40, eval_envs_per=1, EnvCls=SynthAtariEnv,
                                    env_args=dict(game="seaquest"), horizon=4, n_parallel=4, envs_per=2,
                                    max_path_length=25, max_decorrelation_steps=0, device=DEV)
        algo = EpsRainbow(batch_size=32, min_steps_learn=64 * 4, replay_size=64 * 60, training_intensity=8,
                          target_update_steps=64 * 3, eps_greedy_args=dict(anneal_steps=64 * 10))
        policy = AtariCatDqnPolicy(dueling=True, **cnn_specs[0])
        runner = AccelRLEval(algo=algo, policy=policy, sampler=sampler, n_steps=64 * 24, seed=9,
                             eval_interval_steps=64 * 8)
        runner.train()
        tab = runner.last_tabular
        assert np.isfinite(tab["LossAverage"]) and tab["LossAverage"] > 0 and tab["TrajsInEval"] > 0
        k = policy._k_head
        assert not (policy.params[k].detach() * (1 - policy._duel_mask)).any()
        finals.append(policy.get_param_values())
    np.testing.assert_array_equal(finals[0], finals[1])
    with pytest.raises(AssertionError, match="dueling"):
        from accel_rl_amd.spaces import Discrete, UintBox, EnvSpec
        p = AtariCatDqnPolicy(**cnn_specs[0])
        p.initialize(EnvSpec(UintBox((4, 104, 80)), Discrete(6)), device=DEV)
        EpsRainbow(replay_size=64).initialize(p, p.env_spec, sample_size=64, horizon=4, mid_batch_reset=True)
