"""TEST INFRASTRUCTURE: the autograd formulation of the policy network through PyTorch's own conv2d / linear
(MIOpen / hipBLASLt) on the policy's INTERNAL parameter views, and the reference's loss graph on top of it
(accel_rl/algos/pg/aac_base.py:60-70).  The product has no such path -- its only backend is the hand-written
HIP forward / backward; this is what the numerics tests differentiate to check it."""
import torch
import torch.nn.functional as F

from accel_rl_amd.algos.pg.aac_base import valids_mean


def forward(policy, x):
    """prob [B,A], value [B] with gradients flowing into policy.params (hence policy.flat_grads).
    x: what policy._scaled returned (ObsRows on the u8 path, else the scaled NHWC tensor)."""
    if hasattr(x, "obs"):
        x = policy._scaled_f32(x.obs, x.idx)[:, :policy._c_in]
    p = policy.params
    for i, (nf, ci, sz, st, pad, ho, wo) in enumerate(policy._conv_geom):
        x = F.relu(F.conv2d(x, p[2 * i], p[2 * i + 1], stride=st, padding=pad))
    x = x.permute(0, 2, 3, 1).reshape(x.shape[0], -1)
    k = 2 * policy._n_conv
    for _ in policy._hid_geom:
        x = F.relu(F.linear(x, p[k], p[k + 1]))
        k += 2
    out = F.linear(x, p[k], p[k + 1])
    return torch.softmax(out[:, :policy.n_act], dim=1), out[:, policy.n_act]


def losses(algo, mb):
    """(pi_loss, v_loss, ent_loss) of a minibatch dict (full-batch arrays + idx), through autograd."""
    policy = algo.policy
    sel = None if mb.get("idx") is None else mb["idx"].long()
    pick = lambda x: x if (x is None or sel is None) else x.index_select(0, sel)      # noqa: E731
    prob, value = forward(policy, policy._scaled(mb["observations"], mb.get("idx")))
    valids = pick(mb.get("valids"))
    new_info, old_info = dict(prob=prob), dict(prob=pick(mb["old_prob"]))
    v_loss = algo.v_loss_coeff * valids_mean((value - pick(mb["returns"])) ** 2, valids)
    ent_loss = - algo.ent_loss_coeff * valids_mean(policy.distribution.entropy_sym(new_info), valids)
    pi_loss = algo.pi_loss(policy, pick(mb["actions"]), pick(mb["advantages"]), old_info, new_info, valids)
    return pi_loss, v_loss, ent_loss


class _TheanoSurrogate(torch.autograd.Function):
    """PPO's surrogate minimum(r A, clip(r, lo, hi) A) (accel_rl/algos/pg/ppo.py:45-49) with the gradient Theano's
    symbolic differentiation produces -- the closed form of oracle/ref_port.py::ppo_surrogate, written independently
    of accel_rl_amd/util/theano_ops.py (which composes the two ops).  Theano >= 0.8 (`both` False): a tie of the
    minimum goes to its first argument alone, d/dr = A [surr == s1] + A [surr != s1] [lo <= r <= hi]; Theano <= 0.7
    (`both`): every argument equal to the minimum receives it, d/dr = A [surr == s1] + A [surr == s2] [lo <= r <= hi]."""

    @staticmethod
    def forward(ctx, ratio, adv, lo, hi, both):
        s1 = ratio * adv
        s2 = torch.minimum(torch.maximum(ratio, lo), hi) * adv
        surr = torch.minimum(s1, s2)
        inside = (ratio >= lo) & (ratio <= hi)
        first = surr == s1
        ctx.save_for_backward(adv, first, ((surr == s2) if both else ~first) & inside)
        return surr

    @staticmethod
    def backward(ctx, g):
        adv, first, second = ctx.saved_tensors
        return g * adv * (first.to(g.dtype) + second.to(g.dtype)), None, None, None, None


def ppo_surrogate(ratio, adv, clip, tie_rule="theano"):
    """surr[B]; tie_rule "theano" = the reference's graph under Theano >= 0.8 (default of the product), "both" = under
    Theano <= 0.7, "math" = torch.minimum's own rule."""
    lo = torch.as_tensor(1. - clip, dtype=ratio.dtype, device=ratio.device)
    hi = torch.as_tensor(1. + clip, dtype=ratio.dtype, device=ratio.device)
    if tie_rule in ("theano", "both"):
        return _TheanoSurrogate.apply(ratio, adv, lo, hi, tie_rule == "both")
    return torch.minimum(ratio * adv, torch.clamp(ratio, float(lo), float(hi)) * adv)
