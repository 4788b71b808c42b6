"""Theano's gradient rules for the two element-wise ops of PPO's surrogate (accel_rl/algos/pg/ppo.py:47-49),
as torch autograd Functions.  Theano is a third-party dependency of the reference (absent from /root/reference
and from this image: restated from its published source, theano/scalar/basic.py `Minimum.L_op`, `Clip.L_op`;
parity unpinned):

    minimum(x, y):  gx = eq(out, x) g,  gy = eq(out, y) g     -- a tie hands g to BOTH arguments
    clip(x, lo, hi): gx = ((x >= lo) & (x <= hi)) g            -- bounds included

The learner never calls these (the same rule lives inside `head_kernel`, csrc/learner.hip, selected by
ARL_PPO_TIE_THEANO); `BasePPO.pi_loss` is written with them so that the formula the numerics tests differentiate
is the reference's graph and not torch.minimum's (which splits a tie's gradient evenly)."""
import torch


class _Minimum(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, y):
        out = torch.minimum(x, y)
        ctx.save_for_backward(x, y, out)
        return out

    @staticmethod
    def backward(ctx, g):
        x, y, out = ctx.saved_tensors
        return (out == x).to(g.dtype) * g, (out == y).to(g.dtype) * g


class _Clip(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, lo, hi):
        ctx.save_for_backward(x, lo, hi)
        return torch.minimum(torch.maximum(x, lo), hi)

    @staticmethod
    def backward(ctx, g):
        x, lo, hi = ctx.saved_tensors
        return ((x >= lo) & (x <= hi)).to(g.dtype) * g, None, None      # the bounds are not parameters here


def minimum(x, y):
    return _Minimum.apply(x, y)


def clip(x, lo, hi):
    lo = torch.as_tensor(lo, dtype=x.dtype, device=x.device)
    hi = torch.as_tensor(hi, dtype=x.dtype, device=x.device)
    return _Clip.apply(x, lo, hi)
