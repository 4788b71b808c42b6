"""Theano's gradient rules for the two element-wise ops of PPO's surrogate (accel_rl/algos/pg/ppo.py:47-49),
as torch autograd Functions.  Theano is a third-party dependency of the reference (absent from /root/reference
and from this image: restated from its published source, theano/scalar/basic.py; parity unpinned).  The reference
imports theano.gpuarray, i.e. runs on Theano >= 0.9, and since 0.8 the rules are

    minimum(x, y):   e = eq(out, x);  gx = e g;  gy = (1 - e) g     -- a tie hands g to the FIRST argument alone
                     ("This form handle the case when both value are the same. In that case, gx will be gz, gy will
                      be 0."; theano/tensor/tests/test_basic.py::test_maximum_minimum_grad asserts [[1], [0]] at x == y)
    clip(x, lo, hi): gx = ((x >= lo) & (x <= hi)) g                 -- bounds included

(`both=True` is Theano <= 0.7's minimum: gx = eq(out, x) g, gy = eq(out, y) g -- a tie feeds both arguments.)

The learner never calls these (the same rules live inside `head_kernel`, csrc/learner.hip, selected by
ARL_PPO_TIE_*); `BasePPO.pi_loss` is written with them so that the formula the numerics tests differentiate is the
reference's graph and not torch.minimum's (which splits a tie's gradient evenly)."""
import torch


class _Minimum(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, y, both):
        out = torch.minimum(x, y)
        ctx.save_for_backward(x, y, out)
        ctx.both = both
        return out

    @staticmethod
    def backward(ctx, g):
        x, y, out = ctx.saved_tensors
        e = (out == x).to(g.dtype)
        gy = (out == y).to(g.dtype) if ctx.both else 1 - e
        return e * g, gy * g, None


class _Clip(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, lo, hi):
        ctx.save_for_backward(x, lo, hi)
        return torch.minimum(torch.maximum(x, lo), hi)

    @staticmethod
    def backward(ctx, g):
        x, lo, hi = ctx.saved_tensors
        return ((x >= lo) & (x <= hi)).to(g.dtype) * g, None, None      # the bounds are not parameters here


def minimum(x, y, both=False):
    return _Minimum.apply(x, y, both)


def clip(x, lo, hi):
    lo = torch.as_tensor(lo, dtype=x.dtype, device=x.device)
    hi = torch.as_tensor(hi, dtype=x.dtype, device=x.device)
    return _Clip.apply(x, lo, hi)
