"""PPO (reference: accel_rl/algos/pg/ppo.py:11-67): defaults lr 1e-3, adam(eps 1e-5),
4 epochs, minibatch 512, no grad clip, shuffle, gamma 0.99, lambda 0.95,
clip 0.2 * lr_mult, v_loss_coeff 1.

`ppo_tie_rule` (not a reference argument): the surrogate min(r A, clip(r) A) is a tie wherever r is inside the clip
range, and what gradient a tie passes on is the differentiating framework's choice.  "theano" (default) = the
reference's: Theano >= 0.8 hands a tie of T.minimum to its FIRST argument alone (util/theano_ops.py has the source
lines), here the unclipped branch, so an unclipped sample's policy gradient is A; "math" = the mathematical derivative
(the same except where the two branches are equal by rounding outside the range); "both" = Theano <= 0.7, where a tie
fed both arguments: 2 A for every unclipped sample."""
import torch

from accel_rl_amd.algos.pg.aac_base import AdvActorCriticBase, valids_mean
from accel_rl_amd.optimizers import update_methods
from accel_rl_amd.optimizers.single import PpoOptimizer
from accel_rl_amd.optimizers.sync import SyncPpoOptimizer
from accel_rl_amd.util import theano_ops

TIE_RULES = dict(theano=0, math=1, both=2)       # ARL_PPO_TIE_THEANO / _MATH / _BOTH (include/accel_rl_hip.h)


class BasePPO(AdvActorCriticBase):

    def __init__(self, OptimizerCls, optimizer_args=None, discount=0.99, gae_lambda=0.95,
                 clip_param=0.2, ppo_tie_rule="theano", **kwargs):
        if ppo_tie_rule not in TIE_RULES:
            raise ValueError("ppo_tie_rule must be 'theano', 'math' or 'both', got %r" % (ppo_tie_rule,))
        self.ppo_tie_rule = ppo_tie_rule
        self.loss_tie_rule = TIE_RULES[ppo_tie_rule]
        args = dict(num_slices=1, learning_rate=1e-3, epochs=4, minibatch_size=64 * 8,
                    update_method=update_methods.adam, update_method_args=dict(epsilon=1e-5),
                    grad_norm_clip=None, shuffle=True)
        args.update(optimizer_args or dict())
        self.optimizer = OptimizerCls(**args)
        self.clip_param = clip_param
        super().__init__(discount=discount, gae_lambda=gae_lambda, **kwargs)

    loss_kind = 1

    def pi_loss(self, policy, act, adv, old_dist_info, new_dist_info, valids):
        ratio = policy.distribution.likelihood_ratio_sym(act, old_dist_info, new_dist_info)
        clip = self.clip_param * self._lr_mult                 # ppo.py:46 (anneals with lr)
        surr_1 = ratio * adv
        if self.ppo_tie_rule in ("theano", "both"):
            surr_2 = theano_ops.clip(ratio, 1. - clip, 1. + clip) * adv
            return - valids_mean(theano_ops.minimum(surr_1, surr_2, both=self.ppo_tie_rule == "both"), valids)
        surr_2 = torch.minimum(torch.maximum(ratio, 1. - clip), 1. + clip) * adv
        return - valids_mean(torch.minimum(surr_1, surr_2), valids)


class PPO(BasePPO):
    """Single GPU"""

    def __init__(self, OptimizerCls=PpoOptimizer, **kwargs):
        super().__init__(OptimizerCls=OptimizerCls, **kwargs)


class mPPO(BasePPO):
    """Multi-GPU synchronous"""

    def __init__(self, OptimizerCls=SyncPpoOptimizer, **kwargs):
        super().__init__(OptimizerCls=OptimizerCls, **kwargs)
