"""A2C (reference: accel_rl/algos/pg/a2c.py:13-64): defaults lr 7e-4, rmsprop,
grad-norm clip 0.5, gamma 0.99, lambda 1 (n-step returns), v_loss_coeff 0.25."""
from accel_rl_amd.algos.pg.aac_base import AdvActorCriticBase, valids_mean
from accel_rl_amd.optimizers import update_methods
from accel_rl_amd.optimizers.single import A2cOptimizer
from accel_rl_amd.optimizers.sync import SyncA2cOptimizer


class BaseA2C(AdvActorCriticBase):

    def __init__(self, OptimizerCls, optimizer_args=None, discount=0.99, gae_lambda=1,
                 v_loss_coeff=0.25, **kwargs):
        args = dict(learning_rate=7e-4, update_method=update_methods.rmsprop,
                    update_method_args=dict(), grad_norm_clip=0.5)
        args.update(optimizer_args or dict())
        self.optimizer = OptimizerCls(**args)
        super().__init__(discount=discount, gae_lambda=gae_lambda, v_loss_coeff=v_loss_coeff,
                         **kwargs)

    loss_kind = 0

    def pi_loss(self, policy, act, adv, old_dist_info, new_dist_info, valids):
        logli = policy.distribution.log_likelihood_sym(act, new_dist_info)
        return - valids_mean(logli * adv, valids)


class A2C(BaseA2C):
    """Single GPU"""

    def __init__(self, OptimizerCls=A2cOptimizer, **kwargs):
        super().__init__(OptimizerCls=OptimizerCls, **kwargs)


class mA2C(BaseA2C):
    """Multi-GPU synchronous"""

    def __init__(self, OptimizerCls=SyncA2cOptimizer, **kwargs):
        super().__init__(OptimizerCls=OptimizerCls, **kwargs)
