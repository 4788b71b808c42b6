"""DQN-family optimizer (reference: accel_rl/optimizers/single/dqn_optimizer.py:11-54): one gradient
step per call on a replay minibatch, optional global-norm clip, adam / rmsprop on the flat bucket
(csrc/optim.hip).  `loss` is a callable(minibatch tuple) -> (priority f32[B], loss scalar) that leaves
the gradient in the policy's flat bucket (the algorithm builds it)."""
from accel_rl_amd.optimizers.base import BaseOptimizer


class DqnOptimizer(BaseOptimizer):

    def __init__(self, learning_rate, update_method, update_method_args=None, grad_norm_clip=None,
                 scale_conv_grads=False):
        if scale_conv_grads:
            raise NotImplementedError("scale_conv_grads belongs to the dueling architecture, which is not built")
        self._learning_rate = learning_rate
        self._update_method = update_method
        self._update_args = update_method.resolve(**(update_method_args or dict()))
        self._grad_norm_clip = grad_norm_clip

    def initialize(self, inputs, loss, target, priority_expr=None, givens=None, lr_mult=1):
        self._input_names = list(inputs)
        self._loss_fn = loss
        self._setup_bucket(target, lr_mult, dict(explicit_grads=True))
        self._set_updates_per_call(1)

    def optimize(self, inputs):
        priority, loss = self._loss_fn(inputs)
        self._apply_update(1.0)
        return priority, loss

    @property
    def parallelism_tag(self):
        return "single"
