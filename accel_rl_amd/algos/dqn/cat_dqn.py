"""Categorical DQN (reference: accel_rl/algos/dqn/cat_dqn.py:9-109): defaults adam with
epsilon = 0.01 / batch_size, epsilon-greedy 1 -> 0.01 (eval 0.001), support linspace(V_min, V_max,
n_atoms).  The loss graph of the reference is csrc/dqn.hip:arl_catdqn_loss."""
import numpy as np
import torch

from accel_rl_amd.algos.dqn.dqn import DQN
from accel_rl_amd.optimizers import update_methods


class CategoricalDQN(DQN):

    def __init__(self, V_min=-10, V_max=10, **kwargs):
        self.V_min, self.V_max = V_min, V_max
        super().__init__(**kwargs)

    def _get_default_sub_args(self):
        opt_args = dict(learning_rate=2.5e-4, update_method=update_methods.adam,
                        grad_norm_clip=10 if self.dueling_dqn else None,
                        update_method_args=dict(epsilon=0.01 / self.batch_size),
                        scale_conv_grads=self.dueling_dqn)
        eps_greedy_args = dict(initial=1., final=0.01, eval=0.001, anneal_steps=int(1e6))
        priority_args = dict(alpha=0.6, beta_initial=0.4, beta_final=1., beta_anneal_steps=50e6,
                             default_priority=1.)
        return opt_args, eps_greedy_args, priority_args

    def build_loss(self, env_spec, policy):
        assert bool(self.dueling_dqn) == bool(getattr(policy, "_dueling", False)), \
            "dueling_dqn and the policy's `dueling` must agree (the reference's scripts pass both)"
        z = np.linspace(self.V_min, self.V_max, policy.n_atoms, dtype=np.float32)      # cat_dqn.py:49-52
        policy.incorporate_z(z)
        gamma_n = float(np.float32(self.discount ** self.reward_horizon))
        inputs = ["obs", "next_obs", "act", "disc_n_return", "terminal"]
        if self.prioritized_replay:
            inputs.append("importance_sample_weights")

        def loss(minibatch):
            obs, next_obs, act, ret, term = minibatch[:5]
            isw = None
            if self.prioritized_replay:
                isw = minibatch[5]
                if not isinstance(isw, torch.Tensor):
                    isw = torch.as_tensor(np.asarray(isw, np.float32)).to(policy.device)
            term_u8 = term.view(torch.uint8) if term.dtype == torch.bool else term
            loss_rows, kl = policy.cat_loss_and_grads(obs, next_obs, act, ret, term_u8, isw, self.V_min, self.V_max,
                                                      gamma_n, double_dqn=self.double_dqn)
            return kl, loss_rows                    # (the loss is their sum: DqnOptimizer)

        return inputs, loss
