"""Rainbow minus noisy nets: categorical + dueling + double DQN, 3-step returns, prioritized replay,
epsilon-greedy exploration (reference: accel_rl/algos/dqn/eps_rainbow.py:5-57 -- only defaults differ from
CategoricalDQN).  Use with AtariCatDqnPolicy(dueling=True)."""
from accel_rl_amd.algos.dqn.cat_dqn import CategoricalDQN
from accel_rl_amd.optimizers import update_methods


class EpsRainbow(CategoricalDQN):

    def __init__(self, reward_horizon=3, double_dqn=True, dueling_dqn=True, prioritized_replay=True,
                 target_update_steps=int(8e3), min_steps_learn=int(2e4), **kwargs):
        super().__init__(reward_horizon=reward_horizon, double_dqn=double_dqn, dueling_dqn=dueling_dqn,
                         prioritized_replay=prioritized_replay, target_update_steps=target_update_steps,
                         min_steps_learn=min_steps_learn, **kwargs)

    def _get_default_sub_args(self):
        opt_args = dict(learning_rate=6.25e-5, update_method=update_methods.adam, grad_norm_clip=10,
                        update_method_args=dict(epsilon=0.005 / self.batch_size),
                        scale_conv_grads=self.dueling_dqn)
        eps_greedy_args = dict(initial=1., final=0.01, eval=0.001, anneal_steps=int(62.5e3))
        priority_args = dict(alpha=0.5, beta_initial=0.4, beta_final=1., beta_anneal_steps=50e6,
                             default_priority=1.)
        return opt_args, eps_greedy_args, priority_args
