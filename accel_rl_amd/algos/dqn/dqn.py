"""DQN family, host-side logic (reference: accel_rl/algos/dqn/dqn.py:12-221): replay memory,
training intensity, epsilon / beta schedules, target-network period, evaluation hooks, and the
(double) Q-learning loss with the Huber clip -- the reference's Theano graph is csrc/dqn.hip:arl_dqn_loss."""
import numpy as np
import torch

from accel_rl_amd.algos.base import RLAlgorithm
from accel_rl_amd.algos.dqn.replay_buffers.prioritized import PrioritizedReplayBuffer
from accel_rl_amd.algos.dqn.replay_buffers.uniform import UniformReplayBuffer
from accel_rl_amd.optimizers import update_methods
from accel_rl_amd.optimizers.dqn import DqnOptimizer
from accel_rl_amd.util.quick_args import save_args


class DQN(RLAlgorithm):

    def __init__(self, discount=0.99, batch_size=32, min_steps_learn=int(5e4), delta_clip=1,
                 replay_size=int(1e6), training_intensity=8, target_update_steps=int(1e4), reward_horizon=1,
                 OptimizerCls=None, optimizer_args=None, eps_greedy_args=None, double_dqn=False,
                 dueling_dqn=False, prioritized_replay=False, priority_args=None):
        save_args(vars(), underscore=False)
        opt_args, eps_args, pri_args = self._get_default_sub_args()
        opt_args.update(optimizer_args or dict())
        self.optimizer = (OptimizerCls or DqnOptimizer)(**opt_args)
        eps_args.update(eps_greedy_args or dict())
        self._eps_initial, self._eps_final = eps_args["initial"], eps_args["final"]
        self._eps_eval, self._eps_anneal_steps = eps_args["eval"], eps_args["anneal_steps"]
        if prioritized_replay:
            pri_args.update(priority_args or dict())
            self._priority_beta_initial = pri_args["beta_initial"]
            self._priority_beta_final = pri_args["beta_final"]
            self._priority_beta_anneal_steps = pri_args["beta_anneal_steps"]
            self._priority_args = dict(alpha=pri_args["alpha"], beta_initial=pri_args["beta_initial"],
                                       default_priority=pri_args["default_priority"])
        self.need_extra_obs = False

    def _get_default_sub_args(self):
        """dqn.py:62-86"""
        opt_args = dict(learning_rate=2.5e-4, update_method=update_methods.rmsprop,
                        grad_norm_clip=10 if self.dueling_dqn else None,
                        update_method_args=dict(rho=0.95, epsilon=1e-6), scale_conv_grads=self.dueling_dqn)
        eps_greedy_args = dict(initial=1., final=0.1, eval=0.05, anneal_steps=int(1e6))
        d_clip = self.delta_clip
        priority_args = dict(alpha=0.6, beta_initial=0.4, beta_final=1., beta_anneal_steps=50e6,
                             default_priority=d_clip if d_clip is not None else 1.)
        return opt_args, eps_greedy_args, priority_args

    def initialize(self, policy, env_spec, sample_size, horizon, mid_batch_reset):
        """dqn.py:88-135; sample_size and horizon refer to the sampler."""
        assert self.training_intensity * sample_size % self.batch_size == 0
        self._updates_per_optimize = int((self.training_intensity * sample_size) // self.batch_size)
        self._eps_anneal_itr = max(1, self._eps_anneal_steps // sample_size)
        self._target_update_itr = max(1, self.target_update_steps // sample_size)
        self._min_itr_learn = self.min_steps_learn // sample_size
        if self.prioritized_replay:
            self._priority_beta_anneal_itr = max(1, self._priority_beta_anneal_steps // sample_size)
        if not mid_batch_reset:
            raise NotImplementedError
        if int(policy.recurrent):
            raise NotImplementedError
        self.policy = policy
        input_list, loss = self.build_loss(env_spec, policy)
        self.optimizer.initialize(inputs=input_list, loss=loss, target=policy)
        replay_args = dict(env_spec=env_spec, size=self.replay_size, reward_horizon=self.reward_horizon,
                           sampling_horizon=horizon, n_environments=sample_size // horizon,
                           discount=self.discount, reward_dtype="float32", device=policy.device)
        if self.prioritized_replay:
            replay_args.update(self._priority_args)
            self.replay_buffer = PrioritizedReplayBuffer(**replay_args)
        else:
            self.replay_buffer = UniformReplayBuffer(**replay_args)
        self.replay_buffer.reuse_outputs = True       # minibatches are consumed before the next one is drawn

    def build_loss(self, env_spec, policy):
        """dqn.py:137-172"""
        assert bool(self.dueling_dqn) == bool(getattr(policy, "_dueling", False)), \
            "dueling_dqn and the policy's `dueling` must agree (the reference's scripts pass both)"
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
            loss_rows, td_abs = policy.q_loss_and_grads(obs, next_obs, act, ret, term_u8, isw, gamma_n,
                                                        self.delta_clip, double_dqn=self.double_dqn)
            return td_abs, loss_rows                # (the loss is their sum: DqnOptimizer)

        return inputs, loss

    def optimize_policy(self, itr, samples_data):
        """dqn.py:174-193"""
        self.replay_buffer.append_data(samples_data)
        if itr < self._min_itr_learn:
            return None, dict()
        priorities, losses, slots = [], [], []
        for _ in range(self._updates_per_optimize):
            if self.prioritized_replay:
                opt_minibatch = self.replay_buffer.sample_batch(self.batch_size, device_weights=True)
            else:
                opt_minibatch = self.replay_buffer.sample_batch(self.batch_size)
            priority, loss = self.optimizer.optimize(opt_minibatch)
            if self.prioritized_replay:
                self.replay_buffer.update_batch_priorities(priority)
            if loss.dim() == 1:                       # a slot of the optimizer's statistics ring: [loss rows | priorities]
                slots.append(loss)
            else:
                priorities.append(priority[::8].clone())  # (downsample for stats)
                losses.append(loss)
        if slots:                                     # the update loop enqueued nothing for the statistics: one pass here
            stats = torch.stack(slots)
            b = stats.shape[1] // 2
            losses += list(stats[:, :b].sum(dim=1))
            priorities += list(stats[:, b::8])
        if itr % self._target_update_itr == 0:
            self.policy.update_target()
        self.update_epsilon(itr)
        if self.prioritized_replay:
            self.update_priority_beta(itr)
        return opt_minibatch, dict(Priority=priorities, Loss=losses)

    def update_epsilon(self, itr):
        prog = min(1, itr / self._eps_anneal_itr)
        self.policy.set_epsilon(prog * self._eps_final + (1 - prog) * self._eps_initial)

    def update_priority_beta(self, itr):
        prog = min(1, itr / self._priority_beta_anneal_itr)
        self.replay_buffer.set_beta(prog * self._priority_beta_final + (1 - prog) * self._priority_beta_initial)

    def set_n_itr(self, n_itr):
        self.n_itr = n_itr

    def prep_eval(self, itr):
        if itr > 0:
            self._prev_eps = self.policy.get_epsilon()
            self.policy.set_epsilon(self._eps_eval)

    def post_eval(self, itr):
        if itr > 0:
            self.policy.set_epsilon(self._prev_eps)

    @property
    def opt_info_keys(self):
        return ["Priority", "Loss"]
