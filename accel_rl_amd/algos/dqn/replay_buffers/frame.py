"""Frame-dedup replay memory in HBM (reference: accel_rl/algos/dqn/replay_buffers/frame.py:9-166).

Same constructor, `append_data`, `extract_batch`, `extract_observations` as the reference's
FrameReplayBuffer; the per-environment Python objects (EnvBuffer) become one struct-of-arrays
on the device (include/accel_rl_hip.h: arl_replay) written and read by csrc/replay.hip.
`append_data` takes the sampler's samples buffer as it is (device tensors, env-major), so a
rollout goes from the sampler to replay memory without leaving HBM."""
import numpy as np
import torch

from accel_rl_amd import _lib


class FrameReplayBuffer(object):

    def __init__(self, env_spec, size, reward_horizon, sampling_horizon, n_environments, discount,
                 reward_dtype="float32", device="cuda:0", promo=_lib.PROMO_NEP50):
        if reward_dtype != "float32":
            raise NotImplementedError("device replay stores float32 rewards (the reference's default)")
        _lib.load()
        sampling_size = sampling_horizon * n_environments                 # frame.py:41-45
        n_chunks = -(-size // sampling_size)
        replay_size = n_chunks * sampling_size
        self.env_replay_size = env_size = replay_size // n_environments
        assert n_environments * env_size == replay_size
        self.n_environments = n_environments
        obs_shape = tuple(env_spec.observation_space.shape)
        self.num_img_obs = n = obs_shape[0]
        if n < 2:
            raise NotImplementedError("frame-dedup storage needs >= 2 stacked frames (as the reference does)")
        self.reward_horizon, self.sampling_horizon, self.discount = reward_horizon, sampling_horizon, discount
        self.frame_shape = obs_shape[1:]
        frame_bytes = int(np.prod(self.frame_shape))
        self.device = dev = torch.device(device)
        ring = env_size + n - 1
        self.frames = torch.zeros((n_environments, ring) + self.frame_shape, dtype=torch.uint8, device=dev)
        self.n_blanks = torch.zeros((n_environments, ring), dtype=torch.uint8, device=dev)
        self.acts = torch.zeros((n_environments, env_size), dtype=torch.uint8, device=dev)
        self.terminals = torch.zeros((n_environments, env_size), dtype=torch.uint8, device=dev)
        self.rewards = torch.zeros((n_environments, env_size), dtype=torch.float32, device=dev)
        self.returns = torch.zeros((n_environments, env_size), dtype=torch.float32, device=dev)
        rb = _lib.ArlReplay()
        rb.n_env, rb.size, rb.n_stack = n_environments, env_size, n
        rb.frame_bytes, rb.reward_horizon = frame_bytes, reward_horizon
        rb.frames, rb.n_blanks, rb.acts = self.frames.data_ptr(), self.n_blanks.data_ptr(), self.acts.data_ptr()
        rb.terminals, rb.rewards, rb.returns = (self.terminals.data_ptr(), self.rewards.data_ptr(),
                                                self.returns.data_ptr())
        self._rb, self._promo = rb, promo
        self.idx = 0                                                      # where the next state is written
        self._idx_host = None

    def append_data(self, samples_data):
        """frame.py:57-60.  samples_data: the sampler's buffer (observations u8[N*T,F,H,W], actions u8,
        rewards f32, dones bool/u8 -- device tensors, env-major)."""
        dones = samples_data["dones"]
        if dones.dtype == torch.bool:
            dones = dones.view(torch.uint8)
        _lib.replay_append(self._rb, samples_data["observations"], samples_data["actions"],
                           samples_data["rewards"], dones, self.sampling_horizon, self.idx, self.discount,
                           self._promo)
        self.idx = (self.idx + self.sampling_horizon) % self.env_replay_size

    def sample_batch(self, batch_size):
        raise NotImplementedError

    # ---- helpers (frame.py:69-90) ------------------------------------------
    def _upload_idxs(self, env_idxs, step_idxs):
        b = len(env_idxs)
        if self._idx_host is None or self._idx_host.shape[1] < b:
            self._idx_host = torch.zeros((2, b), dtype=torch.int32).pin_memory()
            self._idx_dev = torch.zeros((2, b), dtype=torch.int32, device=self.device)
            self._idx_event = torch.cuda.Event()
        else:
            self._idx_event.synchronize()           # the previous upload has left the pinned buffer
        self._idx_host[0, :b].copy_(torch.from_numpy(np.asarray(env_idxs).astype(np.int32)))
        self._idx_host[1, :b].copy_(torch.from_numpy(np.asarray(step_idxs).astype(np.int32)))
        self._idx_dev[:, :b].copy_(self._idx_host[:, :b], non_blocking=True)
        self._idx_event.record(torch.cuda.current_stream(self.device))
        return self._idx_dev[0, :b], self._idx_dev[1, :b]

    def extract_batch(self, env_idxs, step_idxs):
        """-> (observations, next_observations, actions, returns, terminals), device tensors."""
        if not isinstance(env_idxs, torch.Tensor):
            env_idxs, step_idxs = self._upload_idxs(env_idxs, step_idxs)
        b = env_idxs.numel()
        obs, nxt, acts, rets, terms, terms_bool = self._batch_outputs(b)
        _lib.replay_extract(self._rb, env_idxs, step_idxs, obs, nxt, acts, rets, terms)
        return obs, nxt, acts, rets, terms_bool

    # `reuse_outputs` (set by the DQN algorithms): every batch of one size lands in the SAME device tensors,
    # marked `_arl_static`, so a hipGraph-replayed update reads them in place instead of copies -- the
    # previous batch is overwritten, which the reference's fresh arrays would not be.
    reuse_outputs = False

    def _batch_outputs(self, b):
        cache = self.__dict__.setdefault("_out_cache", dict())
        if self.reuse_outputs and b in cache:
            return cache[b]
        shape = (b, self.num_img_obs) + self.frame_shape
        # one allocation, obs directly followed by next_obs: the Q policies' online pass over both (double DQN)
        # then reads the 2b rows in place (policies/dqn/q_policy_base.py: _pair_rows)
        both = torch.empty((2 * b,) + shape[1:], dtype=torch.uint8, device=self.device)
        obs, nxt = both[:b], both[b:]
        acts = torch.empty(b, dtype=torch.uint8, device=self.device)
        rets = torch.empty(b, dtype=torch.float32, device=self.device)
        terms = torch.empty(b, dtype=torch.uint8, device=self.device)
        out = (obs, nxt, acts, rets, terms, terms.view(torch.bool))
        if self.reuse_outputs:
            for t in out:
                t._arl_static = True
            cache[b] = out
        return out

    def extract_observations(self, env_idxs, step_idxs):
        return self.extract_batch(env_idxs, step_idxs)[0]
