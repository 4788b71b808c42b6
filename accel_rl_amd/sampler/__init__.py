"""Samplers behind the reference's interface (accel_rl/sampler/base.py:11-51).

`ActsrvAltOvrlpSampler(EnvCls=..., env_args=..., horizon=..., n_parallel=..., envs_per=..., ...)` -- the name and the
constructor of the reference's sampler (act_server/alternating/overlap/sampler.py:20-38) -- picks the implementation by
the environment class, as SURVEY 8b's Env row describes the boundary:

  EnvCls.batched_device_env (e.g. SynthAtariEnv)   -> GpuVecSampler   (emulator, preprocessing, rollout buffer: HIP kernels)
  any other rllab-style Env (step / reset / spec)   -> HostEnvSampler  (worker processes on the host's cores feeding the
                                                                        same device rollout buffer)

`AAOEvalSampler(eval_steps=..., eval_envs_per=..., EnvCls=..., ...)` (sampler_with_eval.py:6-54) picks between
GpuVecEvalSampler and HostEnvEvalSampler the same way.
"""


def ActsrvAltOvrlpSampler(EnvCls, **kwargs):
    if getattr(EnvCls, "batched_device_env", False):
        from accel_rl_amd.sampler.gpu_sampler import GpuVecSampler
        return GpuVecSampler(EnvCls=EnvCls, **kwargs)
    from accel_rl_amd.sampler.host_sampler import HostEnvSampler
    return HostEnvSampler(EnvCls=EnvCls, **kwargs)


def AAOEvalSampler(eval_steps, eval_envs_per, EnvCls, **kwargs):
    if getattr(EnvCls, "batched_device_env", False):
        from accel_rl_amd.sampler.gpu_sampler_with_eval import GpuVecEvalSampler
        return GpuVecEvalSampler(eval_steps=eval_steps, eval_envs_per=eval_envs_per, EnvCls=EnvCls, **kwargs)
    from accel_rl_amd.sampler.host_sampler import HostEnvEvalSampler
    return HostEnvEvalSampler(eval_steps=eval_steps, eval_envs_per=eval_envs_per, EnvCls=EnvCls, **kwargs)
