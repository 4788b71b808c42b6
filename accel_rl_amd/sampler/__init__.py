"""Samplers behind the reference's interface (accel_rl/sampler/base.py:11-51).

`ActsrvAltOvrlpSampler(EnvCls=..., env_args=..., horizon=..., n_parallel=..., envs_per=..., ...)` -- the name and the
constructor of the reference's sampler (act_server/alternating/overlap/sampler.py:20-38) -- picks the implementation by
the environment class, as SURVEY 8b's Env row describes the boundary:

  EnvCls.batched_device_env (e.g. SynthAtariEnv)   -> GpuVecSampler   (emulator, preprocessing, rollout buffer: HIP kernels)
  any other rllab-style Env (step / reset / spec)   -> HostEnvSampler  (worker processes on the host's cores feeding the
                                                                        same device rollout buffer)
"""


def ActsrvAltOvrlpSampler(EnvCls, **kwargs):
    if getattr(EnvCls, "batched_device_env", False):
        from accel_rl_amd.sampler.gpu_sampler import GpuVecSampler
        return GpuVecSampler(EnvCls=EnvCls, **kwargs)
    from accel_rl_amd.sampler.host_sampler import HostEnvSampler
    return HostEnvSampler(EnvCls=EnvCls, **kwargs)
