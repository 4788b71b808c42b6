"""End-to-end on the GPU through the reference-shaped plugin surface: build sampler /
algo / policy / runner exactly as accel_rl/scripts/example/example_train_{ppo,a2c}.py
do and call runner.train()."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _build(kind, n_steps, **sampler_kw):
    from accel_rl_amd.algos.pg.a2c import A2C
    from accel_rl_amd.algos.pg.ppo import PPO
    from accel_rl_amd.envs.synthetic_atari import SynthAtariEnv
    from accel_rl_amd.policies.atari_cnn_policy import AtariCnnPolicy
    from accel_rl_amd.policies.atari_cnn_specs import cnn_specs
    from accel_rl_amd.runners.accel_rl import AccelRL
    from accel_rl_amd.sampler.gpu_sampler import GpuVecSampler
    from accel_rl_amd.util import logger
    logger.set_quiet(True)
    sampler = GpuVecSampler(EnvCls=SynthAtariEnv, env_args=dict(game="pong" if kind == "a2c" else "breakout"),
                            horizon=5, n_parallel=4, envs_per=4, max_decorrelation_steps=50,
                            device="cuda:0", **sampler_kw)
    if kind == "ppo":
        algo = PPO(optimizer_args=dict(minibatch_size=64, epochs=2), lr_schedule="linear", standardize_adv=True)
    else:
        algo = A2C(standardize_adv=True)
    policy = AtariCnnPolicy(**cnn_specs[0])
    runner = AccelRL(algo=algo, policy=policy, sampler=sampler, n_steps=n_steps, seed=11,
                     log_interval_steps=640, log_traj_window=50)
    return runner, sampler, algo, policy


@pytest.mark.parametrize("kind,mid_batch_reset,max_len", [("ppo", True, 40), ("a2c", False, 23)])
def test_train_loop_runs_and_logs(kind, mid_batch_reset, max_len):
    runner, sampler, algo, policy = _build(kind, 160 * 12, mid_batch_reset=mid_batch_reset,
                                           max_path_length=max_len)
    runner.train()
    tab = runner.last_tabular
    assert algo._use_valids == (not mid_batch_reset)
    for key in ("Iteration", "CumCompletedTrajs", "CumCompletedSteps", "CumTotalSteps", "NewCompletedTrajs",
                "StepsInTrajWindow", "Entropy", "Perplexity", "LengthAverage", "ReturnAverage",
                "RawReturnAverage", "NonzeroRewardsAverage", "DiscountedReturnAverage", "GradNormAverage",
                "ParamsNorm", "NormFromInit", "CumTime (s)", "SamplesPerSecond"):
        assert key in tab, key
    assert runner._n_itr == 13 and tab["Iteration"] == 11 and tab["CumTotalSteps"] == 12 * 160
    assert tab["CumCompletedTrajs"] > 0 and tab["LengthAverage"] == max_len + 1     # over-length resets
    assert np.isfinite(tab["GradNormAverage"]) and tab["NormFromInit"] > 0
    assert 0 < tab["Entropy"] <= np.log(policy.n_act) + 1e-3 and tab["SamplesPerSecond"] > 0
    assert torch.isfinite(policy.flat_params).all()
    if kind == "ppo":       # linear schedule reached (n_itr - itr)/n_itr at the last iteration
        assert abs(algo._lr_mult.item() - (13 - 12) / 13) < 1e-6


def test_seeded_runs_are_reproducible():
    out = []
    for _ in range(2):
        runner, sampler, algo, policy = _build("ppo", 160 * 4, max_path_length=30)
        runner.train()
        out.append(policy.get_param_values())
    np.testing.assert_array_equal(out[0], out[1])       # deterministic kernels + seeded RNG streams
