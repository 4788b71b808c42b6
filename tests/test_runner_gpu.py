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


def test_grad_norm_stats_vary_over_a_log_interval_in_graph_mode():
    """A2C = one update per iteration, 4 iterations per log interval, the last interval runs entirely from the
    learner's hipGraph: the logged GradNorm entries must be the 4 iterations' own values (the graph-owned output
    tensor is overwritten by every replay), i.e. Std > 0 and Min < Max."""
    runner, sampler, algo, policy = _build("a2c", 160 * 12, max_path_length=40)
    runner.train()
    tab = runner.last_tabular
    assert algo._graph is not None
    assert tab["GradNormStd"] > 0 and tab["GradNormMin"] < tab["GradNormMax"]


def test_seeded_runs_are_reproducible():
    out = []
    for _ in range(2):
        runner, sampler, algo, policy = _build("ppo", 160 * 4, max_path_length=30)
        runner.train()
        out.append(policy.get_param_values())
    np.testing.assert_array_equal(out[0], out[1])       # deterministic kernels + seeded RNG streams


def test_snapshot_resume_restores_the_policy(tmp_path, monkeypatch):
    """itr_N.pkl content (accel_rl_base.py:108-113) and resuming from it through the
    policy's `initial_param_values`, as a reference script would."""
    import os
    from accel_rl_amd.policies.atari_cnn_policy import AtariCnnPolicy
    from accel_rl_amd.policies.atari_cnn_specs import cnn_specs
    from accel_rl_amd.util import logger
    from accel_rl_amd.util import logging as arl_logging
    monkeypatch.setattr(arl_logging, "LOG_DIR", str(tmp_path))
    runner, sampler, algo, policy = _build("ppo", 160 * 4, max_path_length=30)
    with arl_logging.logger_context(str(tmp_path / "run"), "ppo_breakout", 0, snapshot_mode="last") as exp_dir:
        runner.train()
    logger.set_snapshot_mode("none")
    logger.set_snapshot_dir(None)
    snap = logger.load_itr_params(os.path.join(exp_dir, "params.pkl"))
    assert set(snap) == {"itr", "cum_samples", "policy_param_values"} and snap["cum_samples"] == snap["itr"] * 160
    resumed = AtariCnnPolicy(initial_param_values=snap["policy_param_values"], **cnn_specs[0])
    resumed.initialize(sampler.env_spec, device="cuda:0")
    np.testing.assert_array_equal(resumed.get_param_values(), snap["policy_param_values"])
    obs = sampler.samples_buf.observations[:32]
    p0, v0 = policy.prob_value(obs)
    if snap["itr"] == runner._n_itr - 1:            # the last snapshot is the final policy
        p1, v1 = resumed.prob_value(obs)
        assert torch.equal(p0, p1) and torch.equal(v0, v1)
