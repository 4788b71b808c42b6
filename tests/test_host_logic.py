"""CPU-only tests of the host-side mirror of the reference interface: iteration
arithmetic and minibatch index streams against the golden vectors, buffer
construction, spaces' RNG draws, constructor-argument capture, logger."""
import numpy as np
import pytest
import torch

from conftest import load_golden


def test_get_n_itr_matches_reference_table():
    from accel_rl_amd.runners.accel_rl import AccelRLBase
    from accel_rl_amd.util import logger
    logger.set_quiet(True)
    for n_steps, sample_size, log_steps, n_itr, log_itrs in load_golden("g9_nitr")["table"]:
        r = AccelRLBase.__new__(AccelRLBase)
        r.n_steps, r._log_steps = int(n_steps), int(log_steps)
        assert r.get_n_itr(int(sample_size)) == n_itr
        assert r._log_interval_itrs == log_itrs


def test_iterate_mb_idxs_matches_reference_stream():
    from accel_rl_amd.optimizers.base import iterate_mb_idxs
    g = load_golden("g8_mbidx")
    for c in range(int(g["n_cases"])):
        bs, n, seed = [int(x) for x in g["c%d_cfg" % c]]
        np.random.seed(seed)
        for ep in range(3):
            got = list(iterate_mb_idxs(bs, n, shuffle=True))
            want = g["c%d_idx" % c][ep]
            assert len(got) == len(want)
            for a, b in zip(got, want):
                np.testing.assert_array_equal(a, b)
        ns = list(iterate_mb_idxs(bs, n, shuffle=False))
        np.testing.assert_array_equal(
            np.array([[m[0], m[-1] + 1] for m in ns], np.int64).reshape(-1, 2), g["c%d_noshuffle" % c])


def test_iterate_traj_idxs_matches_reference_stream():
    from accel_rl_amd.optimizers.base import iterate_traj_idxs
    g = load_golden("g15_trajidx")
    for c in range(int(g["n_cases"])):
        bs, n, horizon, seed = [int(x) for x in g["c%d_cfg" % c]]
        np.random.seed(seed)
        for ep in range(2):
            got = list(iterate_traj_idxs(bs, n, horizon=horizon, shuffle=True))
            np.testing.assert_array_equal(np.stack([m[0] for m in got]), g["c%d_e%d_idx" % (c, ep)])
            np.testing.assert_array_equal(np.stack([m[1] for m in got]), g["c%d_e%d_trajs" % (c, ep)])
        got = list(iterate_traj_idxs(bs, n, horizon=horizon, shuffle=False))
        np.testing.assert_array_equal(np.stack([m[0] for m in got]), g["c%d_plain_idx" % c])
        np.testing.assert_array_equal(np.stack([m[1] for m in got]), g["c%d_plain_trajs" % c])
        np.testing.assert_array_equal(np.random.randint(0, 2 ** 31 - 1, size=2), g["c%d_after" % c])   # same draws consumed
    with pytest.raises(AssertionError):
        list(iterate_traj_idxs(12, 20, horizon=5))


def test_parallelism_mismatch_raises_type_error():
    from accel_rl_amd.runners.accel_rl import AccelRL
    from accel_rl_amd.algos.pg.ppo import mPPO, PPO
    with pytest.raises(TypeError, match="mismatched parallelism"):
        AccelRL(algo=mPPO(), policy=None, sampler=None, n_steps=10)
    AccelRL(algo=PPO(), policy=None, sampler=None, n_steps=10)


def test_algo_defaults_match_reference():
    from accel_rl_amd.algos.pg.a2c import A2C
    from accel_rl_amd.algos.pg.ppo import PPO
    a = A2C()
    assert (a.discount, a.gae_lambda, a.v_loss_coeff, a.ent_loss_coeff) == (0.99, 1, 0.25, 0.01)
    o = a.optimizer
    assert (o._learning_rate, o._grad_norm_clip, o._update_method.name) == (7e-4, 0.5, "rmsprop")
    assert o._update_args == dict(rho=0.9, epsilon=1e-6)
    p = PPO()
    assert (p.discount, p.gae_lambda, p.v_loss_coeff, p.clip_param) == (0.99, 0.95, 1, 0.2)
    o = p.optimizer
    assert (o._learning_rate, o._epochs, o._minibatch_size, o._shuffle, o._grad_norm_clip) == \
        (1e-3, 4, 512, True, None)
    assert o._update_args == dict(beta1=0.9, beta2=0.999, epsilon=1e-5)
    with pytest.raises(ValueError):
        PPO(lr_schedule="cosine")


def test_buffers_layout_and_errors():
    from accel_rl_amd import buffers as B
    ex = dict(observations=np.zeros((4, 3, 2), np.uint8), rewards=np.float32(0), dones=False,
              env_infos=dict(need_reset=False, raw_reward=np.float32(0)))
    buf = B.buffer_with_segs_view(ex, 6 * 5, 5, "cpu")
    assert buf.observations.shape == (30, 4, 3, 2) and buf.observations.dtype == torch.uint8
    assert buf.dones.dtype == torch.bool and buf.env_infos["raw_reward"].dtype == torch.float32
    assert B.buffer_length(buf) == 30 and len(buf.segs_view) == 6
    buf.segs_view[2].rewards[3] = 7.0                     # views: flat index = env*T + t
    assert buf.rewards[2 * 5 + 3] == 7.0
    buf.segs_view[4].env_infos["need_reset"][0] = True
    assert buf.env_infos["need_reset"][20]
    pol = B.buffer_with_segs_view(dict(actions=np.uint8(0), agent_infos=dict(value=np.float32(0))), 30, 5, "cpu")
    both = B.combine_distinct_buffers(buf, pol)
    assert set(both.segs_view[0]) == {"observations", "rewards", "dones", "env_infos", "actions", "agent_infos"}
    assert B.count_buffer_size(both) == 30 * (24 + 4 + 1 + 1 + 4 + 1 + 4)
    with pytest.raises(ValueError):
        B.view_segments(buf, 7)
    with pytest.raises(TypeError):
        B.build_array(np.array([object()]), 3, "cpu")
    buf.extra_observations = torch.zeros(6, 4, 3, 2)
    assert B.buffer_length(buf) == 30                     # "extra*" keys are exempt
    buf.bad = torch.zeros(3)
    with pytest.raises(RuntimeError):
        B.buffer_length(buf)


def test_spaces_draws_match_the_reference_calls():
    from accel_rl_amd.spaces import Discrete, UintBox
    d = Discrete(6)
    assert d.dtype == "uint8" and Discrete(300).dtype == "uint16" and Discrete(70000).dtype == "uint32"
    np.random.seed(4)
    a = d.sample()
    np.random.seed(4)
    assert a == np.random.randint(6, dtype="uint8")
    box = UintBox(shape=(4, 104, 80), bits=8)
    np.random.seed(5)
    x = box.sample()
    np.random.seed(5)
    np.testing.assert_array_equal(x, np.random.randint(low=0, high=255, size=(4, 104, 80), dtype="uint8"))
    assert x.max() <= 254 and box.contains(x)


def test_env_descriptor_consumes_reference_draws():
    from accel_rl_amd.envs.synthetic_atari import SynthAtariEnv, K_FRAMES
    from oracle.ref_port import PortedAtariEnv
    np.random.seed(9)
    env = SynthAtariEnv(game="seaquest", max_start_noops=7)
    after = np.random.rand()
    np.random.seed(9)
    port = PortedAtariEnv(game="seaquest", max_start_noops=7)
    assert np.random.rand() == after and env.phase == port.phase < K_FRAMES
    assert env.action_space.n == 18 and env.observation_space.shape == (4, 104, 80)
    assert env.env_info_keys == ["raw_reward", "need_reset"]
    with pytest.raises(IOError):
        SynthAtariEnv(game="nonexistent")


def test_gpu_sampler_refuses_plain_envs_and_missing_gpu():
    from accel_rl_amd.sampler.gpu_sampler import GpuVecSampler

    class PlainEnv(object):
        pass
    s = GpuVecSampler(EnvCls=PlainEnv, env_args=dict(), horizon=5)
    with pytest.raises(TypeError, match="batched device protocol"):
        s.initialize(seed=1)
    assert s.total_n_envs == 2 and s.alternating is False


def test_logger_tabular_and_misc_stat(tmp_path):
    from accel_rl_amd.util import logger
    logger.set_quiet(True)
    logger.set_output(str(tmp_path))
    logger.record_tabular("Iteration", 3)
    logger.record_tabular_misc_stat("Return", [1.0, 2.0, 3.0])
    row = logger.dump_tabular()
    assert row["ReturnAverage"] == 2.0 and row["ReturnMedian"] == 2.0 and row["ReturnMax"] == 3.0
    text = (tmp_path / "progress.csv").read_text().splitlines()
    assert text[0].startswith("Iteration,ReturnAverage,ReturnStd")
    logger.set_output(None)


def test_affinity_codes_match_the_reference():
    """G10: encode / decode / build for six machine shapes, recorded from the reference's own
    accel_rl/scripts/launching/affinities.py (tests/golden/gen_golden_affinities.py)."""
    import json
    import os
    from accel_rl_amd.scripts.launching import affinities as A
    with open(os.path.join(os.path.dirname(__file__), "golden", "g10_affinities.json")) as f:
        cases = json.load(f)
    norm = lambda x: json.loads(json.dumps(x))          # noqa: E731  (tuples -> lists, as stored)
    for c in cases:
        kw = c["kwargs"]
        assert A.encode_affinity_params(**kw) == c["code"]
        assert A.decode_affinity_params(c["code"]) == c["decoded"]
        for s in c["slots"]:
            assert norm(A.get_affinities(s["code"])) == s["affinities"], s["code"]
        assert norm(A.build_all_affinities(**kw)) == c["all"]
    with pytest.raises(ValueError):
        A.decode_affinity_params("8gpu_3xyz")


def test_logger_context_files_and_snapshots(tmp_path, monkeypatch):
    """accel_rl/util/logging.py:24-49 + rllab snapshot modes: progress.csv, debug.log, params.json,
    itr_N.pkl under <log_dir>/<name>_<run_ID>/; a snapshot restores the parameter vector."""
    import json
    import os
    from accel_rl_amd.util import logger
    from accel_rl_amd.util import logging as arl_logging
    monkeypatch.setattr(arl_logging, "LOG_DIR", str(tmp_path))
    logger.set_quiet(True)
    with arl_logging.logger_context(str(tmp_path / "exp"), "ppo", 7, dict(game="pong"), snapshot_mode="gap") as exp_dir:
        logger.set_snapshot_gap(2)
        for itr in range(5):
            logger.log("hello %d" % itr)
            logger.record_tabular("Iteration", itr)
            logger.record_tabular_misc_stat("Return", [itr, itr + 1.])
            logger.dump_tabular()
            logger.save_itr_params(itr, dict(itr=itr, cum_samples=itr * 1280,
                                             policy_param_values=np.arange(4, dtype=np.float32) + itr))
    assert exp_dir == str(tmp_path / "exp" / "ppo_7")
    assert json.load(open(os.path.join(exp_dir, "params.json"))) == dict(game="pong", name="ppo", run_ID=7)
    rows = open(os.path.join(exp_dir, "progress.csv")).read().strip().split("\n")
    assert rows[0].startswith("Iteration,ReturnAverage,ReturnStd") and len(rows) == 6
    text = open(os.path.join(exp_dir, "debug.log")).read()
    assert "ppo_7 hello 3" in text and "ReturnAverage" in text
    assert sorted(f for f in os.listdir(exp_dir) if f.endswith(".pkl")) == ["itr_0.pkl", "itr_2.pkl", "itr_4.pkl"]
    snap = logger.load_itr_params(os.path.join(exp_dir, "itr_4.pkl"))
    assert snap["itr"] == 4 and snap["cum_samples"] == 5120
    np.testing.assert_array_equal(snap["policy_param_values"], np.arange(4, dtype=np.float32) + 4)
    logger.set_snapshot_mode("none")
    logger.set_snapshot_dir(None)


def test_ppo_surrogate_gradient_rules():
    """The three statements of Theano's min / clip gradient agree on the samples where the rules differ: the oracle's
    closed form (oracle/ref_port.py::ppo_surrogate), the product's composed autograd ops (util/theano_ops.py, what
    BasePPO.pi_loss is written with) and the tests' one-Function restatement (tests/autograd_ref.py).  Theano >= 0.8
    (what the reference runs on; the default) hands a tie of T.minimum(surr_1, surr_2) to surr_1 alone: A inside the clip
    range; Theano <= 0.7 ("both") fed both arguments: 2 A inside the range, bounds included
    (accel_rl/algos/pg/ppo.py:47-49)."""
    import torch
    import autograd_ref
    from accel_rl_amd.util import theano_ops
    from oracle import ref_port as P
    clip = 0.25                                                    # 0.75 and 1.25 are exact in fp32
    ratio = np.array([1.0, 0.75, 1.25, 0.7499999, 1.2500001, 0.5, 2.0, 0.5, 2.0, 1.0, 0.9, 1.1, 3.0, 1.3], np.float32)
    adv = np.array([1.5, -2.0, 0.5, 1.0, 1.0, 1.0, 1.0, -1.0, -1.0, 0.0, -0.25, 4.0, 0.0, 1e-45], np.float32)
    want = dict(
        theano=[1.5, -2.0, 0.5, 1.0, 0.0, 1.0, 0.0, 0.0, -1.0, 0.0, -0.25, 4.0, 0.0, 1e-45],
        both=[3.0, -4.0, 1.0, 1.0, 0.0, 1.0, 0.0, 0.0, -1.0, 0.0, -0.5, 8.0, 0.0, 1e-45],
        # (last sample: outside the range the two branches are EQUAL by rounding -- the smallest denormal as advantage:
        #  1.3 x 1e-45 and 1.25 x 1e-45 are both 1e-45 -- and the tie goes to the first argument; the mathematical derivative is 0)
        math=[1.5, -2.0, 0.5, 1.0, 0.0, 1.0, 0.0, 0.0, -1.0, 0.0, -0.25, 4.0, 0.0, 0.0])
    surr = None
    for rule, w in want.items():
        surr, g = P.ppo_surrogate(ratio, adv, clip, rule)
        assert np.array_equal(g, np.array(w, np.float32)), rule
    for rule in ("theano", "both"):
        _, g = P.ppo_surrogate(ratio, adv, clip, rule)
        for build in (lambda r, a: theano_ops.minimum(r * a, theano_ops.clip(r, 1. - clip, 1. + clip) * a, both=rule == "both"),
                      lambda r, a: autograd_ref.ppo_surrogate(r, a, clip, rule)):
            r = torch.from_numpy(ratio).requires_grad_()
            out = build(r, torch.from_numpy(adv))
            out.sum().backward()
            assert np.array_equal(out.detach().numpy(), surr)
            assert np.array_equal(r.grad.numpy(), g), rule
    # theano/tensor/tests/test_basic.py::test_maximum_minimum_grad: at x == y the gradients are [[1], [0]]
    x, y = torch.ones(1, requires_grad=True), torch.ones(1, requires_grad=True)
    theano_ops.minimum(x, y).sum().backward()
    assert x.grad.item() == 1. and y.grad.item() == 0.
    from accel_rl_amd.algos.pg.ppo import PPO
    assert PPO().ppo_tie_rule == "theano" and PPO().loss_tie_rule == 0
    assert PPO(ppo_tie_rule="math").loss_tie_rule == 1 and PPO(ppo_tie_rule="both").loss_tie_rule == 2
    with pytest.raises(ValueError):
        PPO(ppo_tie_rule="torch")


def test_committed_pmc_records_belong_to_the_current_kernels():
    """bench.py quotes HBM traffic from the committed rocprofv3 --pmc records (it cannot read counters itself) and
    drops a record whose source hash is not the current kernel file's -- round 3 shipped a stale one and the driver's
    line carried `traffic: null`.  A kernel edit must be followed by tools/refresh_profiles.sh (or tools/env_step_pmc.sh
    / the gae passes alone) and a commit of profiles/*.json; this test is what says so."""
    import hashlib
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for rec, key, src in (("gae_pmc_traffic.json", "scan_hip_sha1", ("scan.hip",)),
                          ("env_step_pmc.json", "env_hip_sha1", ("env.hip", "env_dev.h"))):
        with open(os.path.join(root, "profiles", rec)) as f:
            have = json.load(f)[key]
        h = hashlib.sha1()
        for name in src:
            with open(os.path.join(root, "accel_rl_amd", "csrc", name), "rb") as f:
                h.update(f.read())
        want = h.hexdigest()
        assert have == want, "profiles/%s was measured on another %s: re-run the PMC passes and commit the record" % (rec, src)


def test_runner_pins_itself_to_gpu_cpus():
    """accel_rl/runners/accel_rl_base.py:71-72: p.cpu_affinity(affinities.get("gpu_cpus", <unchanged>))."""
    import os
    from accel_rl_amd.runners.accel_rl import AccelRLBase
    before = os.sched_getaffinity(0)
    r = AccelRLBase.__new__(AccelRLBase)
    try:
        r.affinities = dict(gpu=0)
        assert r.pin_master() is None and os.sched_getaffinity(0) == before
        one = sorted(before)[-1]
        r.affinities = dict(gpu=0, gpu_cpus=(one, 10 ** 6))            # a CPU that does not exist is dropped
        assert r.pin_master() == [one] and os.sched_getaffinity(0) == {one}
        os.sched_setaffinity(0, before)
        r.affinities = dict(gpu_cpus=(10 ** 6,))
        assert r.pin_master() is None and os.sched_getaffinity(0) == before
    finally:
        os.sched_setaffinity(0, before)


def test_padded_action_set_for_a_shared_suite_head():
    """BASELINE config 4 (one policy for eight games): minimal action set + NOOP padding; the oracle's env port pads
    the same way; None keeps the reference's minimal set (atari_env.py:42-43)."""
    from accel_rl_amd.envs.synthetic_atari import GAMES, SynthAtariEnv, padded_action_set
    from oracle.ref_port import PortedAtariEnv
    assert padded_action_set([0, 1, 3, 4], None) == [0, 1, 3, 4]
    assert padded_action_set([0, 1, 3, 4], 6) == [0, 1, 3, 4, 0, 0]
    with pytest.raises(ValueError):
        padded_action_set(list(range(18)), 6)
    for game in GAMES:
        env = SynthAtariEnv(game=game, pad_actions_to=18, rng=np.random.RandomState(0))
        port = PortedAtariEnv(game=game, pad_actions_to=18, rng=np.random.RandomState(0))
        assert env.action_space.n == 18 and list(env.action_set) == list(port.action_set)
        assert list(env.action_set[:len(GAMES[game][1])]) == GAMES[game][1]
        assert SynthAtariEnv(game=game, rng=np.random.RandomState(0)).action_space.n == len(GAMES[game][1])


def test_diagnostics_ring_outlives_the_runners_log_interval():
    """A replayed learner hands out VIEWS of a device ring; the runner keeps them until its next log line
    (accel_rl.py:55-74: up to log_interval_steps // sample_size iterations).  The runner tells the algorithm that
    interval and the ring is sized from it (round-4 advice: 80-step batches with a 1e6-step interval = 12 500
    iterations against a fixed 4096-slot ring silently aliased newer iterations)."""
    from accel_rl_amd.algos.pg.a2c import A2C
    algo = A2C()
    assert algo._ring_slots() == algo.INFO_RING == 4096
    algo.set_log_interval_itrs(12500)
    assert algo._ring_slots() == 12502
    algo.set_log_interval_itrs(10)
    assert algo._ring_slots() == 4096
