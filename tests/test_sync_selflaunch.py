"""AccelRLSync given the reference's LIST of per-GPU affinities in a plain `python script.py` must start its own worker
runners (MultiGpuRLBase.launch_workers, accel_rl/runners/multigpu_rl_base.py:20-45) -- not silently train on one GPU --
and must refuse a list that contradicts a launcher's WORLD_SIZE."""
import glob
import json
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
SCRIPT = os.path.join(HERE, "sync_selflaunch_script.py")


def _run(tmp_path, n, mode, extra_env=None, timeout=300):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(extra_env or {})
    out = subprocess.run([sys.executable, SCRIPT, str(tmp_path), str(n), mode], capture_output=True, text=True,
                         timeout=timeout, env=env, cwd=os.path.dirname(HERE))
    recs = [json.load(open(f)) for f in sorted(glob.glob(os.path.join(str(tmp_path), "rank*.json")))]
    return out, recs


def _check(out, recs, n):
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-3000:])
    assert out.stdout.count("rank0 done") == 1                      # the calling process IS rank 0 and returns from train()
    assert [r["rank"] for r in recs] == list(range(n))
    assert len(set(r["pid"] for r in recs)) == n                    # n processes, not one
    # (the rendez-vous is a store handed to init_process_group: no rank's os.environ is touched -- a second runner built
    #  later in the same process must not find a stale WORLD_SIZE)
    assert all(r["n_runners"] == n and r["world"] is None for r in recs)
    assert [r["seed"] for r in recs] == [7 + 100 * k for k in range(n)]              # multigpu_rl_base.py:28
    assert [r["sampler_seed"] for r in recs] == [8 + 100 * k for k in range(n)]
    assert len(set(r["n_itr"] for r in recs)) == 1
    assert len(set(r["params_crc"] for r in recs)) == 1, recs       # broadcast + identical averaged updates
    assert len(set(tuple(r["params_head"]) for r in recs)) == 1


def test_list_of_affinities_forks_the_worker_runners(tmp_path):
    out, recs = _run(tmp_path, 2, "fake")
    _check(out, recs, 2)
    assert [r["gpu"] for r in recs] == [0, 1]                       # affinities[rank] (multigpu_rl_base.py:30)


def test_three_runners(tmp_path):
    out, recs = _run(tmp_path, 3, "fake")
    _check(out, recs, 3)


def test_a_dying_worker_ends_the_job(tmp_path):
    """Rank 1 raises in its second batch: rank 0 would wait in the next all-reduce for ever; its monitor ends the job."""
    out, recs = _run(tmp_path, 2, "fake", dict(ARL_TEST_CRASH_RANK="1"), timeout=120)
    assert out.returncode != 0
    assert "injected failure in rank 1" in out.stderr
    assert "rank0 done" not in out.stdout


def test_list_must_match_a_launchers_world_size(monkeypatch):
    from test_sync_gloo import _FakeAlgo, _FakePolicy, _FakeSampler
    from accel_rl_amd.runners.sync import AccelRLSync
    monkeypatch.setenv("WORLD_SIZE", "4")
    monkeypatch.setenv("RANK", "1")
    with pytest.raises(ValueError, match="2 affinities but WORLD_SIZE=4"):
        AccelRLSync(algo=_FakeAlgo(), policy=_FakePolicy(), sampler=_FakeSampler(), n_steps=100, seed=1,
                    affinities=[dict(gpu=0), dict(gpu=1)])
    r = AccelRLSync(algo=_FakeAlgo(), policy=_FakePolicy(), sampler=_FakeSampler(), n_steps=100, seed=1,
                    affinities=[dict(gpu=k) for k in range(4)])
    assert r.rank == 1 and r.n_runners == 4 and r.affinities == dict(gpu=1) and r.seed == 101
    assert r._worker_affinities is None                              # launched ranks fork nothing


def test_a_single_dict_is_one_runner(monkeypatch):
    from test_sync_gloo import _FakeAlgo, _FakePolicy, _FakeSampler
    from accel_rl_amd.runners.sync import AccelRLSync
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    r = AccelRLSync(algo=_FakeAlgo(), policy=_FakePolicy(), sampler=_FakeSampler(), n_steps=100, seed=1,
                    affinities=dict(gpu=0))
    assert r.n_runners == 1 and r._worker_affinities is None
    r = AccelRLSync(algo=_FakeAlgo(), policy=_FakePolicy(), sampler=_FakeSampler(), n_steps=100, seed=1,
                    affinities=[dict(gpu=0)])
    assert r.n_runners == 1 and r._worker_affinities is None and r.affinities == dict(gpu=0)


@pytest.mark.gpu
def test_real_learner_two_self_launched_ranks_on_one_gpu(tmp_path):
    """The product's mPPO / AtariCnnPolicy / GpuVecSampler, two runners forked from one script, both on GPU 0 over gloo
    (development mode): different rollouts (seeds 7 / 107), bit-identical parameters after every synchronous update."""
    out, recs = _run(tmp_path, 2, "real", timeout=600)
    _check(out, recs, 2)
