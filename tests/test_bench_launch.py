"""bench.py's own launch paths (the driver starts it as `python bench.py --gpus N ...`): started without a launcher it
must spawn its N ranks itself (as the reference's runner forks its n - 1 workers, multigpu_rl_base.py:20-45), and the
strong-scaling mode must split the job it is given.  On a one-GPU box N > 1 runs in the development mode (both ranks on
GPU 0, gloo: RCCL refuses two ranks on one device) -- a launch-path check, never a measurement."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, extra_env=None, timeout=600):
    env = dict(os.environ)
    env.update(extra_env or {})
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, cwd=ROOT, env=env,
                          capture_output=True, text=True, timeout=timeout)


def _line(out):
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1, (out.stdout[-2000:], out.stderr[-2000:])      # rank 0 prints ONE line
    return json.loads(lines[0])


def test_spawns_its_ranks_without_a_gpu():
    """No GPU here: the spawned ranks must get as far as the product's refusal to run on a CPU -- i.e. the supervised
    re-exec under torch.distributed.run (127.0.0.1 rendezvous, RANK / WORLD_SIZE in the environment) works -- and the
    supervisor must answer the failed run with exactly ONE retry on eager collectives, then give up with a non-zero exit
    and no JSON line.  (torch.distributed.run ends the other ranks as soon as one exits: the refusal is printed at
    least once per attempt, not necessarily once per rank.)"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("covered by the GPU tests below")
    out = _run(["--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-roofline"])
    assert out.returncode != 0
    text = out.stdout + out.stderr
    assert text.count("bench.py needs a GPU") >= 2, (out.stdout[-2000:], out.stderr[-2000:])      # >= 1 per attempt
    assert text.count("retrying once with eager collectives (ARL_SYNC_GRAPH=0)") == 1
    assert text.count("the eager retry failed too") == 1
    assert not [ln for ln in out.stdout.splitlines() if ln.startswith('{"metric"')]


def test_no_retry_when_the_collectives_are_eager_already():
    import torch
    if torch.cuda.is_available():
        pytest.skip("needs the CPU refusal as the failure")
    out = _run(["--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-roofline"],
               dict(ARL_SYNC_GRAPH="0"))
    assert out.returncode != 0 and "retrying" not in out.stderr and "run failed" in out.stderr


@pytest.mark.gpu
def test_two_ranks_spawned_in_development_mode():
    out = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--no-graph", "--no-cpu-baseline", "--no-roofline"],
               dict(ARL_BENCH_ONE_GPU="1", ARL_BENCH_BACKEND="gloo"))
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-2000:])
    d = _line(out)
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["backend"] == "gloo" and d["scaling"] == "weak"
    assert d["steps"] == 2 and d["config"]["total_envs"] == 512 and d["value"] > 0
    assert d["config"]["parallelism"] == "dp2 sync all-reduce (gloo)"          # the label names the backend that ran
    _check_multi_gpu_keys(d, 2)


def _check_multi_gpu_keys(d, world):
    """The N > 1 line describes itself (bench.py::multi_gpu_diagnostics)."""
    m = d["multi_gpu"]
    for k in ("step", "rollout", "learner"):
        lo, hi = m["per_rank_ms"][k]
        assert 0 < lo <= hi
    assert m["params_bit_identical_across_ranks"] is True       # the synchronous update's invariant, on real ranks
    assert m["graph_captured"] is False                         # development mode: gloo cannot be captured, --no-graph
    assert isinstance(m["allreduce_exposed_ms"], float) and m["learner_without_collectives_ms"] > 0
    assert m["allreduce_bytes_per_update"] > 14e6               # the flat fp32 bucket of the spec-1 policy


@pytest.mark.gpu
def test_eight_ranks_spawned_in_development_mode():
    """The driver's 8-GPU launch shape, all eight ranks on GPU 0 over gloo: rendezvous, per-rank seeds, broadcast, eight
    contributions per all-reduce, the self-describing keys -- a launch-path check, never a measurement."""
    out = _run(["--gpus", "8", "--steps", "2", "--warmup", "1", "--no-graph", "--no-cpu-baseline", "--no-roofline"],
               dict(ARL_BENCH_ONE_GPU="1", ARL_BENCH_BACKEND="gloo"), timeout=900)
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-2000:])
    d = _line(out)
    assert d["n_gpus"] == 8 and d["rccl_ranks"] == 8 and d["backend"] == "gloo"
    assert d["config"]["total_envs"] == 2048 and d["config"]["global_minibatch"] == 4096 and d["value"] > 0
    _check_multi_gpu_keys(d, 8)


@pytest.mark.gpu
def test_strong_scaling_mode_splits_the_job():
    """--scaling strong: --total-envs environments and one global minibatch of 512 x 8 rows per update in all; at N = 1
    the single rank owns all of it."""
    out = _run(["--scaling", "strong", "--total-envs", "2048", "--steps", "2", "--warmup", "1"])
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-2000:])
    d = _line(out)
    assert d["scaling"] == "strong" and d["n_gpus"] == 1 and d["config"]["total_envs"] == 2048
    assert d["config"]["global_minibatch"] == 4096 and d["value"] > 0
    assert "cpu_baseline" not in d and "roofline" not in d


@pytest.mark.gpu
def test_a_hung_capture_still_yields_one_line_on_eager_collectives(tmp_path):
    """The first multi-GPU node this code meets may hang in the capture of its collectives instead of throwing
    (DESIGN 7).  Injected here: the last rank never arrives at its third learner call.  The ranks' watchdogs end the
    attempt (exit 3), the supervisor re-runs once with ARL_SYNC_GRAPH=0, and ONE line comes out that says so.  The
    watchdog also leaves a note on the node, so that a run started by a LAUNCHER afterwards (the driver's N = 4, 8 after
    a failed N = 2: no supervising parent) takes the eager path by itself."""
    out = _run(["--gpus", "2", "--steps", "2", "--warmup", "0", "--no-cpu-baseline", "--no-roofline"],
               dict(ARL_BENCH_ONE_GPU="1", ARL_BENCH_BACKEND="gloo", ARL_BENCH_INJECT="capture_hang",
                    ARL_BENCH_STALL_S="25", TMPDIR=str(tmp_path)), timeout=900)
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-3000:])
    assert "no progress for 25 s in learner (injected capture hang)" in out.stderr
    assert out.stderr.count("retrying once with eager collectives") == 1
    d = _line(out)
    assert d["graph_fallback"].startswith("eager after exit code")
    assert d["multi_gpu"]["graph_captured"] is False
    assert d["multi_gpu"]["params_bit_identical_across_ranks"] is True
    assert d["n_gpus"] == 2 and d["value"] > 0
    # (the note is keyed on this build and this user: another checkout's runs on the node are not affected)
    notes = [f for f in os.listdir(str(tmp_path)) if f.startswith("arl_bench_sync_graph_stalled.%d." % os.getuid())]
    assert len(notes) == 1
    # the driver's launch shape, after the stall: torch.distributed.run around bench.py, nothing injected
    env = dict(os.environ, ARL_BENCH_ONE_GPU="1", ARL_BENCH_BACKEND="gloo", TMPDIR=str(tmp_path))
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "ARL_SYNC_GRAPH"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", "29533", os.path.join(ROOT, "bench.py"), "--gpus", "2",
                          "--steps", "2", "--warmup", "0", "--no-cpu-baseline", "--no-roofline"], cwd=ROOT, env=env,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-3000:])
    d = _line(out)
    assert d["graph_fallback"] == "eager after a capture stall in an earlier run on this node"
    assert d["multi_gpu"]["graph_captured"] is False and d["multi_gpu"]["params_bit_identical_across_ranks"] is True
    assert "this run issues them eagerly" in out.stderr               # the mode flip is announced, not silent


def test_the_stall_note_is_keyed_on_build_and_user_and_can_be_cleared(tmp_path, monkeypatch):
    """(advice r5) The note that switches later multi-rank runs to eager collectives belongs to ONE build and ONE user,
    and a completed captured run removes it (bench.py: clear_stall_marker, called when graph_captured is true)."""
    monkeypatch.setenv("TMPDIR", str(tmp_path))
    import importlib
    import bench
    importlib.reload(bench)
    try:
        name = os.path.basename(bench.STALL_MARKER)
        assert name.startswith("arl_bench_sync_graph_stalled.%d." % os.getuid()) and len(name.split(".")[-1]) >= 5
        assert os.path.dirname(bench.STALL_MARKER) == str(tmp_path)
        assert not bench.stalled_before()
        with open(bench.STALL_MARKER, "w") as f:
            f.write("%d 2 learner\n" % int(__import__("time").time()))
        assert bench.stalled_before()
        bench.clear_stall_marker()
        assert not bench.stalled_before() and not os.path.exists(bench.STALL_MARKER)
        bench.clear_stall_marker()              # idempotent
    finally:
        monkeypatch.undo()
        importlib.reload(bench)
