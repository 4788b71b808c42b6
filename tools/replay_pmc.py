"""HBM traffic of arl_replay_extract (config 5's bandwidth-bound kernel) from rocprofv3 --pmc passes.
  probe:      python tools/replay_pmc.py probe [batch]           (run under rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE)
  summarise:  python tools/replay_pmc.py summarise fetch.csv write.csv [batch] > replay_extract_pmc.json
The probe builds the 1M-transition prioritized store of `bench.py --workload catdqn` (256 envs, 8.3 GB of frames) and
launches the extract of `batch` transitions (= 2 x batch stacked observations) 30 times on random indices -- the geometry
of bench.py::replay_roofline.  FETCH_SIZE is doubled as /opt/skills/guides/MI355X_MICROARCH.md prescribes for 16-byte-per-lane
coalesced reads (what the kernel's copies are)."""
import csv
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def probe(batch):
    import numpy as np
    import torch
    from accel_rl_amd import _lib
    from accel_rl_amd.algos.dqn.replay_buffers.prioritized import PrioritizedReplayBuffer
    dev = "cuda:0"

    class _Space(object):
        shape = (4, 104, 80)

    class _Spec(object):
        observation_space = _Space()
    n_env, t = 256, 4
    buf = PrioritizedReplayBuffer(alpha=0.6, beta_initial=0.4, default_priority=1., env_spec=_Spec(), size=1000000,
                                  reward_horizon=3, sampling_horizon=t, n_environments=n_env, discount=0.99, device=dev)
    gen = torch.Generator(device=dev).manual_seed(11)
    buf.frames.random_(0, 256, generator=gen)                     # every frame slot holds data (no blank-history zeros)
    stack = int(np.prod(buf.frames.shape[2:])) * buf.num_img_obs
    e_idx = torch.randint(0, n_env, (batch,), dtype=torch.int32, device=dev, generator=gen)
    s_idx = torch.randint(0, buf.env_replay_size - 8, (batch,), dtype=torch.int32, device=dev, generator=gen)
    shape = (batch, buf.num_img_obs) + tuple(buf.frames.shape[2:])
    obs, nxt = (torch.empty(shape, dtype=torch.uint8, device=dev) for _ in range(2))
    a, r, tm = (torch.empty(batch, dtype=torch.uint8, device=dev), torch.empty(batch, device=dev),
                torch.empty(batch, dtype=torch.uint8, device=dev))
    for _ in range(30):
        _lib.replay_extract(buf._rb, e_idx, s_idx, obs, nxt, a, r, tm)
    torch.cuda.synchronize()
    print("probe: %d launches, algorithmic bytes per launch %d" % (30, batch * (2 * 2 * stack + 6)))


def median_of(path, counter):
    vals = []
    with open(path) as f:
        for row in csv.DictReader(f):
            if "extract_kernel" in row["Kernel_Name"] and row["Counter_Name"] == counter:
                vals.append(float(row["Counter_Value"]))
    vals.sort()
    return vals[len(vals) // 2], len(vals)


def summarise(fetch_csv, write_csv, batch):
    fetch_kb, n = median_of(fetch_csv, "FETCH_SIZE")
    write_kb, _ = median_of(write_csv, "WRITE_SIZE")
    fetch_b, write_b = int(fetch_kb * 1024 * 2), int(write_kb * 1024)
    stack = 4 * 104 * 80
    algo = batch * (2 * 2 * stack + 6)
    with open(os.path.join(ROOT, "accel_rl_amd", "csrc", "replay.hip"), "rb") as f:
        sha = hashlib.sha1(f.read()).hexdigest()
    print(json.dumps(dict(
        batch=batch, kernel="extract_kernel (arl_replay_extract)", FETCH_SIZE_KB=fetch_kb, WRITE_SIZE_KB=write_kb,
        fetch_bytes_corrected=fetch_b, write_bytes=write_b, hbm_bytes_per_launch=fetch_b + write_b,
        algorithmic_bytes_per_launch=algo, traffic_over_algorithmic=round((fetch_b + write_b) / algo, 4),
        replay_hip_sha1=sha,
        note="rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `python tools/replay_pmc.py probe`; "
             "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports 1/2 of wide coalesced reads); median over %d "
             "launches; algorithmic = per transition two stacked observations read and written + 6 B of scalars" % n), indent=1))


if __name__ == "__main__":
    if sys.argv[1] == "probe":
        probe(int(sys.argv[2]) if len(sys.argv) > 2 else 4096)
    else:
        summarise(sys.argv[2], sys.argv[3], int(sys.argv[4]) if len(sys.argv) > 4 else 4096)
