#!/usr/bin/env python
"""G15: outputs of the reference's own accel_rl.optimizers.util.iterate_traj_idxs (build container only)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shims  # noqa: E402

ref_shims.install()
from accel_rl.optimizers.util import iterate_traj_idxs  # noqa: E402

out = dict()
cases = [(40, 80, 5, 1), (20, 20, 5, 2), (512, 2048, 8, 3), (6, 18, 1, 4)]
for ci, (bs, n, horizon, seed) in enumerate(cases):
    np.random.seed(seed)
    for ep in range(2):
        got = list(iterate_traj_idxs(bs, n, horizon=horizon, shuffle=True))
        out["c%d_e%d_idx" % (ci, ep)] = np.stack([g[0] for g in got])
        out["c%d_e%d_trajs" % (ci, ep)] = np.stack([g[1] for g in got])
    got = list(iterate_traj_idxs(bs, n, horizon=horizon, shuffle=False))
    out["c%d_plain_idx" % ci] = np.stack([g[0] for g in got])
    out["c%d_plain_trajs" % ci] = np.stack([g[1] for g in got])
    out["c%d_after" % ci] = np.random.randint(0, 2 ** 31 - 1, size=2)     # RNG stream position afterwards
    out["c%d_cfg" % ci] = np.array([bs, n, horizon, seed])
out["n_cases"] = np.array(len(cases))
np.savez_compressed(os.path.join(HERE, "g15_trajidx.npz"), **out)
print("wrote g15_trajidx.npz", os.path.getsize(os.path.join(HERE, "g15_trajidx.npz")))
