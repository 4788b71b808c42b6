#!/usr/bin/env python
"""G10: outputs of the reference's own accel_rl/scripts/launching/affinities.py (build container only)."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shims  # noqa: E402

ref_shims.install()
from accel_rl.scripts.launching import affinities as A  # noqa: E402

cases = [dict(n_gpu=8, ctx_per_gpu=1, ctx_per_run=1, n_cpu_cores=64, ht_offset=64, n_socket=2),
         dict(n_gpu=8, ctx_per_gpu=1, ctx_per_run=8, n_cpu_cores=128, ht_offset=128, n_socket=2),
         dict(n_gpu=8, ctx_per_gpu=2, ctx_per_run=4, n_cpu_cores=40, ht_offset=40, n_socket=2),
         dict(n_gpu=4, ctx_per_gpu=3, ctx_per_run=1, n_cpu_cores=16),
         dict(n_gpu=1, ctx_per_gpu=1, ctx_per_run=1, n_cpu_cores=8, ht_offset=8),
         dict(n_gpu=2, ctx_per_gpu=1, ctx_per_run=2, n_cpu_cores=20, n_socket=1)]
out = []
for kw in cases:
    code = A.encode_affinity_params(**kw)
    n_slots = kw["n_gpu"] * kw["ctx_per_gpu"] // kw["ctx_per_run"]
    out.append(dict(kwargs=kw, code=code, decoded=A.decode_affinity_params(code),
                    slots=[dict(code=A.prepend_run_slot_code(s, code), affinities=A.get_affinities(A.prepend_run_slot_code(s, code)))
                           for s in range(n_slots)],
                    all=A.build_all_affinities(**kw)))
with open(os.path.join(HERE, "g10_affinities.json"), "w") as f:
    json.dump(out, f, indent=0)
print("wrote g10_affinities.json", os.path.getsize(os.path.join(HERE, "g10_affinities.json")))
