"""Learner cost per row by minibatch size (efficiency must not fall with the batch): one
forward + losses + backward pass of the policy (no optimiser step) inside a hipGraph, for the PPO network (spec 1)
and the A2C network (spec 0).  usage: python tools/batch_sweep.py [rows ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__; __graft_entry__.build()
from accel_rl_amd import _lib
from accel_rl_amd.policies.atari_cnn_policy import AtariCnnPolicy
from accel_rl_amd.policies.atari_cnn_specs import cnn_specs
from accel_rl_amd.spaces import Discrete, UintBox, EnvSpec
DEV = "cuda:0"
sizes = [int(x) for x in sys.argv[1:]] or [256, 512, 1024, 2048, 4096, 5120]
n = max(sizes)
obs = torch.randint(0, 256, (n, 4, 104, 80), device=DEV, dtype=torch.int32).to(torch.uint8)
lr = torch.ones(1, device=DEV)
for spec, kind in ((1, 1), (0, 0)):
    policy = AtariCnnPolicy(**cnn_specs[spec])
    policy.initialize(EnvSpec(UintBox((4, 104, 80)), Discrete(4)), device=DEV)
    for chunk in (None, "auto"):
        policy.max_rows_per_pass = chunk
        for b in sizes:
            mb = dict(observations=obs, actions=torch.randint(0, 4, (n,), device=DEV, dtype=torch.int32).to(torch.uint8),
                      advantages=torch.randn(n, device=DEV), returns=torch.randn(n, device=DEV),
                      old_prob=torch.full((n, 4), 0.25, device=DEV), valids=None,
                      idx=torch.randperm(n, device=DEV)[:b].to(torch.int32))
            step = lambda: policy.loss_and_grads(mb, kind, 0.2, 1.0, 0.01, lr)       # noqa: E731
            for _ in range(2): step()
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                step()
            g.replay(); torch.cuda.synchronize()
            a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = max(3, 20480 // b)
            a.record()
            for _ in range(reps): g.replay()
            e.record(); torch.cuda.synchronize()
            us = a.elapsed_time(e) / reps * 1e3
            print("spec %d  rows per pass %-5s B = %5d: %8.1f us, %.4f us per row" % (spec, policy.rows_per_pass() or "all", b, us, us / b), flush=True)
            del g
            policy._scratch.clear()
