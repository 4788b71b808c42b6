"""One PPO minibatch (forward, heads + losses, backward, update) of the spec-1 policy at B = 512 
timed as 8 minibatches per hipGraph (the learner's shape).  usage: python tools/learner_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__; __graft_entry__.build()
from accel_rl_amd import _lib
from accel_rl_amd.policies.atari_cnn_policy import AtariCnnPolicy
from accel_rl_amd.policies.atari_cnn_specs import cnn_specs
from accel_rl_amd.spaces import Discrete, UintBox, EnvSpec
DEV = "cuda:0"
lib = _lib.load()
policy = AtariCnnPolicy(**cnn_specs[1])
policy.initialize(EnvSpec(UintBox((4, 104, 80)), Discrete(4)), device=DEV)
n = 1280
obs = torch.randint(0, 256, (n, 4, 104, 80), device=DEV, dtype=torch.int32).to(torch.uint8)
mb = dict(observations=obs, actions=torch.randint(0, 4, (n,), device=DEV, dtype=torch.int32).to(torch.uint8),
          advantages=torch.randn(n, device=DEV), returns=torch.randn(n, device=DEV),
          old_prob=torch.full((n, 4), 0.25, device=DEV), valids=None)
idxs = [torch.randperm(n, device=DEV)[:512].to(torch.int32) for _ in range(8)]
lr = torch.ones(1, device=DEV)


def step():
    for ix in idxs:
        policy.loss_and_grads(dict(mb, idx=ix), 1, 0.2, 1.0, 0.01, lr)


def gt(rep=10):
    for _ in range(2): step()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        step()
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(rep): g.replay()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / (8 * rep) * 1e3


for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    print("%.1f us per minibatch (forward + backward, no update)" % gt())
