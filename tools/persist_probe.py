"""Persistent launches (arl_conv_persistent) of the many-tile kernels at the PPO minibatch: conv 1 forward from u8
rows and the stride-2 data gradient of conv 2, time per launch (20 launches per hipGraph) for 0 (one workgroup per
tile) and 2 .. 6 resident workgroups per CU; then the whole minibatch.  usage: python tools/persist_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__; __graft_entry__.build()
from accel_rl_amd import _lib
DEV = "cuda:0"
lib = _lib.load()


def gt(fn, per=20, rep=5):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(per): fn()
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(rep): g.replay()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / (per * rep) * 1e3


B = 512
obs = torch.randint(0, 256, (1280, 4, 104, 80), device=DEV, dtype=torch.int32).to(torch.uint8)
idx = torch.randperm(1280, device=DEV)[:B].to(torch.int32)
g1 = _lib.conv_geom(B, 104, 80, 4, 32, 8, 8, 4, 0, 0)
w1, b1 = torch.randn(32, 4, 8, 8, device=DEV) * 0.05, torch.randn(32, device=DEV)
y1 = torch.empty(B, 25, 19, 32, device=DEV)
g2 = _lib.conv_geom(B, 25, 19, 32, 64, 4, 4, 2, 1, 1)
dy2, w2 = torch.randn(B, 12, 9, 64, device=DEV), torch.randn(64, 4, 4, 32, device=DEV) * 0.05
dx2 = torch.empty(B, 25, 19, 32, device=DEV)
f1 = 2.0 * B * 25 * 19 * 32 * 256
f2 = 2.0 * B * 12 * 9 * 64 * 16 * 32
WAYS = [int(x) for x in sys.argv[1:]] or [0, 2, 3, 4, 5, 6, 0]
TC = int(os.environ.get("ARL_TC", "0"))
for w in WAYS:
    lib.arl_conv_persistent(w)
    lib.arl_conv_tile_choice(TC)
    t1 = gt(lambda: _lib.conv2d_u8_fwd(obs, idx, 1. / 255, w1, b1, y1, g1, True))
    t2 = gt(lambda: _lib.conv2d_bwd_data(dy2, w2, y1, dx2, g2))
    print("persistent %d: conv1 fwd (u8) %.1f us (%.1f TF/s)   conv2 dgrad %.1f us (%.1f TF/s)" %
          (w, t1, f1 / t1 / 1e6, t2, f2 / t2 / 1e6))
lib.arl_conv_persistent(0)
