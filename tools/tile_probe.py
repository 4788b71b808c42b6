"""64x64 (32x32 MFMA) vs 112x64 and 32x64 (16x16 MFMA) tiles for the 64-column layers: time (20 launches per hipGraph)
and max deviation from the first.  usage: python tools/tile_probe.py [batch ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__; __graft_entry__.build()
from accel_rl_amd import _lib
DEV = "cuda:0"
CHOICES = (1, 2, 3)
NAMES = {1: "64x64", 2: "112x64 (3 stages)", 3: "32x64 (5 waves / SIMD)"}


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


lib = _lib.load()
ws = _lib.conv_workspace(DEV)
for b in [int(x) for x in sys.argv[1:]] or [512, 256]:
    for name, h, w, c, k, ks, st, p in [("conv2", 25, 19, 32, 64, 4, 2, 1), ("conv3", 12, 9, 64, 64, 3, 1, 1),
                                        ("dense (forward + split fold)", 1, 1, 6912, 512, 1, 1, 0)]:
        geom = _lib.conv_geom(b, h, w, c, k, ks, ks, st, p, p)
        ho, wo = _lib.conv_out_hw(geom)
        x = torch.randn(b, h, w, c, device=DEV); wt = torch.randn(k, ks, ks, c, device=DEV) * 0.05
        bias = torch.randn(k, device=DEV); dy = torch.randn(b, ho, wo, k, device=DEV)
        flops = 2.0 * b * ho * wo * k * ks * ks * c
        outs = {}
        for choice in CHOICES:
            lib.arl_conv_tile_choice(choice)
            y = torch.empty(b, ho, wo, k, device=DEV); dx = torch.empty_like(x)
            tf = gt(lambda: _lib.conv2d_fwd(x, wt, bias, y, geom, True, ws))
            td = gt(lambda: _lib.conv2d_bwd_data(dy, wt, None, dx, geom)) if st == 1 and h > 1 else float("nan")
            outs[choice] = (y.clone(), dx.clone())
            print("B=%d %s tiles=%s fwd %.1f us (%.1f TF/s)  dgrad %.1f us (%.1f TF/s)" %
                  (b, name, NAMES[choice], tf, flops / tf / 1e6, td, flops / td / 1e6))
        lib.arl_conv_tile_choice(0)
        for c in CHOICES[1:]:
            d = (outs[1][0] - outs[c][0]).abs().max().item() / outs[1][0].abs().max().item()
            print("   %s vs 64x64: fwd rel dev %.2e" % (NAMES[c], d), " dgrad rel dev %.2e" % ((outs[1][1] - outs[c][1]).abs().max().item() / max(outs[1][1].abs().max().item(), 1e-9)) if st == 1 and h > 1 else "")
