"""The dense layers' data + weight gradient as ONE launch (arl_conv2d_bwd_pair: bwd_pair_kernel, 128 x 128 tiles, 16-deep
k-tiles) against the two separate launches (weight-gradient fold included), by batch: in-graph timing, six alternating
measurements each, bit-for-bit comparison.  (From the build that carries the size rule on, both columns show the rule's
choice where it picks the separate launches.)   usage: python tools/pair_size_probe.py [batch ...]"""
import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from accel_rl_amd import _lib
from bench import graph_time_ms
DEV = "cuda:0"
for case in [(b, 1, 1, c, k) for (c, k) in ((3456, 256), (6912, 512), (512, 512)) for b in ([int(a) for a in sys.argv[1:]] or (512, 1536, 2048, 2560, 3072, 4096, 5120))]:
    b, h, w, c, k = case
    geom = _lib.conv_geom(b, h, w, c, k, 1, 1, 1, 0, 0)
    ws, ws2 = _lib.conv_workspace(DEV), _lib.conv_workspace(DEV)
    x = torch.randn(b, 1, 1, c, device=DEV).relu()
    wt = torch.randn(k, 1, 1, c, device=DEV) / np.sqrt(c)
    dy = torch.randn(b, 1, 1, k, device=DEV)
    dx, dw = torch.empty_like(x), torch.empty_like(wt)
    dx2, dw2 = torch.empty_like(x), torch.empty_like(wt)
    folds, folds2 = _lib.FoldList(), _lib.FoldList()
    def sep():
        _lib.conv2d_bwd_data(dy, wt, x, dx, geom)
        folds2.conv2d_bwd_weight(dy, x, dw, geom, ws2)
        folds2.run()
    def pair():
        folds.conv2d_bwd_pair(dy, wt, x, dx2, x, dw2, geom, ws)
        folds.run()
    ts = {"sep": [], "pair": []}
    for rep in range(6):
        ts["sep"].append(graph_time_ms(sep) * 1e3)
        ts["pair"].append(graph_time_ms(pair) * 1e3)
    torch.cuda.synchronize()
    print(case, "separate %.1f us  pair %.1f us  same dx %s dw %s" % (np.median(ts["sep"]), np.median(ts["pair"]), torch.equal(dx, dx2), torch.equal(dw, dw2)), flush=True)
