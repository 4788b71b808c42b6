"""Launch ONE conv / dense kernel of the spec-1 trunk a few times (for rocprofv3 --pmc passes).
usage: python tools/one_kernel.py <conv1|conv2|conv3|dense> <fwd|dgrad|wgrad|pair> [batch] [reps]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from accel_rl_amd import _lib

DEV = "cuda:0"
LAYERS = dict(conv1=(104, 80, 4, 32, 8, 4, 0), conv2=(25, 19, 32, 64, 4, 2, 1), conv3=(12, 9, 64, 64, 3, 1, 1),
              dense=(1, 1, 6912, 512, 1, 1, 0))


def main():
    name, op = sys.argv[1], sys.argv[2]
    b = int(sys.argv[3]) if len(sys.argv) > 3 else 512
    reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
    lib = _lib.load()
    h, w, c, k, ks, st, p = LAYERS[name]
    geom = _lib.conv_geom(b, h, w, c, k, ks, ks, st, p, p)
    ho, wo = _lib.conv_out_hw(geom)
    ws = _lib.conv_workspace(DEV)
    x = torch.randn(b, h, w, c, device=DEV).relu()
    wt = torch.randn(k, ks, ks, c, device=DEV) / np.sqrt(ks * ks * c)
    bias = torch.randn(k, device=DEV)
    dy = torch.randn(b, ho, wo, k, device=DEV)
    y, dx, dw = torch.empty(b, ho, wo, k, device=DEV), torch.empty_like(x), torch.empty_like(wt)
    for _ in range(reps):
        if op == "fwd":
            _lib.conv2d_fwd(x, wt, bias, y, geom, True, ws)
        elif op == "dgrad":
            _lib.conv2d_bwd_data(dy, wt, x, dx, geom)
        elif op == "wgrad":
            _lib.conv2d_bwd_weight(dy, x, dw, geom, ws)
        else:
            folds = _lib.FoldList()
            folds.conv2d_bwd_pair(dy, wt, x, dx, x, dw, geom, ws)
            folds.run()
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
