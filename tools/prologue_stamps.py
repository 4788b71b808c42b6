"""Where a contraction workgroup's prologue goes (a library built with ARL_HIPCC_FLAGS=-DARL_PROLOGUE_STAMPS):
cycles from the kernel's first instruction to the first tile's loads ISSUED (row decode, descriptors, offsets), from there
to the tile split and stored (load latency + split), from there past the first barrier.  Split-route forward kernels.
usage: python tools/ab_lib.py <stamped lib.so> tools/prologue_stamps.py [batch]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from accel_rl_amd import _lib

DEV = "cuda:0"


def main():
    b = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    layers = [("conv1", 104, 80, 4, 32, 8, 4, 0), ("conv2", 25, 19, 32, 64, 4, 2, 1),
              ("conv3", 12, 9, 64, 64, 3, 1, 1), ("dense", 1, 1, 6912, 512, 1, 1, 0)]
    ws = _lib.conv_workspace(DEV)
    lib = _lib.load()
    for op in ("fwd", "dgrad"):
        for name, h, w, c, k, ks, st, p in layers:
            if name == "conv1":
                continue
            geom = _lib.conv_geom(b, h, w, c, k, ks, ks, st, p, p)
            ho, wo = _lib.conv_out_hw(geom)
            x = torch.randn(b, h, w, c, device=DEV).relu()
            wt = torch.randn(k, ks, ks, c, device=DEV) * 0.05
            bias = torch.randn(k, device=DEV)
            y = torch.empty(b, ho, wo, k, device=DEV)
            dy, dx = torch.randn_like(y), torch.empty_like(x)
            scratch = torch.empty(64 << 20, device=DEV)

            def launch():
                if op == "fwd":
                    _lib.conv2d_fwd(x, wt, bias, y, geom, True, ws)
                else:
                    _lib.conv2d_bwd_data(dy, wt, x, dx, geom)
            for cold in (False, True):
                for _ in range(3):
                    launch()
                if cold:
                    scratch.fill_(1.0)             # 256 MB through the caches: the operands come from HBM
                torch.cuda.synchronize()
                tr = torch.zeros(8192 * 8, dtype=torch.int64, device=DEV)
                lib.arl_dev_conv_trace_buffer(tr.data_ptr())
                launch()
                torch.cuda.synchronize()
                lib.arl_dev_conv_trace_buffer(None)
                t = tr.cpu().numpy().reshape(-1, 8)
                t = t[t[:, 0] != 0]
                if not len(t) or t[:, 6].max() < t[:, 0].min():
                    continue
                med = lambda v: int(np.median(v))                       # noqa: E731
                print("%s %s B=%d %s: %d workgroups | to loads issued %d | loads landed + split + LDS stores %d | first barrier %d | "
                      "= prologue %d | loop %d | epilogue %d cycles (medians)" %
                      (name, op, b, "operands COLD" if cold else "operands warm", len(t), med(t[:, 6] - t[:, 0]), med(t[:, 7] - t[:, 6]),
                       med(t[:, 1] - t[:, 7]), med(t[:, 1] - t[:, 0]), med(t[:, 2] - t[:, 1]), med(t[:, 3] - t[:, 2])))


main()
