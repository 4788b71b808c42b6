"""Per-workgroup timeline of one fp32-MFMA conv kernel launch (arl_dev_conv_trace_buffer):
how long prologue / main loop / epilogue take in shader clocks, the effective clock,
how the dispatcher spread the workgroups over CUs.  usage: python tools/conv_trace.py [batch]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import collections
import numpy as np
import torch
from accel_rl_amd import _lib

DEV = "cuda:0"


def main():
    b = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    op = sys.argv[2] if len(sys.argv) > 2 else "fwd"          # fwd | wgrad | dgrad
    layers = [("conv1", 104, 80, 4, 32, 8, 4, 0), ("conv2", 25, 19, 32, 64, 4, 2, 1),
              ("conv3", 12, 9, 64, 64, 3, 1, 1), ("dense", 1, 1, 6912, 512, 1, 1, 0)]
    ws = _lib.conv_workspace(DEV)
    lib = _lib.load()
    for name, h, w, c, k, ks, st, p in layers:
        geom = _lib.conv_geom(b, h, w, c, k, ks, ks, st, p, p)
        ho, wo = _lib.conv_out_hw(geom)
        x = torch.randn(b, h, w, c, device=DEV)
        wt = torch.randn(k, ks, ks, c, device=DEV) * 0.05
        bias = torch.randn(k, device=DEV)
        y = torch.empty(b, ho, wo, k, device=DEV)
        dy, dw, dx = torch.randn_like(y), torch.empty_like(wt), torch.empty_like(x)
        if op == "dgrad" and name == "conv1":
            continue

        def launch():
            if op == "fwd":
                _lib.conv2d_fwd(x, wt, bias, y, geom, True, ws)
            elif op == "wgrad":
                _lib.conv2d_bwd_weight(dy, x, dw, geom, ws)
            else:
                _lib.conv2d_bwd_data(dy, wt, x, dx, geom)
        for _ in range(3):
            launch()
        torch.cuda.synchronize()
        tr = torch.zeros(8192 * 8, dtype=torch.int64, device=DEV)
        lib.arl_dev_conv_trace_buffer(tr.data_ptr())
        launch()
        torch.cuda.synchronize()
        lib.arl_dev_conv_trace_buffer(None)
        t = tr.cpu().numpy().reshape(-1, 8)
        t = t[t[:, 0] != 0]
        hw0, xcc0 = t[:, 6], t[:, 7] & 0xf
        cu0 = ((hw0 >> 8) & 0xf) | (((hw0 >> 12) & 1) << 4) | (((hw0 >> 13) & 7) << 5) | (xcc0 << 8)
        for x in np.unique(cu0):                                    # cycle counters are not global: align per CU
            sel = cu0 == x
            t[sel, 0:4] -= t[sel, 0].min() - 1
        n = len(t)
        t0 = t[:, 0].min()
        start, pro, loop, epi = t[:, 0] - t0, t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2]
        end = t[:, 3] - t0
        real = (t[:, 5].max() - t[:, 4].min()) / 100.0          # us (100 MHz)
        clk = (t[:, 3].max() - t0) / real / 1e3                 # GHz if s_memtime ticks at shader clock
        hw, xcc = t[:, 6], t[:, 7] & 0xf
        cu = ((hw >> 8) & 0xf) | (((hw >> 12) & 1) << 4) | (((hw >> 13) & 7) << 5) | (xcc << 8)
        per_cu = collections.Counter(cu.tolist())
        # busy intervals per CU: how many workgroups are inside their main loop at a time
        ev = sorted([(x, 1) for x in (t[:, 1] - t0)] + [(x, -1) for x in (t[:, 2] - t0)])
        area, cur, last = 0, 0, 0
        for x, d in ev:
            area += cur * (x - last)
            cur, last = cur + d, x
        in_loop = area / float(end.max()) / len(per_cu)
        hist = collections.Counter(per_cu.values())
        print("%s %s: %d WGs, wall %.1f us, counter/wall = %.2f ticks/ns; start skew p50/max %d/%d; prologue p50 %d, "
              "loop p50/max %d/%d, epilogue p50 %d, end max %d; CUs used %d, WGs/CU histogram %s" %
              (name, op, n, real, clk, np.median(start), start.max(), np.median(pro), np.median(loop), loop.max(),
               np.median(epi), end.max(), len(per_cu), dict(sorted(hist.items()))))
        rs, re = (t[:, 4] - t[:, 4].min()) / 100.0, (t[:, 5] - t[:, 4].min()) / 100.0      # us, the GPU-wide 100 MHz clock
        wg_clk = np.median((t[:, 3] - t[:, 0]) / np.maximum(t[:, 5] - t[:, 4], 1) / 10.0)
        print("    real time (100 MHz clock): workgroup starts p50/p90/max %.1f/%.1f/%.1f us, ends p10/p50/max %.1f/%.1f/%.1f us; "
              "per-workgroup cycles / ns = %.3f GHz" % (np.median(rs), np.percentile(rs, 90), rs.max(),
                                                        np.percentile(re, 10), np.median(re), re.max(), wg_clk))
        print("    mean workgroups per CU inside the main loop: %.2f; total/median WG lifetime %d" % (in_loop, np.median(t[:, 3] - t[:, 0])))


if __name__ == "__main__":
    main()
