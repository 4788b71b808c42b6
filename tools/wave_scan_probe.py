"""The wave suffix scan (ARL_PROMO_ASSOC) at 2^26 elements by segment groups per wave (arl_dev_scan_wave_groups) next to the
exact walk, fraction of the 8 TB/s HBM peak.  usage: python tools/wave_scan_probe.py [log2 elements]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from accel_rl_amd import _lib

DEV = "cuda:0"


def ev(fn, reps=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s = [torch.cuda.Event(enable_timing=True) for _ in range(reps)]
    e = [torch.cuda.Event(enable_timing=True) for _ in range(reps)]
    for i in range(reps):
        s[i].record(); fn(); e[i].record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in zip(s, e)])) * 1e3


def main():
    lg = int(sys.argv[1]) if len(sys.argv) > 1 else 26
    lib = _lib.load()
    gen = torch.Generator(device=DEV).manual_seed(1)
    lib.arl_dev_scan_force_wave(1)
    for t in (5, 16, 32, 64, 128, 256, 512):
        n = (1 << lg) // t
        r = torch.randn(n * t, device=DEV, generator=gen)
        v = torch.randn(n * t, device=DEV, generator=gen)
        d = (torch.rand(n * t, device=DEV, generator=gen) < 0.05).to(torch.uint8)
        lv = torch.randn(n, device=DEV, generator=gen)
        adv, ret = torch.empty_like(r), torch.empty_like(r)
        nbytes = 17 * n * t + 4 * n
        us = ev(lambda: _lib.gae_scan(r, v, d, lv, 0.99, 0.95, n, t, adv, ret))
        line = "T = %3d  exact walk %7.1f us %.3f |" % (t, us, nbytes / us / 1e3 / 8000.)
        for g in (1, 2, 4, 0):
            lib.arl_dev_scan_wave_groups(g)
            us = ev(lambda: _lib.gae_scan(r, v, d, lv, 0.99, 0.95, n, t, adv, ret, promo=_lib.PROMO_ASSOC))
            line += "  groups %d: %7.1f us %.3f" % (g, us, nbytes / us / 1e3 / 8000.)
        print(line)
        del r, v, d, lv, adv, ret
    lib.arl_dev_scan_force_wave(0)


if __name__ == "__main__":
    main()
