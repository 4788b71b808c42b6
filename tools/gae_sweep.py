"""GAE-scan roofline sweep: N*T in 2^10 .. 2^28, T in {5, 32, 128}; HIP-event time per launch,
algorithmic bytes 17*N*T + 4*N against the 8 TB/s HBM peak.  usage: python tools/gae_sweep.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from accel_rl_amd import _lib

DEV = "cuda:0"


def ev(fn, reps=20, warm=3):
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
    print("GAE scan (arl_gae_scan), fp32 in / f64 carry; us per launch, GB/s algorithmic, fraction of 8 TB/s")
    print("%8s %5s %12s %10s %10s %8s" % ("log2(NT)", "T", "n_env", "us", "GB/s", "frac"))
    gen = torch.Generator(device=DEV).manual_seed(1)
    for t in (5, 32, 128):
        for lg in (10, 14, 18, 20, 22, 24, 26, 28):
            n = max(1, (1 << lg) // t)
            r = torch.randn(n * t, device=DEV, generator=gen)
            v = torch.randn(n * t, device=DEV, generator=gen)
            d = (torch.rand(n * t, device=DEV, generator=gen) < 0.05).to(torch.uint8)
            lv = torch.randn(n, device=DEV, generator=gen)
            adv, ret = torch.empty_like(r), torch.empty_like(r)
            us = ev(lambda: _lib.gae_scan(r, v, d, lv, 0.99, 0.95, n, t, adv, ret))
            us2 = ev(lambda: _lib.gae_scan(r, v, d, lv, 0.99, 0.95, n, t, adv, ret, promo=_lib.PROMO_ASSOC))
            nbytes = 17 * n * t + 4 * n
            gbs, gbs2 = nbytes / us / 1e3, nbytes / us2 / 1e3
            print("%8d %5d %12d %10.1f %10.1f %8.3f   | wave suffix scan (ARL_PROMO_ASSOC) %10.1f us %10.1f GB/s %8.3f" %
                  (lg, t, n, us, gbs, gbs / 8000., us2, gbs2, gbs2 / 8000.))
            del r, v, d, lv, adv, ret


if __name__ == "__main__":
    main()
