"""Launch time of every contraction kernel of the spec-1 network at the PPO minibatch (bench.py's mfma table: 20 launches
back to back in one hipGraph) on each route of arl_conv_geom.route -- nine / six bf16-split products, and ONE product on
rounded operands (ARL_CONV_ROUTE_BF16).  The one-product column is the kernels' NON-matrix time (loads, LDS traffic,
barriers, prologue / epilogue) plus a ninth of the products: what is left of a launch when the matrix pipe is nearly idle.
usage: python tools/route_kernel_times.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import bench                                  # noqa: E402
from accel_rl_amd import _lib                 # noqa: E402
from accel_rl_amd.util import logger          # noqa: E402

if __name__ == "__main__":
    logger.set_quiet(True)
    dev = "cuda:0"
    cols = {}
    for mode in (9, 6, 1):
        _lib.set_conv_precision(mode)
        runner, sampler, algo, policy = bench.build_workload(dev, 0, 0, 1, bench.GAME, True)
        t = bench.mfma_table(dev, policy)
        cols[mode] = {r["kernel"]: r for r in t["kernels"]}
        clk = t.get("sustained_clock_ghz")
        runner.shutdown()
        del runner, sampler, algo, policy
        torch.cuda.synchronize()
    _lib.set_conv_precision(9)
    print("%-34s %9s %9s %9s   %s" % ("kernel (us per launch, in a graph)", "split9", "split6", "bf16 x1", "products at the bf16 peak (9): us"))
    tot = {9: 0.0, 6: 0.0, 1: 0.0}
    for name in cols[9]:
        r9 = cols[9][name]
        pipe9 = r9["flops_per_launch"] * r9.get("products_per_multiply", 0) / 2.5e15 * 1e6
        print("%-34s %9.2f %9.2f %9.2f   %6.2f" % (name, r9["avg_launch_us"], cols[6][name]["avg_launch_us"],
                                                  cols[1][name]["avg_launch_us"], pipe9))
        for m in tot:
            tot[m] += cols[m][name]["avg_launch_us"]
    print("%-34s %9.2f %9.2f %9.2f   (sustained clock %s GHz)" % ("sum", tot[9], tot[6], tot[1], clk))
