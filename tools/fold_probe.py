"""The ONE fold launch of a PPO minibatch's backward pass (arl_fold_many: conv 1 / conv 2 / conv 3 weight gradients, their
bias sums, the head's partials, the loss sums) by the split count from which an output is summed by 64 threads instead
of 16 (arl_dev_fold_wide_from): in-graph timing of the recorded items, 20 launches per graph.
usage: python tools/fold_probe.py [minibatch rows]"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__; __graft_entry__.build()
from accel_rl_amd import _lib
from accel_rl_amd.policies.atari_cnn_policy import AtariCnnPolicy
from accel_rl_amd.policies.atari_cnn_specs import cnn_specs
from accel_rl_amd.spaces import Discrete, UintBox, EnvSpec
from bench import graph_time_ms

DEV = "cuda:0"


def main():
    b = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    n = 1280
    lib = _lib.load()
    policy = AtariCnnPolicy(**cnn_specs[1])
    policy.initialize(EnvSpec(UintBox((4, 104, 80)), Discrete(4)), device=DEV)
    obs = torch.randint(0, 256, (n, 4, 104, 80), device=DEV, dtype=torch.int32).to(torch.uint8)
    mb = dict(observations=obs, actions=torch.randint(0, 4, (n,), device=DEV, dtype=torch.int32).to(torch.uint8),
              advantages=torch.randn(n, device=DEV), returns=torch.randn(n, device=DEV),
              old_prob=torch.full((n, 4), 0.25, device=DEV), valids=None,
              idx=torch.randperm(n, device=DEV)[:b].to(torch.int32))
    recorded = []
    run = _lib.FoldList.run

    def recording_run(self, stream=None):
        items = (_lib.ArlFoldItem * _lib.FOLD_MAX_ITEMS)()
        C.memmove(items, self._items, C.sizeof(items))
        recorded.append((items, self._n))
        return run(self, stream)
    _lib.FoldList.run = recording_run
    policy.loss_and_grads(mb, 1, 0.2, 1.0, 0.01, torch.ones(1, device=DEV))
    torch.cuda.synchronize()
    _lib.FoldList.run = run
    items, count = max(recorded, key=lambda r: r[1])
    print("fold items of one backward pass (minibatch %d): %s" %
          (b, ", ".join("%d x %d splits" % (items[i].total, items[i].splits) for i in range(count))))
    mbytes = sum(items[i].total * items[i].splits * 4 for i in range(count)) / 1e6
    for wide in (1 << 30, 256, 128, 64, 32, 1):
        lib.arl_dev_fold_wide_from(wide)
        ms = graph_time_ms(lambda: _lib._check(lib.arl_fold_many(items, count, _lib.stream_ptr(None)), "arl_fold_many"))
        print("64 threads per output from %10s splits on: %6.2f us  (%.1f MB of partials, %.2f TB/s)" %
              ("never" if wide == 1 << 30 else wide, ms * 1e3, mbytes, mbytes / ms / 1e6), flush=True)
    lib.arl_dev_fold_wide_from(0)


main()
