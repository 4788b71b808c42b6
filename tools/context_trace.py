"""Per-workgroup timeline (arl_dev_conv_trace_buffer) of ONE conv kernel launch as the learner runs it: inside an eager
PPO minibatch (B = 512, spec 1), i.e. on activations the previous layer has just written, after a long busy stretch.
usage: python tools/context_trace.py [c1f | c2f | c3f | df | c3d | c2d]   (forward conv 1..3 / dense, data gradient
of conv 3 / conv 2)"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__; __graft_entry__.build()
from accel_rl_amd import _lib
from accel_rl_amd.policies.atari_cnn_policy import AtariCnnPolicy
from accel_rl_amd.policies.atari_cnn_specs import cnn_specs
from accel_rl_amd.spaces import Discrete, UintBox, EnvSpec
DEV = "cuda:0"
lib = _lib.load()
policy = AtariCnnPolicy(**cnn_specs[1])
policy.initialize(EnvSpec(UintBox((4, 104, 80)), Discrete(4)), device=DEV)
n = 1280
obs = torch.randint(0, 256, (n, 4, 104, 80), device=DEV, dtype=torch.int32).to(torch.uint8)
mb = dict(observations=obs, actions=torch.randint(0, 4, (n,), device=DEV, dtype=torch.int32).to(torch.uint8),
          advantages=torch.randn(n, device=DEV), returns=torch.randn(n, device=DEV),
          old_prob=torch.full((n, 4), 0.25, device=DEV), valids=None)
idxs = [torch.randperm(n, device=DEV)[:512].to(torch.int32) for _ in range(8)]
lr = torch.ones(1, device=DEV)
tr = torch.zeros(8192 * 8, dtype=torch.int64, device=DEV)
# which: c1f | c2f | c3f | df (forward conv 1..3, dense)  |  c3d | c2d (data gradient of conv 3 / conv 2)
TARGETS = dict(c1f=("u8", 0), c2f=("fwd", 0), c3f=("fwd", 1), df=("fwd", 2), c3d=("pair", 1), c2d=("pair", 2))
kind, nth = TARGETS[sys.argv[1] if len(sys.argv) > 1 else "c2f"]
which = sys.argv[1] if len(sys.argv) > 1 else "c2f"
state = dict(calls=dict(u8=0, fwd=0, pair=0), armed=False)


def wrap(fn, k):
    def wrapped(*a, **kw):
        hit = state["armed"] and kind == k and state["calls"][k] == nth
        state["calls"][k] += 1
        if hit:
            lib.arl_dev_conv_trace_buffer(tr.data_ptr())
        out = fn(*a, **kw)
        if hit:
            lib.arl_dev_conv_trace_buffer(None)
        return out
    return wrapped


_lib.conv2d_fwd = wrap(_lib.conv2d_fwd, "fwd")
_lib.conv2d_u8_fwd = wrap(_lib.conv2d_u8_fwd, "u8")
_lib.FoldList.conv2d_bwd_pair = wrap(_lib.FoldList.conv2d_bwd_pair, "pair")


def report(tag):
    t = tr.cpu().numpy().reshape(-1, 8)
    t = t[t[:, 0] != 0]
    hw, xcc = t[:, 6], t[:, 7] & 0xf
    cu = ((hw >> 8) & 0xf) | (((hw >> 12) & 1) << 4) | (((hw >> 13) & 7) << 5) | (xcc << 8)
    for x in np.unique(cu):
        sel = cu == x
        t[sel, 0:4] -= t[sel, 0].min() - 1
    pro, loop, epi = t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2]
    wall = (t[:, 5].max() - t[:, 4].min()) / 100.0
    clk = np.median((t[:, 3] - t[:, 0]) / np.maximum(t[:, 5] - t[:, 4], 1) / 10.0)
    per_cu = collections.Counter(cu.tolist())
    hist = dict(sorted(collections.Counter(per_cu.values()).items()))
    order = np.argsort(loop)
    idx = np.nonzero(tr.cpu().numpy().reshape(-1, 8)[:, 0])[0]
    print("   loop percentiles 10/50/90/99/max: %s" % [int(np.percentile(loop, q)) for q in (10, 50, 90, 99, 100)])
    by_cu = collections.defaultdict(list)
    for i in range(len(t)):
        by_cu[int(cu[i])].append(i)
    for i in order[-6:]:
        mates = [j for j in by_cu[int(cu[i])] if j != i]
        print("   slow WG block %d: loop %d, prologue %d, cu %#x xcc %d; CU mates' loops %s" %
              (idx[i], loop[i], pro[i], int(cu[i]) & 0xff, int(cu[i]) >> 8, [int(loop[j]) for j in mates]))
    by_n = collections.defaultdict(list)
    for c, members in by_cu.items():
        by_n[len(members)].append(sorted(int(loop[j]) for j in members))
    for k in sorted(by_n):
        arr = np.array(by_n[k])
        print("   CUs with %d WGs (%d): median loop cycles, fastest -> slowest: %s" % (k, len(arr), np.median(arr, axis=0).astype(int).tolist()))
    # one busy CU's timeline (cycles since its first workgroup's start): start, loop begin, loop end, end
    big = max(by_cu, key=lambda c: (len(by_cu[c]), -c))
    rows = sorted((int(t[j, 0]), int(t[j, 1]), int(t[j, 2]), int(t[j, 3])) for j in by_cu[big])
    z = rows[0][0]
    print("   timeline of CU %#x: %s" % (big, [tuple(x - z for x in r_) for r_ in rows]))
    xcc_med = {int(x): int(np.median(loop[(cu >> 8) == x])) for x in np.unique(cu >> 8)}
    print("   median loop by XCD: %s" % xcc_med)
    rs = (t[:, 4] - t[:, 4].min()) / 100.0
    print("%s: %d WGs, wall %.1f us, clock %.3f GHz; cycles p50: prologue %d, loop %d (max %d), epilogue %d, lifetime %d; "
          "starts p50/max %.1f/%.1f us; WGs/CU %s" % (tag, len(t), wall, clk, np.median(pro), np.median(loop), loop.max(),
                                                      np.median(epi), np.median(t[:, 3] - t[:, 0]), np.median(rs), rs.max(), hist))


for choice in (0,):
    for rep in range(3):                       # two warm passes over the 8 minibatches, then the traced one
        for j, ix in enumerate(idxs):
            state["calls"] = dict(u8=0, fwd=0, pair=0)
            state["armed"] = rep == 2 and j == 6
            if state["armed"]:
                tr.zero_()
            policy.loss_and_grads(dict(mb, idx=ix), 1, 0.2, 1.0, 0.01, lr)
    torch.cuda.synchronize()
    report(which)
