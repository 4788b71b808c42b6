"""conv 1 from the u8 observations at the PPO minibatch, a few launches of the forward and of the weight gradient (for
rocprofv3 --pmc passes: tools/conv1_pmc.sh).  usage: python tools/conv1_u8_probe.py [reps]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from accel_rl_amd import _lib  # noqa: E402

DEV = "cuda:0"
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
_lib.load()
obs = torch.randint(0, 256, (1280, 4, 104, 80), device=DEV, dtype=torch.int32).to(torch.uint8)
idx = torch.randperm(1280, device=DEV)[:512].to(torch.int32)
geom = _lib.conv_geom(512, 104, 80, 4, 32, 8, 8, 4, 0, 0)
w = torch.randn(32, 4, 8, 8, device=DEV) / 16
bias = torch.zeros(32, device=DEV)
y = torch.empty(512, 25, 19, 32, device=DEV)
dy = torch.randn(512, 25, 19, 32, device=DEV) * (torch.rand(512, 25, 19, 32, device=DEV) < 0.5)
dw, db = torch.empty_like(w), torch.empty(32, device=DEV)
ws = _lib.conv_workspace(DEV)
folds = _lib.FoldList()
for _ in range(reps):
    _lib.conv2d_u8_fwd(obs, idx, float(np.float32(1. / 255.)), w, bias, y, geom, True)
    folds.conv2d_u8_bwd_weight(dy, obs, idx, float(np.float32(1. / 255.)), dw, geom, ws, dbias=db)
    folds.run()
torch.cuda.synchronize()
