"""Can RCCL's all-reduce be captured into a hipGraph through torch.distributed (world size 1, backend nccl)?
Tries: blocking all_reduce inline, async all_reduce + wait, each under thread_local / relaxed capture modes."""
import os, sys, traceback
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
import torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0), rank=0, world_size=1)
x = torch.ones(1 << 20, device="cuda")
y = torch.zeros_like(x)
for _ in range(3):
    dist.all_reduce(x); w = dist.all_reduce(x, async_op=True); w.wait()
torch.cuda.synchronize()
for mode in ("thread_local", "relaxed", "global"):
    for kind in ("blocking", "async"):
        try:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode=mode):
                y.copy_(x).mul_(2.0)
                if kind == "blocking":
                    dist.all_reduce(y)
                else:
                    w = dist.all_reduce(y, async_op=True)
                    x.add_(1.0)                 # work on the capturing stream while the collective runs
                    w.wait()
                y.add_(1.0)
            x.fill_(3.0); g.replay(); torch.cuda.synchronize()
            print("capture %-12s %-8s ok: y[0] = %.1f (want 7), x[0] = %.1f" % (mode, kind, y[0].item(), x[0].item()), flush=True)
        except Exception as e:
            print("capture %-12s %-8s FAILED: %s" % (mode, kind, str(e).split("\n")[0][:200]), flush=True)
            torch.cuda.synchronize()
dist.destroy_process_group()
