"""Process plumbing shared by the multi-rank tests: ranks are daemonic and are killed when they overrun -- a rank that
hangs (a collective waiting for a peer that died, a process-group teardown that never returns) fails ITS test after
the deadline instead of keeping the interpreter from exiting at the end of the whole run."""
import os
import threading
import time


def run_ranks(ctx, target, argsets, timeout, before_join=None):
    """Start one daemonic process per argument tuple, (collect what they send with `before_join()`,) wait for all of
    them until `timeout` seconds from the start, kill what is still alive; returns (exit codes, before_join's result)
    -- exit code None = killed here."""
    procs = [ctx.Process(target=target, args=a, daemon=True) for a in argsets]
    for p in procs:
        p.start()
    deadline = time.time() + timeout
    got = None
    try:
        if before_join is not None:
            got = before_join()
    finally:
        for p in procs:
            p.join(max(0.1, deadline - time.time()))
        codes = [p.exitcode for p in procs]
        _reap(procs)
    return codes, got


def _reap(procs):
    for p in procs:
        if p.is_alive():
            p.terminate()
            p.join(5)
            if p.is_alive():
                p.kill()
                p.join(5)


def leave_group(dist, grace=20.0):
    """End of a rank: barrier, then tear the process group down -- with a deadline: results are on disk / in the queue
    by now, and a teardown that blocks (seen with watchdog threads of the `nccl` backend) must not turn a finished rank
    into a hung one."""
    dist.barrier()
    t = threading.Timer(grace, lambda: os._exit(0))
    t.daemon = True
    t.start()
    dist.destroy_process_group()
    t.cancel()
