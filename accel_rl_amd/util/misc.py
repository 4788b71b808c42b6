"""Small host-side helpers mirroring accel_rl/util/misc.py (attr-dict `struct`,
byte-size formatting, time-jitter seed)."""
import time


class struct(dict):
    """dict whose keys are also attributes (reference: accel_rl/util/misc.py:3-31).
    `copy()` duplicates the container chain (struct/dict/list) but shares leaves."""

    def __init__(self, **kwargs):
        super().__init__(**kwargs)
        self.__dict__ = self

    def copy(self):
        return struct(**{k: _dup(v) for k, v in self.items()})


def _dup(obj):
    if isinstance(obj, struct):
        return obj.copy()
    if isinstance(obj, dict):
        return {k: _dup(v) for k, v in obj.items()}
    if isinstance(obj, list):
        return [_dup(v) for v in obj]
    return obj


def nbytes_unit(nbytes):
    """(value, unit) with unit in KB/MB/GB (reference: util/misc.py:34-39)."""
    value, unit = float(nbytes), "B"
    for unit in ("KB", "MB", "GB"):
        value /= 1024.
        if value < 1000:
            break
    return value, unit


def make_seed():
    """A seed in [0, 10000) from wall-clock jitter (reference: util/misc.py:42-65)."""
    span = 10000
    now = time.time()
    a = int(now * span) % span
    b = int(now * span * span) % span
    time.sleep(1e-3 * b / span)
    t1 = time.time()
    t1 = int((t1 - int(t1)) * span * 1e3) % span
    time.sleep(1e-3 * a / span)
    t2 = time.time()
    t2 = int((t2 - int(t2)) * span * 1e4) % span
    return (t2 - t1) % span


def graph_capture_mode():
    """capture_error_mode for torch.cuda.graph: once a torch.distributed process group exists, its watchdog thread may
    query events at any time -- legal next to a capture only in thread_local mode (a global-mode capture is invalidated
    by ANY other thread's HIP call; seen as 'operation failed due to a previous error during capture')."""
    import torch.distributed as dist
    return "thread_local" if dist.is_available() and dist.is_initialized() else "global"
