import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    # The plain-PyTorch references of the kernel tests (F.conv2d and its autograd) run on ATen's own convolution
    # (im2col + rocBLAS), NOT on MIOpen: on a fresh box MIOpen has no tuning database and compiles / searches kernels
    # on first use, and that search aborted the interpreter (SIGABRT inside torch.autograd.grad, conv backward of a
    # random geometry) in two of three first runs on fresh boxes.  The product never calls MIOpen.
    try:
        import torch
        torch.backends.cudnn.enabled = False
    except ImportError:
        pass


def pytest_sessionfinish(session, exitstatus):
    session.config._arl_exitstatus = int(exitstatus)


@pytest.hookimpl(trylast=True)
def pytest_unconfigure(config):
    """Once the device was used, leave WITHOUT interpreter finalisation (after every other plugin has finished and the
    summary is out): hundreds of hipGraphs, RCCL communicators and the ctypes-loaded library are otherwise destroyed in
    whatever order finalisation picks, against a HIP runtime that tears itself down through its own exit handlers --
    one run in a dozen on a fresh box ended in a core dump or a hang AFTER the last test had passed, which turns a
    green run into a failed one.  The exit status is pytest's own."""
    torch = sys.modules.get("torch")
    if torch is None or not torch.cuda.is_available() or not torch.cuda.is_initialized():
        return
    if any(k.startswith(("ROCPROF", "ROCTRACER")) for k in os.environ):      # a profiler writes its output at exit
        return
    try:
        torch.cuda.synchronize()
    except Exception:
        pass
    sys.stdout.flush()
    sys.stderr.flush()
    os._exit(getattr(config, "_arl_exitstatus", 0))


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


@pytest.fixture(scope="session")
def golden():
    return load_golden
