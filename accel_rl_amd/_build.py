"""Build libaccel_rl_hip.so (hand-written HIP, gfx950) in-tree with hipcc.

The shared object lives next to this file so that it travels with a repo
snapshot; it is git-ignored.  hipcc cross-compiles without a GPU present.
"""
import os
import shutil
import subprocess

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG_DIR)
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_PATH = os.path.join(PKG_DIR, "libaccel_rl_hip.so")
SOURCES = ["mfma_conv_p1.hip", "mfma_conv_p2.hip", "mfma_conv_p3.hip", "mfma_conv_p4.hip", "mfma_conv_p5.hip", "mfma_conv_p6.hip",
           "mfma_conv_p7.hip", "mfma_conv.hip", "img_conv.hip", "batch_ops.hip", "scan.hip", "env.hip", "serve_step.hip", "optim.hip", "learner.hip", "replay.hip",
           "dqn.hip", "lstm.hip", "gru.hip"]
ARCH = "gfx950"


def _flags_stamp():
    return os.path.join(CSRC, "_obj", ".flags")


def _stale():
    if not os.path.exists(LIB_PATH):
        return True
    extra = os.environ.get("ARL_HIPCC_FLAGS", "")           # a library built with other (development) flags is stale
    stamp = _flags_stamp()
    if os.path.exists(stamp) and open(stamp).read().split("||")[-1] != extra:
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)]
    deps += [os.path.join(ROOT, "include", h) for h in ("accel_rl_hip.h", "accel_rl_hip_dev.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build_extension(force=False, verbose=False):
    """Compile every HIP source for gfx950 into one shared library.  Serialised by a file lock:
    under torch.distributed.run every rank calls this at start-up."""
    if not force and not _stale():
        return LIB_PATH
    import fcntl
    with open(os.path.join(PKG_DIR, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not _stale():          # another rank built it while we waited
                return LIB_PATH
            return _build_locked(verbose)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build_locked(verbose):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build libaccel_rl_hip.so")
    objs = []
    build_dir = os.path.join(PKG_DIR, "csrc", "_obj")
    os.makedirs(build_dir, exist_ok=True)
    common = [hipcc, "--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC",
              "-ffp-contract=off",           # numpy-exact arithmetic (no implicit fma)
              "-I", os.path.join(ROOT, "include"), "-I", CSRC]
    extra = os.environ.get("ARL_HIPCC_FLAGS", "").split()     # development only (e.g. -DARL_NO_SPLIT6: fewer kernels)
    common += extra
    stamp = _flags_stamp()
    flags_now = " ".join(common) + "||" + os.environ.get("ARL_HIPCC_FLAGS", "")
    flags_same = os.path.exists(stamp) and open(stamp).read() == flags_now
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers += [os.path.join(ROOT, "include", h) for h in ("accel_rl_hip.h", "accel_rl_hip_dev.h")]
    newest_header = max(os.path.getmtime(h) for h in headers)
    procs = []
    for src in SOURCES:
        obj = os.path.join(build_dir, src.replace(".hip", ".o"))
        objs.append(obj)
        path = os.path.join(CSRC, src)
        if flags_same and os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(path), newest_header):
            continue                                # object is current: only the changed sources recompile
        cmd = common + ["-c", path, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("hipcc failed on %s:\n%s" % (src, out.decode(errors="replace")))
    tmp = LIB_PATH + ".tmp.%d" % os.getpid()        # link aside, then rename: readers never see a partial file
    cmd = [hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", tmp] + objs
    out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if out.returncode != 0:
        raise RuntimeError("link failed:\n%s" % out.stdout.decode(errors="replace"))
    os.replace(tmp, LIB_PATH)
    with open(stamp, "w") as f:
        f.write(flags_now)
    return LIB_PATH


if __name__ == "__main__":
    print(build_extension(force=True, verbose=True))
