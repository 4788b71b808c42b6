"""Minimal stand-in for rllab's logger (rllab/misc/logger.py): timestamped text,
prefix stack, tabular key/values dumped as a table + progress.csv, snapshot
hook.  Only what the runner on the hot path calls."""
import contextlib
import csv
import datetime
import os
import sys

import numpy as np

_prefixes = []
_tabular = []
_csv_path = None
_csv_header = None
_snapshot_dir = None
_snapshot_mode = "none"
_quiet = False


def set_quiet(q=True):
    global _quiet
    _quiet = q


def log(msg):
    if _quiet:
        return
    stamp = datetime.datetime.utcnow().strftime("%Y-%m-%d %H:%M:%S.%f UTC")
    sys.stdout.write("%s | %s%s\n" % (stamp, "".join(_prefixes), msg))
    sys.stdout.flush()


@contextlib.contextmanager
def prefix(p):
    _prefixes.append(p)
    try:
        yield
    finally:
        _prefixes.pop()


def record_tabular(key, value):
    _tabular.append((str(key), value))


def record_tabular_misc_stat(key, values):
    """Average/Std/Median/Min/Max (rllab/misc/logger.py:439-457)."""
    if len(values) > 0:
        v = np.asarray(values, dtype=np.float64)
        stats = (np.average(v), np.std(v), np.median(v), np.min(v), np.max(v))
    else:
        stats = (np.nan,) * 5
    for name, s in zip(("Average", "Std", "Median", "Min", "Max"), stats):
        record_tabular(key + name, s)


def set_output(log_dir=None, snapshot_mode="none"):
    global _csv_path, _csv_header, _snapshot_dir, _snapshot_mode
    _snapshot_mode = snapshot_mode
    _csv_header = None
    if log_dir is None:
        _csv_path = _snapshot_dir = None
        return
    os.makedirs(log_dir, exist_ok=True)
    _csv_path = os.path.join(log_dir, "progress.csv")
    _snapshot_dir = log_dir


def dump_tabular(with_prefix=False):
    global _csv_header
    rows = list(_tabular)
    del _tabular[:]
    if not rows:
        return dict()
    if not _quiet:
        width = max(len(k) for k, _ in rows)
        for k, v in rows:
            log("%-*s  %s" % (width, k, v))
    if _csv_path is not None:
        keys = [k for k, _ in rows]
        new_file = _csv_header != keys
        with open(_csv_path, "a" if not new_file or _csv_header is not None else "w", newline="") as f:
            w = csv.writer(f)
            if new_file:
                w.writerow(keys)
                _csv_header = keys
            w.writerow([v for _, v in rows])
    return dict(rows)


def save_itr_params(itr, params):
    """Snapshot modes all/last/gap/none (rllab/misc/logger.py:319-340)."""
    if _snapshot_dir is None or _snapshot_mode == "none":
        return None
    import joblib
    name = "itr_%d.pkl" % itr if _snapshot_mode == "all" else "params.pkl"
    path = os.path.join(_snapshot_dir, name)
    joblib.dump(params, path, compress=3)
    return path
