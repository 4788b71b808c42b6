"""Stand-in for rllab's logger (rllab/misc/logger.py) with the parts the runners and
accel_rl/util/logging.py:logger_context use: timestamped text to stdout and text log files,
prefix stack, tabular key/values dumped as a table + progress.csv, snapshots
(itr_N.pkl / params.pkl; modes all / last / gap / none, rllab/misc/logger.py:319-340)."""
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
_snapshot_gap = 1
_text_outputs = dict()
_quiet = False


def set_quiet(q=True):
    global _quiet
    _quiet = q


def log(msg):
    stamp = datetime.datetime.utcnow().strftime("%Y-%m-%d %H:%M:%S.%f UTC")
    line = "%s | %s%s\n" % (stamp, "".join(_prefixes), msg)
    for f in _text_outputs.values():
        f.write(line)
        f.flush()
    if _quiet:
        return
    sys.stdout.write(line)
    sys.stdout.flush()


def push_prefix(p):
    _prefixes.append(p)


def pop_prefix():
    _prefixes.pop()


def add_text_output(file_name):
    if file_name not in _text_outputs:
        os.makedirs(os.path.dirname(os.path.abspath(file_name)), exist_ok=True)
        _text_outputs[file_name] = open(file_name, "a")


def remove_text_output(file_name):
    f = _text_outputs.pop(file_name, None)
    if f is not None:
        f.close()


def add_tabular_output(file_name):
    global _csv_path, _csv_header
    os.makedirs(os.path.dirname(os.path.abspath(file_name)), exist_ok=True)
    _csv_path, _csv_header = file_name, None


def remove_tabular_output(file_name):
    global _csv_path, _csv_header
    if _csv_path == file_name:
        _csv_path = _csv_header = None


def set_snapshot_dir(dir_name):
    global _snapshot_dir
    _snapshot_dir = dir_name
    if dir_name is not None:
        os.makedirs(dir_name, exist_ok=True)


def get_snapshot_dir():
    return _snapshot_dir


def set_snapshot_mode(mode):
    global _snapshot_mode
    if mode not in ("all", "last", "gap", "none"):
        raise NotImplementedError("snapshot mode %r" % (mode,))
    _snapshot_mode = mode


def get_snapshot_mode():
    return _snapshot_mode


def set_snapshot_gap(gap):
    global _snapshot_gap
    _snapshot_gap = int(gap)


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
    if not _quiet or _text_outputs:
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
    if _snapshot_mode == "gap" and itr % _snapshot_gap != 0:
        return None
    import joblib
    name = "params.pkl" if _snapshot_mode == "last" else "itr_%d.pkl" % itr
    path = os.path.join(_snapshot_dir, name)
    joblib.dump(params, path, compress=3)
    return path


def load_itr_params(path):
    """A snapshot written by save_itr_params: dict(itr, cum_samples, policy_param_values)
    (accel_rl/runners/accel_rl_base.py:108-113).  Resume by passing
    `initial_param_values=snapshot["policy_param_values"]` to the policy constructor, as the
    reference's policies take it (policies/pg/atari_cnn_policy.py:24,54-56)."""
    import joblib
    return joblib.load(path)
