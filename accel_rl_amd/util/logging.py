"""Run directories and logger set-up (reference: accel_rl/util/logging.py:11-49).

`with logger_context(log_dir, name, run_ID, log_params, snapshot_mode): runner.train()` writes
<log_dir>/<name>_<run_ID>/{progress.csv, debug.log, params.json} (+ itr_N.pkl / params.pkl
snapshots), the files the reference's plotting and launch tooling read.  The reference roots
relative directories under rllab.config.LOG_DIR/local/<yyyymmdd>/; here the root is
$ACCEL_RL_LOG_DIR (default ./data)."""
import datetime
import json
import os
from contextlib import contextmanager

from accel_rl_amd.util import logger

LOG_DIR = os.path.abspath(os.environ.get("ACCEL_RL_LOG_DIR", os.path.join(os.getcwd(), "data")))


def make_log_dir(experiment_name, sub_name=None):
    yyyymmdd = datetime.datetime.today().strftime("%Y%m%d")
    log_dir = os.path.join(LOG_DIR, "local", yyyymmdd, experiment_name)
    return log_dir if sub_name is None else os.path.join(log_dir, sub_name)


@contextmanager
def logger_context(log_dir, name, run_ID, log_params=None, snapshot_mode="none"):
    logger.set_snapshot_mode(snapshot_mode)
    abs_log_dir = os.path.abspath(log_dir)
    if LOG_DIR != os.path.commonpath([abs_log_dir, LOG_DIR]):
        abs_log_dir = make_log_dir(log_dir)
    exp_dir = os.path.join(abs_log_dir, "{}_{}".format(name, run_ID))
    tabular_log_file = os.path.join(exp_dir, "progress.csv")
    text_log_file = os.path.join(exp_dir, "debug.log")
    params_log_file = os.path.join(exp_dir, "params.json")
    logger.set_snapshot_dir(exp_dir)
    logger.add_text_output(text_log_file)
    logger.add_tabular_output(tabular_log_file)
    logger.push_prefix("{}_{} ".format(name, run_ID))
    log_params = dict(log_params or dict())
    log_params["name"] = name
    log_params["run_ID"] = run_ID
    with open(params_log_file, "w") as f:
        json.dump(log_params, f)
    try:
        yield exp_dir
    finally:
        logger.remove_tabular_output(tabular_log_file)
        logger.remove_text_output(text_log_file)
        logger.pop_prefix()
