"""Run directories (the contract of accel_rl/util/logging.py:11-49, which the reference's plotting and launch
tooling read):

    <root>/<name>_<run_ID>/progress.csv   tabular log
                           debug.log      text log
                           params.json    the launch parameters + name + run_ID
                           itr_N.pkl / params.pkl   snapshots, by snapshot_mode

`with logger_context(log_dir, name, run_ID, log_params, snapshot_mode) as exp_dir: runner.train()`.
A `log_dir` outside the log root is re-rooted under <LOG_DIR>/local/<yyyymmdd>/ as the reference does with
rllab.config.LOG_DIR; here the root is $ACCEL_RL_LOG_DIR (default ./data).  The outputs are detached again when the
block is left by an exception too."""
import contextlib
import datetime
import json
import os

from accel_rl_amd.util import logger

LOG_DIR = os.path.abspath(os.environ.get("ACCEL_RL_LOG_DIR", os.path.join(os.getcwd(), "data")))
FILES = dict(tabular="progress.csv", text="debug.log", params="params.json")


def make_log_dir(experiment_name, sub_name=None):
    """<LOG_DIR>/local/<yyyymmdd>/<experiment_name>[/<sub_name>]"""
    parts = [LOG_DIR, "local", datetime.date.today().strftime("%Y%m%d"), experiment_name]
    return os.path.join(*(parts + ([sub_name] if sub_name is not None else [])))


class RunDir(object):
    """One run's directory and the logger outputs that live in it."""

    def __init__(self, log_dir, name, run_ID):
        root = os.path.abspath(log_dir)
        inside = os.path.commonpath([root, LOG_DIR]) == LOG_DIR
        self.tag = "{}_{}".format(name, run_ID)
        self.path = os.path.join(root if inside else make_log_dir(log_dir), self.tag)
        self.name, self.run_ID = name, run_ID

    def file(self, kind):
        return os.path.join(self.path, FILES[kind])

    def write_params(self, log_params):
        record = dict(log_params or ())
        record.update(name=self.name, run_ID=self.run_ID)
        os.makedirs(self.path, exist_ok=True)
        with open(self.file("params"), "w") as f:
            json.dump(record, f)

    def attach(self, snapshot_mode):
        logger.set_snapshot_mode(snapshot_mode)
        logger.set_snapshot_dir(self.path)
        logger.add_text_output(self.file("text"))
        logger.add_tabular_output(self.file("tabular"))
        logger.push_prefix(self.tag + " ")

    def detach(self):
        logger.remove_tabular_output(self.file("tabular"))
        logger.remove_text_output(self.file("text"))
        logger.pop_prefix()


@contextlib.contextmanager
def logger_context(log_dir, name, run_ID, log_params=None, snapshot_mode="none"):
    run = RunDir(log_dir, name, run_ID)
    run.attach(snapshot_mode)
    try:
        run.write_params(log_params)
        yield run.path
    finally:
        run.detach()
