"""Runner interface (reference: accel_rl/runners/base.py:2-5)."""


class Runner(object):

    def train(self):
        raise NotImplementedError
