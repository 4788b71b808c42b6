"""Update-rule selectors.  The reference passes `lasagne.updates.rmsprop` /
`lasagne.updates.adam` callables to its optimizers (accel_rl/algos/pg/a2c.py:26,
ppo.py:27); here they are small descriptors that pick the HIP kernel variant
(csrc/optim.hip) and carry Lasagne's default hyper-parameters
(restated in the reference at optimizers/update_methods_stats.py:11,55)."""
from accel_rl_amd import _lib


class _UpdateMethod(object):
    def __init__(self, name, kernel_id, defaults):
        self.name, self.kernel_id, self.defaults = name, kernel_id, defaults

    def resolve(self, **overrides):
        unknown = set(overrides) - set(self.defaults)
        if unknown:
            raise TypeError("unexpected update_method_args for %s: %s" % (self.name, sorted(unknown)))
        args = dict(self.defaults)
        args.update(overrides)
        return args

    def __repr__(self):
        return "update_methods.%s" % self.name


rmsprop = _UpdateMethod("rmsprop", _lib.OPT_RMSPROP, dict(rho=0.9, epsilon=1e-6))
adam = _UpdateMethod("adam", _lib.OPT_ADAM, dict(beta1=0.9, beta2=0.999, epsilon=1e-8))
