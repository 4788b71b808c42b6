"""Constructor-argument capture (reference: accel_rl/util/quick_args.py:10-25)."""
import inspect


def save_args(values, underscore=False):
    """Store every __init__ argument found in `values` (normally `vars()`) on
    `values['self']`, walking the MRO so subclass + base arguments are kept."""
    obj = values["self"]
    prefix = "_" if underscore else ""
    for cls in type(obj).__mro__:
        init = cls.__dict__.get("__init__")
        if init is None or not inspect.isfunction(init):
            continue
        params = list(inspect.signature(init).parameters.values())[1:]
        for name in [p.name for p in params if p.kind == p.POSITIONAL_OR_KEYWORD]:
            if name in values:
                setattr(obj, prefix + name, values[name])


class Bunch(object):
    def __init__(self, d):
        self.__dict__.update(d)


def retrieve_args(obj, bunch=True):
    args = {k.lstrip("_"): v for k, v in vars(obj).items()}
    return Bunch(args) if bunch else args
