"""Trajectory statistics container (reference: accel_rl/sampler/util.py:75-101).
The accumulation itself runs on the device (csrc/env.hip: act_step_kernel); this
class only carries completed episodes back to the runner's logging."""
from accel_rl_amd.util.misc import struct


class TrajInfo(struct):
    """Attributes not starting with "_" are logged by the runner."""

    def __init__(self, Length=0, Return=0., RawReturn=0., NonzeroRewards=0,
                 DiscountedReturn=0., **kwargs):
        super().__init__(Length=Length, Return=Return, RawReturn=RawReturn,
                         NonzeroRewards=NonzeroRewards, DiscountedReturn=DiscountedReturn,
                         **kwargs)
