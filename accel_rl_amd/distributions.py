"""Categorical distribution over action indices (reference:
accel_rl/distributions/categorical.py).  `*_sym` methods are the torch (autograd)
counterparts of the reference's Theano graph builders; the others are host/
device array versions.  TINY as in categorical.py:6."""
import torch

TINY = 1e-8


class Categorical(object):

    def __init__(self, dim):
        self._dim = dim

    dim = property(lambda self: self._dim)
    dist_info_keys = property(lambda self: ["prob"])

    @staticmethod
    def _pick(prob, actions):
        return prob.gather(1, actions.long().view(-1, 1)).squeeze(1)

    def kl_sym(self, old_dist_info, new_dist_info):                     # categorical.py:35-45
        old, new = old_dist_info["prob"], new_dist_info["prob"]
        return torch.sum(old * (torch.log(old + TINY) - torch.log(new + TINY)), dim=-1)

    kl = kl_sym

    def likelihood_ratio_sym(self, actions, old_dist_info, new_dist_info):   # :66-70
        return (self._pick(new_dist_info["prob"], actions) + TINY) / \
               (self._pick(old_dist_info["prob"], actions) + TINY)

    def entropy_sym(self, dist_info):                                   # :76-78
        p = dist_info["prob"]
        return -torch.sum(p * torch.log(p + TINY), dim=1)

    entropy = entropy_sym

    def log_likelihood_sym(self, actions, dist_info):                   # :86-88
        return torch.log(self._pick(dist_info["prob"], actions) + TINY)

    log_likelihood = log_likelihood_sym
