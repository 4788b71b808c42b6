"""Synthetic fixed-frame Atari: the host-side description of the emulator that
runs on the GPU (accel_rl_amd/csrc/env.hip).  BASELINE.json's north_star replaces
the third-party ALE emulator by "synthetic fixed-frame Atari batches"; the
emulator's specification is restated in oracle/synth_ale.py (test infra) and
tests pin the device implementation to it.

`SynthAtariEnv` keeps the constructor signature of the reference's AtariEnv
(accel_rl/envs/atari_env.py:18-26) so `EnvCls=SynthAtariEnv, env_args=dict(game=...)`
drops into a sampler constructor.  It is a *descriptor*: stepping happens only
on the device, batched over all envs of a sampler (there is no CPU env loop in
the product).  Constructing it consumes the same numpy global-RNG draws the
reference constructor does (emulator phase, start no-ops), so the master RNG
stream stays aligned with the reference (DESIGN.md "RNG streams").
"""
import numpy as np

from accel_rl_amd.spaces import Discrete, UintBox, EnvSpec

K_FRAMES = 64
RAW_H, RAW_W = 210, 160
OBS_H, OBS_W = 104, 80          # reference: atari_env.py:13
LIFE_PERIOD = 251

# game -> (game_id, minimal action set as ALE action codes, start_lives)
GAMES = {
    "pong": (0, [0, 1, 3, 4, 11, 12], 0),
    "breakout": (1, [0, 1, 3, 4], 5),
    "seaquest": (2, list(range(18)), 4),
    "space_invaders": (3, [0, 1, 3, 4, 11, 12], 3),
    "qbert": (4, [0, 1, 2, 3, 4, 5], 4),
    "beam_rider": (5, [0, 1, 2, 3, 4, 6, 7, 11, 12], 3),
    "enduro": (6, [0, 1, 3, 4, 5, 8, 9, 11, 12], 0),
    "ms_pacman": (7, [0, 2, 3, 4, 5, 6, 7, 8, 9], 3),
}


def frame_bank(game_id):
    """u8[K,210,160] deterministic frame bank (uploaded once to HBM)."""
    rs = np.random.RandomState(1000 + game_id)
    return rs.randint(0, 256, size=(K_FRAMES, RAW_H, RAW_W), dtype=np.uint8)


def draw_phase(rng):
    """Emulator construction: one phase draw (oracle/synth_ale.py: loadROM)."""
    return int(rng.randint(0, K_FRAMES))


def draw_noops(rng, max_start_noops, size=None):
    """AtariEnv.reset's np.random.randint(0, max_start_noops + 1) (atari_env.py:97)."""
    return rng.randint(0, max_start_noops + 1, size=size)


def padded_action_set(action_set, pad_actions_to):
    """One action space for a suite of games that share a policy (BASELINE config 4: eight games, one synchronous
    clique, ONE flat gradient bucket -- every runner must build the same head): the game's minimal action set followed
    by NOOP (ALE code 0) up to `pad_actions_to` entries.  None = the minimal set, as the reference (atari_env.py:42-43)."""
    action_set = list(action_set)
    if pad_actions_to is None:
        return action_set
    if pad_actions_to < len(action_set):
        raise ValueError("pad_actions_to=%d is smaller than the game's %d actions" % (pad_actions_to, len(action_set)))
    return action_set + [0] * (int(pad_actions_to) - len(action_set))


class SynthAtariEnv(object):
    batched_device_env = True    # protocol marker checked by GpuVecSampler

    def __init__(self, game="pong", frame_skip=4, num_img_obs=4, clip_reward=True,
                 episodic_lives=True, max_start_noops=30, repeat_action_probability=0.,
                 rng=None, pad_actions_to=None, resample="box2x"):
        if game not in GAMES:
            raise IOError("You asked for game {} but it is not one of {}".format(
                game, sorted(GAMES)))
        if repeat_action_probability != 0.:
            raise NotImplementedError("sticky actions are not modelled by the synthetic emulator")
        rng = np.random if rng is None else rng
        self.game = game
        self.game_id, self.action_set, self.start_lives = GAMES[game]
        self.action_set = padded_action_set(self.action_set, pad_actions_to)
        self.frame_skip = int(frame_skip)
        self.num_img_obs = int(num_img_obs)
        self.clip_reward = bool(clip_reward)
        self.episodic_lives = bool(episodic_lives)
        self.max_start_noops = int(max_start_noops)
        self.repeat_action_probability = repeat_action_probability
        if resample not in ("box2x", "nearest"):
            raise ValueError("resample must be 'box2x' (what atari_env.py:155 computes: the constant sits in cv2.resize's "
                             "dst slot, so INTER_LINEAR runs) or 'nearest' (what it names)")
        self.resample = resample
        self._action_space = Discrete(len(self.action_set))
        self._observation_space = UintBox(shape=(self.num_img_obs, OBS_H, OBS_W), bits=8)
        self.phase = draw_phase(rng)                    # emulator construction
        draw_noops(rng, self.max_start_noops)           # constructor's reset (atari_env.py:63)

    action_space = property(lambda self: self._action_space)
    observation_space = property(lambda self: self._observation_space)
    spec = property(lambda self: EnvSpec(self._observation_space, self._action_space))

    @property
    def env_info_keys(self):
        """Keys of the reference env's `info` dict (atari_env.py:73-77,186)."""
        keys = []
        if self.clip_reward:
            keys.append("raw_reward")
        if self.episodic_lives:
            keys.append("need_reset")
        return keys
