"""
ORACLE / TEST INFRASTRUCTURE -- not part of the shipped product path.

SynthALE: a deterministic "synthetic fixed-frame Atari" emulator with the
`atari_py.ALEInterface` call surface that the reference's AtariEnv uses
(/root/reference/accel_rl/envs/atari_env.py:31-46,69-71,94,98,149,166,173-186).

The real ALE (atari-py, unpinned in the reference's requirements.txt:2) and its
ROMs are third-party and absent; BASELINE.json's north_star asks for "synthetic
fixed-frame Atari batches" instead.  This file is the *specification* of that
synthetic emulator.  The same constants are restated in
  - oracle/arl_oracle.c            (C restatement, CPU baseline)
  - accel_rl_amd/csrc/emu.hip      (the device implementation, the product)
  - accel_rl_amd/envs/synthetic_atari.py (host-side constants + frame bank upload)
and tests pin all of them against each other and against the golden vectors
produced by running the *real* reference AtariEnv / sampler on top of this
class (tests/golden/gen_golden.py).

Emulator definition (all integer arithmetic):
  state   : tick (emulator frames since reset_game), lives, over, phase
  phase   : np.random.randint(0, K) drawn ONCE in loadROM from the constructing
            process's global numpy RNG (decorrelates emulators; the product
            replays the same draw on the host, see DESIGN.md "RNG streams")
  screen  : bank[(phase + tick) mod K]                   u8[210,160]
  act(a)  : if over: return 0 (no advance)
            tick += 1
            r = +(1 + tick mod 3) if (7*tick + 3*a) mod 97 == 0 else 0
            r -= 1                if (5*tick +   a) mod 89 == 0
            if start_lives > 0 and tick mod life_period == 0:
                lives -= 1 ; over = (lives == 0)
            if start_lives == 0 and tick >= life_period * 5: over = True
            return r
  bank    : np.random.RandomState(1000 + game_id).randint(0, 256, (K,210,160), uint8)
"""

import numpy as np

K_FRAMES = 64
RAW_H, RAW_W = 210, 160
LIFE_PERIOD = 251

# game -> (game_id, minimal action set (ALE action codes), start_lives)
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

_BANK_CACHE = {}


def frame_bank(game_id):
    """u8[K,210,160] deterministic frame bank for a game."""
    if game_id not in _BANK_CACHE:
        rs = np.random.RandomState(1000 + game_id)
        _BANK_CACHE[game_id] = rs.randint(
            0, 256, size=(K_FRAMES, RAW_H, RAW_W), dtype=np.uint8)
    return _BANK_CACHE[game_id]


def reward_fn(tick, a):
    r = 0
    if (7 * tick + 3 * a) % 97 == 0:
        r += 1 + tick % 3
    if (5 * tick + a) % 89 == 0:
        r -= 1
    return r


def get_game_path(game):
    """atari_py.get_game_path stand-in: any existing path (atari_env.py:31-34)."""
    if game not in GAMES:
        return "/nonexistent/rom/" + str(game)
    # an existing path that still encodes the game: (game_id + 1) slashes == "/"
    return "/" * (GAMES[game][0] + 1)


class SynthALE(object):
    """Drop-in for atari_py.ALEInterface as used by the reference AtariEnv."""

    def __init__(self):
        self._game = None
        self.tick = 0
        self._lives = 0
        self._over = False

    # -- configuration -------------------------------------------------------
    def setFloat(self, key, value):
        pass

    def loadROM(self, game_path):
        # the reference passes the path from get_game_path(): decode the game
        game_id = len(game_path) - 1
        game = [g for g, v in GAMES.items() if v[0] == game_id][0]
        self._game = game
        self.game_id, self.action_set, self.start_lives = GAMES[game]
        self.bank = frame_bank(self.game_id)
        self.phase = int(np.random.randint(0, K_FRAMES))
        self.reset_game()

    def getMinimalActionSet(self):
        return np.array(self.action_set, dtype=np.int32)

    # -- emulation -----------------------------------------------------------
    def reset_game(self):
        self.tick = 0
        self._lives = self.start_lives
        self._over = False

    def act(self, a):
        if self._over:
            return 0
        self.tick += 1
        r = reward_fn(self.tick, int(a))
        if self.start_lives > 0:
            if self.tick % LIFE_PERIOD == 0:
                self._lives -= 1
                if self._lives == 0:
                    self._over = True
        elif self.tick >= LIFE_PERIOD * 5:
            self._over = True
        return r

    def lives(self):
        return self._lives

    def game_over(self):
        return self._over

    def screen_index(self):
        return (self.phase + self.tick) % K_FRAMES

    def getScreenGrayscale(self, buf=None):
        frame = self.bank[self.screen_index()]
        if buf is None:
            return frame.reshape(RAW_H, RAW_W, 1).copy()
        buf[:] = frame.reshape(RAW_H, RAW_W, 1)
        return buf
