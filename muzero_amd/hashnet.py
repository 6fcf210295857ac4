"""Deterministic "hash network": a synthetic stand-in for a MuZero network.

Every output is a pure integer-hash function of a 32-bit state code and the
action, scaled by powers of two, so the values are exactly representable and
bit-identical whether they are produced here (NumPy/torch, host), by the C
checker or by the HIP engine's built-in synthetic network
(`muzero_amd/csrc/hashnet.h`).  It exposes the two methods the search consumes
from a network (reference `network.py:86-126`): ``initial_inference`` and
``recurrent_inference``, with the same tensor shapes and dtypes
(logits f32 ``(1,A)``, hidden f32 ``(1,H)``, value/reward f64 ``(1,)``).

State code <-> hidden state: ``hidden[0] = code & 0xFFFF``,
``hidden[1] = code >> 16`` (both exact in f32); the other slots are filler
``(hash(code, 1000+i) & 255) / 256``.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

_M32 = 0xFFFFFFFF


def fmix32(h: int) -> int:
    h &= _M32
    h ^= h >> 16
    h = (h * 0x85EBCA6B) & _M32
    h ^= h >> 13
    h = (h * 0xC2B2AE35) & _M32
    h ^= h >> 16
    return h


def hash2(code: int, k: int) -> int:
    return fmix32((code + 0x9E3779B9 * (k + 1)) & _M32)


@dataclass(frozen=True)
class HashNetParams:
    n_actions: int
    hidden_size: int
    logit_shift: int = 5        # logit = ((h % 257) - 128) * 2**-logit_shift
    reward_on: int = 1          # reward = (h % 65) / 64 if on else 0
    parent_predicts: int = 1    # 1: logits/value come from the parent's code
    #                             (what reference network.py:121 does), 0: child's


def logits_of(code: int, p: HashNetParams) -> np.ndarray:
    scale = np.float32(2.0 ** -p.logit_shift)
    return np.array(
        [np.float32((hash2(code, 1 + a) % 257) - 128) * scale for a in range(p.n_actions)],
        dtype=np.float32)


def value_of(code: int) -> float:
    return ((hash2(code, 200) % 513) - 256) / 256.0


def reward_of(code: int, p: HashNetParams) -> float:
    return (hash2(code, 201) % 65) / 64.0 if p.reward_on else 0.0


def child_code(code: int, action: int) -> int:
    return hash2(code, 300 + action)


def hidden_of(code: int, p: HashNetParams) -> np.ndarray:
    h = np.empty(p.hidden_size, dtype=np.float32)
    for i in range(p.hidden_size):
        if i == 0:
            h[i] = code & 0xFFFF
        elif i == 1:
            h[i] = code >> 16
        else:
            h[i] = (hash2(code, 1000 + i) & 0xFF) / 256.0
    return h


def code_of(hidden: np.ndarray) -> int:
    flat = np.asarray(hidden, dtype=np.float32).reshape(-1)
    return (int(flat[0]) | (int(flat[1]) << 16)) & _M32


def observation_of(code: int, shape=(1, 1, 4)) -> np.ndarray:
    """An observation array that carries a root state code in its first two cells."""
    obs = np.zeros(shape, dtype=np.float64)
    flat = obs.reshape(-1)
    flat[0] = code & 0xFFFF
    flat[1] = code >> 16
    return obs


def code_of_observation(obs_row: np.ndarray) -> int:
    """State code of an arbitrary observation: a hash chain over its values rounded to 1/1024."""
    c = 0x12345678
    for i, v in enumerate(np.asarray(obs_row, dtype=np.float64).reshape(-1)):
        c = hash2(c, (int(round(float(v) * 1024)) + 3 * i * 4099) & _M32)
    return c


class HashNetwork:
    """Host-side hash network with the reference's network call interface.  With ``obs_hash`` the
    root code is a hash of the whole observation (for environments whose observations are real
    game states); otherwise it is read from the observation's first two cells."""

    def __init__(self, params: HashNetParams, obs_hash: bool = False) -> None:
        self.params = params
        self.obs_hash = obs_hash
        self.calls = 0

    # -- the two methods `MCTS.search` uses ------------------------------
    def initial_inference(self, observation):
        import torch
        obs = observation.detach().cpu().numpy().reshape(observation.shape[0], -1)
        logits, hidden, value = [], [], []
        for row in obs:
            code = code_of_observation(row) if self.obs_hash else (int(row[0]) | (int(row[1]) << 16)) & _M32
            logits.append(logits_of(code, self.params))
            hidden.append(hidden_of(code, self.params))
            value.append(value_of(code))
        return (torch.from_numpy(np.stack(logits)),
                torch.from_numpy(np.stack(hidden)),
                torch.from_numpy(np.array(value, dtype=np.float64)))

    def recurrent_inference(self, hidden_state, encoded_action):
        import torch
        hs = hidden_state.detach().cpu().numpy().reshape(hidden_state.shape[0], -1)
        ea = encoded_action.detach().cpu().numpy().reshape(encoded_action.shape[0], -1)
        logits, hidden, value, reward = [], [], [], []
        for h, e in zip(hs, ea):
            self.calls += 1
            pcode = code_of(h)
            action = int(np.argmax(e))
            ccode = child_code(pcode, action)
            src = pcode if self.params.parent_predicts else ccode
            logits.append(logits_of(src, self.params))
            value.append(value_of(src))
            reward.append(reward_of(ccode, self.params))
            hidden.append(hidden_of(ccode, self.params))
        return (torch.from_numpy(np.stack(logits)),
                torch.from_numpy(np.stack(hidden)),
                torch.from_numpy(np.array(value, dtype=np.float64)),
                torch.from_numpy(np.array(reward, dtype=np.float64)))
