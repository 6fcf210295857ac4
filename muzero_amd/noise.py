"""Counter-based tie-break noise shared by host code, the checker and the HIP engine.

The reference draws ``1e-7 * np.random.uniform(size=A)`` once per PUCT
selection call (`mcts/action_selection.py:46`); the number of calls is
data-dependent, so a batched engine cannot consume a shared stream.  Instead the
uniform for (tree, call index within the search, action) is a pure function:

    u = (splitmix64(splitmix64(seed ^ tree * K) + call * 64 + action) >> 11) * 2**-53

The HIP kernels and the C checker implement exactly this (``mz_noise_u`` in
`muzero_amd/csrc/noise.h`).
"""
from __future__ import annotations

import numpy as np

_M64 = 0xFFFFFFFFFFFFFFFF
_TREE_K = 0xD6E8FEB86659FD93


def splitmix64(z: int) -> int:
    z = (z + 0x9E3779B97F4A7C15) & _M64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
    return z ^ (z >> 31)


def tree_key(seed: int, tree: int) -> int:
    return splitmix64((seed ^ ((tree * _TREE_K) & _M64)) & _M64)


def noise_u(seed: int, tree: int, call: int, action: int) -> float:
    z = splitmix64((tree_key(seed, tree) + call * 64 + action) & _M64)
    return (z >> 11) * (2.0 ** -53)


def noise_row(seed: int, tree: int, call: int, n_actions: int) -> np.ndarray:
    return np.array([noise_u(seed, tree, call, a) for a in range(n_actions)], dtype=np.float64)


def move_seed(seed: int, move: int) -> int:
    """Noise key of the `move`-th search of a run that searches the same trees again and again (self-play):
    a different, reproducible key per move, so that tree b does not replay the uniforms of its previous search."""
    return splitmix64((seed + move * 0xA0761D6478BD642F) & _M64) if move else seed & _M64
