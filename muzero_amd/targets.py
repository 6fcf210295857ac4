"""Finished games on the device and the training batches sampled from them.

Host-side mirror of what the reference does with the records the search produces:
`GameHistory` lists (games/game.py:77-118) become flat device arrays, and the per-game body of
`ReplayBuffer.sample` (replay_buffer.py:59-105: stacked observation, unrolled actions with absorbing
padding, `GameHistory.make_target`) runs as one kernel over all sampled positions
(`mz_make_targets`).  Sampling draws from `np.random` in the reference's order.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Sequence

import numpy as np
import torch

from . import engine as _engine
from .engine import _check, _ptr, _stream, lib


class MzTrajectories(C.Structure):
    _fields_ = [('n_games', C.c_int32), ('obs_size', C.c_int32), ('n_actions', C.c_int32), ('reserved', C.c_int32),
                ('game_offset', C.c_void_p), ('observations', C.c_void_p), ('actions', C.c_void_p),
                ('rewards', C.c_void_p), ('to_plays', C.c_void_p), ('root_values', C.c_void_p), ('probs', C.c_void_p),
                ('re_root_values', C.c_void_p), ('re_probs', C.c_void_p), ('has_reanalysed', C.c_void_p),
                ('game_base', C.c_void_p), ('position_stride', C.c_int64)]


class MoveLog:
    """Device log of the moves B environments play (mz_move_log of the C ABI): row k = move k of the current
    collection window.  The fused action + environment kernel appends to it; `TrajectoryStore.from_log` reads the
    finished games out of it in place."""

    def __init__(self, capacity: int, n_envs: int, obs_shape, n_actions: int, device) -> None:
        self.capacity, self.B, self.A = int(capacity), int(n_envs), int(n_actions)
        self.obs_shape = tuple(obs_shape)
        self.obs_size = int(np.prod(self.obs_shape))
        self.device = torch.device(device)
        z = lambda dt, *tail: torch.zeros((self.capacity, self.B) + tail, dtype=dt, device=self.device)  # noqa: E731
        self.observations = z(torch.float32, self.obs_size)
        self.actions, self.to_plays = z(torch.int32), z(torch.int32)
        self.rewards, self.root_values = z(torch.float64), z(torch.float64)
        self.probs = z(torch.float64, self.A)
        self.done = z(torch.uint8)
        # 1 = the environment was in the middle of a game when the window opened (set by the owner of the environments)
        self.open_at_start = torch.zeros(self.B, dtype=torch.uint8, device=self.device)
        self.rows = 0   # rows written in the current window
        p = lambda t: t.data_ptr()  # noqa: E731
        self.struct = _engine.MzMoveLog(self.capacity, self.B, self.obs_size, self.A, p(self.observations), p(self.actions),
                                        p(self.rewards), p(self.to_plays), p(self.root_values), p(self.probs), p(self.done),
                                        p(self.open_at_start))

    def games(self):
        """(first position, length) of every finished game in the written rows, as device tensors (int64, int32):
        environment by environment, in playing order inside an environment."""
        count = torch.empty(self.B, dtype=torch.int32, device=self.device)
        _check(lib().mz_log_count_games(C.byref(self.struct), C.c_int32(self.rows), _ptr(count), _stream()),
               'mz_log_count_games')
        ends = torch.cumsum(count, 0, dtype=torch.int64)
        first = (ends - count).contiguous()
        G = int(ends[-1].item())
        base = torch.empty(G, dtype=torch.int64, device=self.device)
        length = torch.empty(G, dtype=torch.int32, device=self.device)
        if G:
            _check(lib().mz_log_list_games(C.byref(self.struct), C.c_int32(self.rows), _ptr(first), _ptr(base), _ptr(length),
                                           _stream()), 'mz_log_list_games')
        return base, length


def scalar_to_support(x: torch.Tensor, support_limit: int) -> torch.Tensor:
    """Categorical form of value / reward targets, as the trainer builds it (trainer.py:70-71 ->
    utils/network_utils.py:99-129): float32 device tensor (...,) -> (..., 2 * support_limit + 1)."""
    if not x.is_cuda:
        raise _engine.EngineError('scalar_to_support needs a device tensor (there is no CPU fallback)')
    xf = x.to(torch.float32).contiguous()
    out = torch.empty(tuple(xf.shape) + (2 * int(support_limit) + 1,), dtype=torch.float32, device=x.device)
    _check(lib().mz_scalar_to_support(_ptr(xf), C.c_int64(xf.numel()), C.c_int32(int(support_limit)), _ptr(out), _stream()),
           'mz_scalar_to_support')
    return out


class TrajectoryStore:
    """Flat device copy of a list of finished games (objects with the `GameHistory` fields)."""

    def __init__(self, histories: Sequence, n_actions: int, device=None) -> None:
        if not torch.cuda.is_available():
            raise _engine.EngineError('TrajectoryStore needs a GPU (there is no CPU fallback)')
        if len(histories) == 0 or any(len(h.observations) == 0 for h in histories):
            raise ValueError('TrajectoryStore: every game needs at least one recorded move')
        self.device = torch.device(device) if device is not None else torch.device('cuda', torch.cuda.current_device())
        self.A = int(n_actions)
        self.lengths = np.array([len(h.observations) for h in histories], dtype=np.int64)
        self.offsets = np.concatenate([[0], np.cumsum(self.lengths)]).astype(np.int64)
        self.obs_shape = tuple(np.asarray(histories[0].observations[0]).shape)
        self.obs_size = int(np.prod(self.obs_shape))
        T = int(self.offsets[-1])
        cat = lambda xs, dt: np.concatenate([np.asarray(x, dtype=dt).reshape(len(x), -1) for x in xs])  # noqa: E731
        dev = lambda x, dt: torch.as_tensor(np.ascontiguousarray(x), dtype=dt, device=self.device)  # noqa: E731
        self.game_offset = dev(self.offsets, torch.int64)
        self.observations = dev(cat([h.observations for h in histories], np.float64), torch.float32)
        self.actions = dev(cat([h.actions for h in histories], np.int64).reshape(T), torch.int32)
        self.rewards = dev(cat([h.rewards for h in histories], np.float64).reshape(T), torch.float64)
        self.to_plays = dev(cat([h.to_plays for h in histories], np.int64).reshape(T), torch.int32)
        self.root_values = dev(cat([h.root_values for h in histories], np.float64).reshape(T), torch.float64)
        self.probs = dev(cat([h.action_probabilities for h in histories], np.float64), torch.float64)
        self.re_root_values = torch.zeros(T, dtype=torch.float64, device=self.device)
        self.re_probs = torch.zeros(T, self.A, dtype=torch.float64, device=self.device)
        self.has_reanalysed = torch.zeros(len(histories), dtype=torch.uint8, device=self.device)
        for g, h in enumerate(histories):
            if getattr(h, 'reanalysed_action_probabilities', None):
                self.save_reanalysed_stats(g, h.reanalysed_action_probabilities, h.reanalysed_root_values)

    game_base, stride = None, 1

    @classmethod
    def from_log(cls, log: 'MoveLog') -> 'TrajectoryStore':
        """The finished games of a move log, read where they lie: the store's arrays ARE the log's (position i of
        game g at row-major index game_base[g] + i * n_envs), only the game lengths travel to the host (the sampler
        draws positions there, in the reference's `np.random` order).  The log must not be rewound while the store
        is in use."""
        base, length = log.games()
        if base.numel() == 0:
            raise ValueError('TrajectoryStore.from_log: no game has finished inside the log')
        self = cls.__new__(cls)
        self.device, self.A = log.device, log.A
        self.obs_shape, self.obs_size = log.obs_shape, log.obs_size
        self.lengths = length.cpu().numpy().astype(np.int64)
        self.offsets = np.concatenate([[0], np.cumsum(self.lengths)]).astype(np.int64)
        self.game_offset = torch.as_tensor(self.offsets, device=self.device)
        self.game_base, self.stride = base, log.B
        T = log.capacity * log.B
        self.observations = log.observations.view(T, log.obs_size)
        self.actions, self.rewards, self.to_plays = log.actions.view(T), log.rewards.view(T), log.to_plays.view(T)
        self.root_values, self.probs = log.root_values.view(T), log.probs.view(T, log.A)
        self.re_root_values = self.re_probs = None     # allocated by the first save_reanalysed_stats
        self.has_reanalysed = torch.zeros(len(self.lengths), dtype=torch.uint8, device=self.device)
        self._log = log
        return self

    def __len__(self) -> int:
        return len(self.lengths)

    def _positions(self, g: int) -> torch.Tensor:
        """Indices of game g's positions in the per-position arrays."""
        n = int(self.lengths[g])
        if self.game_base is None:
            return torch.arange(int(self.offsets[g]), int(self.offsets[g]) + n, device=self.device)
        return self.game_base[g] + self.stride * torch.arange(n, device=self.device)

    def histories(self, action_encoder, obs_shape=None) -> List:
        """The stored games as host `GameHistory` objects (one copy of the arrays, then slicing)."""
        from .games.game import GameHistory
        shape = tuple(obs_shape) if obs_shape is not None else self.obs_shape
        host = lambda t: t.cpu().numpy()  # noqa: E731
        obs, act, rew, tp = host(self.observations), host(self.actions), host(self.rewards), host(self.to_plays)
        rv, pr = host(self.root_values), host(self.probs)
        base = self.offsets[:-1] if self.game_base is None else host(self.game_base)
        out = []
        for g in range(len(self)):
            idx = int(base[g]) + self.stride * np.arange(int(self.lengths[g]))
            h = GameHistory(action_encoder, obs[idx[0]].reshape(shape))
            h.observations = list(obs[idx].reshape((len(idx),) + shape))
            h.actions = [int(a) for a in act[idx]]
            h.encoded_actions = [action_encoder(a) for a in h.actions]
            h.rewards = [float(r) for r in rew[idx]]
            h.to_plays = [int(x) for x in tp[idx]]
            h.action_probabilities = list(pr[idx])
            h.root_values = [float(v) for v in rv[idx]]
            out.append(h)
        return out

    def save_reanalysed_stats(self, g: int, action_probabilities, root_values) -> None:
        """GameHistory.save_reanalysed_stats (games/game.py:114-118) for game `g`; device or host arrays."""
        if self.re_probs is None:
            self.re_root_values = torch.zeros_like(self.root_values)
            self.re_probs = torch.zeros_like(self.probs)
        if self.game_base is not None:
            idx, n = self._positions(g), int(self.lengths[g])
            as_dev = lambda x: torch.as_tensor(np.asarray(x) if not torch.is_tensor(x) else x, dtype=torch.float64,  # noqa: E731
                                               device=self.device)
            self.re_probs[idx] = as_dev(action_probabilities).reshape(n, self.A)
            self.re_root_values[idx] = as_dev(root_values).reshape(n)
            self.has_reanalysed[g] = 1
            return
        o, n = int(self.offsets[g]), int(self.lengths[g])
        self.re_probs[o:o + n] = torch.as_tensor(np.asarray(action_probabilities) if not torch.is_tensor(action_probabilities)
                                                 else action_probabilities, dtype=torch.float64, device=self.device).reshape(n, self.A)
        self.re_root_values[o:o + n] = torch.as_tensor(np.asarray(root_values) if not torch.is_tensor(root_values)
                                                       else root_values, dtype=torch.float64, device=self.device).reshape(n)
        self.has_reanalysed[g] = 1

    def _struct(self) -> MzTrajectories:
        p = lambda t: None if t is None else t.data_ptr()  # noqa: E731
        return MzTrajectories(len(self), self.obs_size, self.A, 0, p(self.game_offset), p(self.observations),
                              p(self.actions), p(self.rewards), p(self.to_plays), p(self.root_values), p(self.probs),
                              p(self.re_root_values), p(self.re_probs),
                              p(self.has_reanalysed if self.re_probs is not None else None), p(self.game_base),
                              int(self.stride))

    def make_targets(self, game_idx, t, absorbed, td_steps: int, gamma: float, unroll_steps: int,
                     n_stack: int = 1, stack_action: bool = False):
        """Batch tensors for the sampled positions: (observations, actions, value, reward, policy), float32,
        shaped like ReplayBuffer.sample's (actions as flat one-hot rows of n_actions)."""
        gi = np.asarray(game_idx, dtype=np.int64).reshape(-1)
        ts = np.asarray(t, dtype=np.int64).reshape(-1)
        n, U1 = len(gi), int(unroll_steps) + 1
        if n == 0 or len(ts) != n:
            raise ValueError('make_targets: game_idx and t must be non-empty and of equal length')
        if (gi < 0).any() or (gi >= len(self)).any() or (ts < 0).any() or (ts >= self.lengths[gi]).any():
            raise ValueError('make_targets: sampled position outside the stored games')
        ab = np.asarray(absorbed, dtype=np.int64).reshape(n, U1)
        need = np.maximum(0, ts + U1 - self.lengths[gi])          # padded slots per sample
        for i in np.nonzero(need)[0]:
            pad = ab[i, :need[i]]
            if (pad < 0).any() or (pad >= self.lengths[gi[i]]).any():
                raise ValueError('make_targets: absorbing index outside its game')
        dev = lambda x: torch.as_tensor(np.ascontiguousarray(x), dtype=torch.int32, device=self.device)  # noqa: E731
        gi_d, ts_d, ab_d = dev(gi), dev(ts), dev(np.where(ab < 0, 0, ab))
        gp = torch.as_tensor(np.array([float(gamma) ** k for k in range(int(td_steps) + 1)]), device=self.device)
        planes = 2 if stack_action else 1
        f32 = lambda *s: torch.empty(*s, dtype=torch.float32, device=self.device)  # noqa: E731
        obs, act = f32(n, n_stack * planes * self.obs_size), f32(n, U1, self.A)
        val, rew, pol = f32(n, U1), f32(n, U1), f32(n, U1, self.A)
        st = self._struct()
        _check(lib().mz_make_targets(C.byref(st), C.c_int32(n), _ptr(gi_d), _ptr(ts_d), _ptr(ab_d),
                                     C.c_int32(int(td_steps)), C.c_int32(int(unroll_steps)), C.c_double(float(gamma)),
                                     _ptr(gp), C.c_int32(int(n_stack)), C.c_int32(int(bool(stack_action))), _ptr(obs),
                                     _ptr(act), _ptr(val), _ptr(rew), _ptr(pol), _stream()), 'mz_make_targets')
        lead = self.obs_shape[0] if len(self.obs_shape) > 1 else 1
        obs = obs.reshape((n, n_stack * planes * lead) + tuple(self.obs_shape[1:])) if len(self.obs_shape) > 1 else obs
        return obs, act, val, rew, pol

    def sample(self, batch_size: int, td_steps: int, gamma: float, unroll_steps: int, n_stack: int = 1,
               stack_action: bool = False):
        """ReplayBuffer.sample (replay_buffer.py:59-105): games with replacement, one uniform position per
        game, absorbing actions sampled from the same game; `np.random` is consumed in the same order."""
        sel = np.random.choice(range(len(self)), size=batch_size)
        U1 = unroll_steps + 1
        ts, absorbed = [], np.full((batch_size, U1), -1, dtype=np.int64)
        for i, g in enumerate(sel):
            T = int(self.lengths[g])
            t = np.random.randint(T)
            ts.append(t)
            short = U1 - len(range(t, min(T, t + U1)))
            if short > 0:
                absorbed[i, :short] = np.random.choice(range(T), size=short)
        return self.make_targets(sel, ts, absorbed, td_steps, gamma, unroll_steps, n_stack, stack_action)
