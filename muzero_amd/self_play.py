"""Batched self-play: B environments stepped on the device, one search per move per environment.

The loop of the reference's `SelfPlay.play` (self_play.py:77-115) for a whole batch of games at once:
search from each game's observation, act, step the game, record the move, restart finished games.
Per move this is three launches on the fused path (root inference + root rows, all simulations with
in-kernel leaf evaluation, action rule + environment step); trees and environments never leave HBM.
Every move's record (the fields of `GameHistory.save`) is appended by the action + environment kernel to a
device move log; finished games are read from it in place by `TrajectoryStore.from_log`, or copied out once
as `GameHistory` objects for a host replay buffer.
"""
from __future__ import annotations

from typing import Callable, List, Optional

import numpy as np
import torch

from . import engine as _engine
from . import noise as _noise
from .games.game import GameHistory
from .targets import MoveLog, TrajectoryStore


class DeviceEnvs:
    """Caller-owned device state of B environments (mz_env_state of the C ABI)."""

    def __init__(self, kind: str, B: int, device, max_moves: int, stale_observation: bool = True, seed: int = 0,
                 n_stacked_observations: int = 1, stack_action: bool = False):
        """n_stacked_observations / stack_action (config fields of the same names): the search input becomes
        GameHistory.stack_n_observations(-1, ...) of every game so far (games/game.py:120-160), kept on the device."""
        self.kind, self.B, self.device = kind, int(B), device
        self.n_stack, self.planes = max(1, int(n_stacked_observations)), 2 if stack_action else 1
        if (self.n_stack > 1 or stack_action) and not stale_observation:
            raise ValueError('stacked observations follow the saved history (the stale rule): stale_observation must be on')
        self.rng = np.random.default_rng(seed)
        if kind == 'tictactoe':
            self.A, self.obs_size, self.code = 9, 27, _engine.ENV_TICTACTOE
            self.state = torch.zeros((B, 9), dtype=torch.int8, device=device)
            self.mover = torch.full((B,), -1, dtype=torch.int32, device=device)
            obs = torch.zeros((B, 27), dtype=torch.float32, device=device)
            obs[:, 18:] = -1.0   # empty board, X (-1) to move
            self.legal = torch.ones((B, 9), dtype=torch.uint8, device=device)
            self.reset_state = None
        elif kind == 'cartpole':
            self.A, self.obs_size, self.code = 2, 4, _engine.ENV_CARTPOLE
            s0 = self.rng.uniform(-0.05, 0.05, size=(B, 4))
            self.state = torch.as_tensor(s0, dtype=torch.float64, device=device)
            self.mover = None
            obs = self.state.to(torch.float32)
            self.legal = torch.ones((B, 2), dtype=torch.uint8, device=device)
            self.reset_state = torch.as_tensor(self.rng.uniform(-0.05, 0.05, size=(B, 4)), device=device)
        else:
            raise ValueError(f'unknown environment {kind!r}')
        self.cur_obs = obs.contiguous()
        # a game that has not started: the first observation followed by zero planes (games/game.py:134-141)
        self.search_obs = torch.zeros((B, self.n_stack * self.planes * self.obs_size), dtype=torch.float32, device=device)
        self.search_obs[:, :self.obs_size] = obs
        self.to_play = torch.zeros(B, dtype=torch.int32, device=device)
        self.moves = torch.zeros(B, dtype=torch.int32, device=device)
        p = lambda t: None if t is None else t.data_ptr()  # noqa: E731
        self.struct = _engine.MzEnvState(self.code, int(max_moves), int(bool(stale_observation)), self.obs_size,
                                         p(self.state), p(self.mover), p(self.to_play), p(self.legal),
                                         p(self.cur_obs), p(self.search_obs), p(self.moves), p(self.reset_state),
                                         self.n_stack, self.planes - 1)

    NOISE_BLOCK = 16   # moves whose random inputs are drawn by one set of launches

    def refresh_reset_states(self):
        """New initial states for environments that restart on the next step (CartPole), drawn on the device:
        NOISE_BLOCK moves' worth per draw, one slice handed to the step kernel per move."""
        if self.reset_state is not None:
            if getattr(self, '_gen', None) is None:
                self._gen = torch.Generator(device=self.device)
                self._gen.manual_seed(int(self.rng.integers(0, 2**31)))
                self._reset_block, self._reset_k = None, self.NOISE_BLOCK
            if self._reset_k >= self.NOISE_BLOCK:
                self._reset_block = torch.empty((self.NOISE_BLOCK,) + tuple(self.reset_state.shape), dtype=self.reset_state.dtype,
                                                device=self.device).uniform_(-0.05, 0.05, generator=self._gen)
                self._reset_k = 0
            self.reset_state = self._reset_block[self._reset_k]
            self.struct.reset_state = self.reset_state.data_ptr()
            self._reset_k += 1


class BatchedSelfPlay:
    """config: the search config plus `max_moves`; network: a muzero_amd.network.MuZeroDenseNetwork
    (fused in-kernel leaf evaluation when it fits in LDS, matrix-core evaluation otherwise) or any
    object with the reference's network interface (evaluated between the kernels)."""

    def __init__(self, config, game: str, network, B: int, seed: int = 0, stale_observation: bool = True,
                 tree_id0: int = 0, device=None, noise_mode=_engine.NOISE_COUNTER):
        self.config, self.network, self.B = config, network, int(B)
        self.device = torch.device('cuda', torch.cuda.current_device()) if device is None else device
        self.envs = DeviceEnvs(game, B, self.device, config.max_moves, stale_observation, seed,
                               int(getattr(config, 'n_stacked_observations', 1) or 1), bool(getattr(config, 'stack_action', False)))
        stack = self.envs.n_stack * self.envs.planes * self.envs.obs_size
        if getattr(network, 'obs_size', stack) != stack:
            raise ValueError(f'the network reads {network.obs_size} floats per observation, the environments stack '
                             f'{self.envs.n_stack} x {self.envs.planes} planes of {self.envs.obs_size}')
        A = self.envs.A
        self.gumbel = bool(config.gumbel)
        self.tree_id0 = tree_id0
        self.rng = np.random.default_rng(seed + 1)
        self.mode = 'generic'
        self._noise_mode, self._seed = noise_mode, seed
        H = None
        if hasattr(network, 'recurrent_flat'):
            from .packing import pack_network, pack_network_mfma, pack_root_network_mfma
            H = network.H
            # (a representation evaluated by torch - downsampled observation stacks - has no program to pack)
            self._torch_root = getattr(network.rep, 'steps', None) is None
            packed = None if self._torch_root else pack_network(network)
            self.engine = _engine.SearchEngine(config, H, B, seed=seed, noise_mode=noise_mode, device=self.device)
            lanes_groups = 256 // 16
            fits = (packed is not None and
                    packed.recurrent.weight_bytes() + lanes_groups * packed.recurrent.scratch * 4 < 150 * 1024 and
                    packed.represent.weight_bytes() + packed.predict.weight_bytes() < 100 * 1024)
            if fits:
                self.engine.set_network(packed)
                self.mode = 'fused'
            else:
                network.to(self.device)
                self.engine.set_network_mfma(*pack_network_mfma(network))
                if not self._torch_root:
                    self.engine.set_root_network_mfma(*pack_root_network_mfma(network))   # the roots, too, on the device
                self.mode = 'mfma'
        else:
            self.engine = None  # created on the first step, once the hidden size is known
        self.out = dict(action=torch.empty(B, dtype=torch.int32, device=self.device),
                        probs=torch.empty((B, A), dtype=torch.float64, device=self.device),
                        value=torch.empty(B, dtype=torch.float64, device=self.device),
                        obs_before=torch.empty((B, self.envs.obs_size), dtype=torch.float32, device=self.device),
                        reward=torch.empty(B, dtype=torch.float32, device=self.device),
                        done=torch.empty(B, dtype=torch.uint8, device=self.device),
                        to_play=torch.empty(B, dtype=torch.int32, device=self.device))
        self.steps_done = 0

    # ------------------------------------------------------------------ one move for every environment
    def _root_noise(self):
        """Fresh Gumbel / Dirichlet root noise (and the final sampling uniforms) for every environment, drawn on
        the device for DeviceEnvs.NOISE_BLOCK moves at a time: the half dozen small generator launches then cost
        a sixteenth per move."""
        A, B, K = self.envs.A, self.B, DeviceEnvs.NOISE_BLOCK
        if getattr(self, '_gen', None) is None:
            self._gen = torch.Generator(device=self.device)
            self._gen.manual_seed(int(self.rng.integers(0, 2**31)))
            self._noise_k = K
        if self._noise_k >= K:
            self._u_block = torch.rand((K, B), dtype=torch.float64, device=self.device, generator=self._gen)
            if self.gumbel:
                uu = torch.rand((K, B, A), dtype=torch.float64, device=self.device, generator=self._gen).clamp_min_(1e-300)
                self._noise_block = -torch.log(-torch.log(uu))
            else:
                alpha = torch.full((K, B, A), float(self.config.dirichlet_alpha), dtype=torch.float64, device=self.device)
                gam = torch._standard_gamma(alpha, generator=self._gen).clamp_min_(1e-300)
                self._noise_block = gam / gam.sum(dim=2, keepdim=True)
            self._noise_k = 0
        k = self._noise_k
        self._noise_k += 1
        self._u_next = self._u_block[k]
        return (None, self._noise_block[k]) if self.gumbel else (self._noise_block[k], None)

    def step(self, temperature: float = 1.0, root_noise=None, u=None, noise=None):
        """Searches and plays one move in every environment.  root_noise: (dirichlet, gumbel) tensors
        to use instead of fresh draws; u: the sampling uniforms (PUCT).  Returns the per-step record
        dict (device tensors, overwritten by the next step)."""
        e, cfg, eng = self.envs, self.config, self.engine
        dirichlet, g = root_noise if root_noise is not None else self._root_noise()
        frac = 0.0 if self.gumbel else cfg.exploration_frac
        if eng is not None and not self.gumbel and self._noise_mode == _engine.NOISE_COUNTER:
            eng.set_seed(_noise.move_seed(self._seed, self.steps_done))   # fresh tie-break uniforms every move
        if self.mode == 'fused':
            eng.begin_net(e.search_obs, e.legal, e.to_play, dirichlet, frac, g, tree_id0=self.tree_id0)
            eng.search_net(noise)
        else:
            net = self.network
            with torch.no_grad():
                if self.mode == 'mfma':
                    lg, hid, val = net.initial_flat(e.search_obs) if self._torch_root else eng.initial_inference_mfma(e.search_obs)
                else:
                    shape = getattr(net, 'observation_shape', None)
                    obs = e.search_obs.cpu().reshape(self.B, *shape) if shape else e.search_obs.cpu()
                    lg, hid, val = net.initial_inference(obs)
                if eng is None:
                    H = int(np.prod(hid.shape[1:]))
                    eng = self.engine = _engine.SearchEngine(cfg, H, self.B, seed=self._seed, device=self.device,
                                                             noise_mode=self._noise_mode)
                prior = eng.root_prepare(lg.reshape(self.B, e.A), e.legal, dirichlet, frac)
                eng.reset(prior, val.reshape(self.B), hid.reshape(self.B, -1), e.to_play, e.legal, g,
                          tree_id0=self.tree_id0)
                if self.mode == 'mfma':
                    eng.search_mfma(noise)   # the whole simulation loop in one call
                for _ in range(int(cfg.simulations) if self.mode != 'mfma' else 0):
                    parent, action, phid = eng.select(noise)
                    enc = torch.nn.functional.one_hot(action.cpu().long(), net.action_size if hasattr(net, 'action_size') else e.A).float()
                    l2, nh, v2, r2 = net.recurrent_inference(phid.cpu().reshape(self.B, *hid.shape[1:]), enc)
                    eng.expand_backup(l2.reshape(self.B, e.A), v2.reshape(self.B), r2.reshape(self.B),
                                      nh.reshape(self.B, -1))
        if u is None and not self.gumbel:
            if root_noise is not None or getattr(self, '_u_next', None) is None:
                self._root_noise()   # advances the block that also holds the sampling uniforms
            u, self._u_next = self._u_next, None
        e.refresh_reset_states()
        log = self.log
        if log is not None and log.rows >= log.capacity:
            raise _engine.EngineError(f'the move log is full ({log.capacity} moves per environment): read it with '
                                      'harvest() or start a new window with record(capacity)')
        eng.act_env_step(e.struct, temperature, u, self.out, None if log is None else log.struct,
                         0 if log is None else log.rows)
        if log is not None:
            log.rows += 1
        self.steps_done += 1
        return self.out

    # ------------------------------------------------------------------ trajectories
    log = None

    def record(self, capacity: int) -> MoveLog:
        """Starts a collection window: the next `capacity` moves of every environment go to a device move log
        (re-used when its size allows).  Only whole games are listed afterwards, as SelfPlay.play stores them
        (self_play.py:77-115): an environment in the middle of a game when the window opens is flagged, and its moves
        up to the end of that game are skipped by the game list, like the moves after the last finished game."""
        e = self.envs
        if self.log is None or self.log.capacity < capacity:
            shape = (3, 3, 3) if e.kind == 'tictactoe' else (1, 1, 4)
            self.log = MoveLog(capacity, self.B, shape, e.A, self.device)
        self.log.rows = 0
        self.log.done.zero_()
        torch.ne(e.moves, 0, out=self.log.open_at_start.view(torch.bool))
        return self.log

    def stop_recording(self) -> None:
        self.log = None

    def harvest(self) -> TrajectoryStore:
        """The games that finished inside the current window, as a device store over the log itself."""
        if self.log is None:
            raise _engine.EngineError('harvest: no collection window (call record(capacity) first)')
        return TrajectoryStore.from_log(self.log)

    def play(self, n_steps: int, temperature: float = 1.0, record: bool = True) -> List[GameHistory]:
        """Plays n_steps moves in every environment; with `record`, returns the games that finished inside
        them as GameHistory objects (host): the log is copied out once and sliced, no per-move transfers."""
        if record:
            self.record(n_steps)
        else:
            self.stop_recording()
        for _ in range(n_steps):
            self.step(temperature)
        if not record:
            return []
        try:
            store = self.harvest()
        except ValueError:
            return []
        A = self.envs.A
        encoder = (lambda a: np.eye(9)[a].reshape(3, 3)) if self.envs.kind == 'tictactoe' else (lambda a: np.eye(A)[a])
        return store.histories(encoder)
