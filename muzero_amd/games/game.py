"""`Game` interface and the `GameHistory` trajectory (reference games/game.py:12-253)."""
from __future__ import annotations

from abc import ABC, abstractmethod
from typing import Callable, List, Tuple

import numpy as np


class Game(ABC):
    """What self-play needs from an environment (games/game.py:12-53)."""

    def __init__(self, players: int, observation_dim: List[int], n_actions: int) -> None:
        self.players, self.observation_dim, self.n_actions = players, observation_dim, n_actions

    @abstractmethod
    def reset(self): ...

    @abstractmethod
    def legal_actions(self) -> np.ndarray: ...

    @abstractmethod
    def step(self, action: int): ...

    @abstractmethod
    def action_encoder(self, action: int): ...

    @abstractmethod
    def visit_softmax_temperature_func(self, training_steps: int, training_step: int) -> float: ...

    def render(self) -> None:
        pass

    def to_play(self) -> int:
        return 0


class BoardGame(Game):
    """Two-player square board, X (-1) moves first (games/game.py:56-74)."""

    def __init__(self, size: int, observation_dim: List[int]) -> None:
        super().__init__(2, observation_dim, size ** 2)
        self.size = size
        self.reset()

    def reset(self):
        self.board = np.zeros((self.size, self.size))
        self._to_play = -1
        self.winner = None
        return self.observation()

    def render(self) -> None:
        sym = {-1: 'X', 1: 'O', 0: ' '}
        for row in self.board:
            print(' | '.join(sym[int(c)] for c in row))


class GameHistory:
    """One game's trajectory in the reference's list layout (games/game.py:77-253): per move the
    observation BEFORE the move, the action (+ its encoding), reward, player to move, search policy
    and root value."""

    def __init__(self, action_encoder: Callable, initial_observation) -> None:
        self.observations, self.actions, self.encoded_actions = [], [], []
        self.rewards, self.to_plays = [], []
        self.action_probabilities, self.root_values = [], []
        self.reanalysed_action_probabilities = None
        self.reanalysed_root_values = None
        self.action_encoder = action_encoder
        self.initial_observation = initial_observation

    def __len__(self) -> int:
        return len(self.observations)

    def save(self, observation, action: int, reward: float, to_play: int, pi, root_value) -> None:
        self.observations.append(observation)
        self.actions.append(action)
        self.encoded_actions.append(self.action_encoder(action))
        self.rewards.append(reward)
        self.to_plays.append(to_play)
        self.action_probabilities.append(pi)
        self.root_values.append(root_value)

    def save_reanalysed_stats(self, action_probabilities, root_values) -> None:
        self.reanalysed_action_probabilities = action_probabilities
        self.reanalysed_root_values = root_values

    def stack_n_observations(self, t: int, n: int, n_actions: int, stack_action: bool) -> np.ndarray:
        """o_t, o_{t-1}, ... (newest first), each optionally followed by a plane filled with
        action/n_actions, zero planes where the game is younger than n (games/game.py:120-160)."""
        def pair(obs, act_plane):
            return [obs, act_plane] if stack_action else [obs]
        planes = []
        if len(self) == 0:
            zero = np.zeros_like(self.initial_observation)
            planes += pair(self.initial_observation, zero)
            for _ in range(n - 1):
                planes += pair(zero, zero)
        else:
            t = t % len(self)
            have = min(n, t + 1)
            for step in range(t, t - have, -1):
                planes += pair(self.observations[step],
                               np.full_like(self.observations[step], self.actions[step] / n_actions))
            zero = np.zeros_like(self.observations[t - have + 1])
            for _ in range(n - have):
                planes += pair(zero, zero)
        return np.concatenate(planes, axis=0)

    def compute_return(self, gamma: float, player: int, players: int) -> float:
        """Undiscounted episode return from `player`'s point of view (games/game.py:162-183; the sign
        rule indexes the last reward by len-1 and earlier ones by len-i, as the reference does)."""
        def signed(reward, step):
            flip = players == 2 and ((step % 2 == 0 and player == 1) or (step % 2 == 1 and player == 0))
            return -reward if flip else reward
        T = len(self.observations)
        total = signed(self.rewards[-1], T - 1)
        for i, r in enumerate(reversed(self.rewards[0:-1])):
            total = total + signed(r, T - i)
        return total

    def make_target(self, t: int, td_steps: int, gamma: float, unroll_steps: int, n_actions: int
                    ) -> Tuple[list, list, list]:
        """Value / reward / policy targets for positions t..t+unroll_steps (games/game.py:185-253):
        signed sum of remaining rewards when gamma == 1 (board games), else the n-step bootstrapped
        return; absorbing targets (0, 0, uniform) past the end."""
        def value_at(step):
            if gamma == 1:
                return sum(self.rewards[i] if self.to_plays[i] == self.to_plays[step] else -self.rewards[i]
                           for i in range(step, len(self)))
            boot = step + td_steps
            values = self.reanalysed_root_values if self.reanalysed_root_values else self.root_values
            if boot < len(self):
                b = values[boot] if self.to_plays[boot] == self.to_plays[step] else -values[boot]
                b *= gamma ** td_steps
            else:
                b = 0
            disc = [(r if self.to_plays[step + k] == self.to_plays[step] else -r) * gamma ** k
                    for k, r in enumerate(self.rewards[step + 1: boot + 1])]
            return sum(disc) + b
        vt, rt, pt = [], [], []
        for step in range(t, t + unroll_steps + 1):
            v = value_at(step)
            if step < len(self):
                vt.append(v)
                rt.append(self.rewards[step])
                pt.append(self.reanalysed_action_probabilities[step] if self.reanalysed_action_probabilities
                          else self.action_probabilities[step])
            else:
                vt.append(0)
                rt.append(0)
                pt.append([1 / n_actions] * n_actions)
        return vt, rt, pt
