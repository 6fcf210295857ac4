"""CartPole with the reference's interface (games/cartpole.py:9-48).  The reference delegates the
physics to `gymnasium.make('CartPole-v1')` (version unpinned, not installed here); this class restates
the public CartPole-v1 specification directly (Euler, tau 0.02, float64 state, float32 observation)."""
from __future__ import annotations

import math

import numpy as np

from .game import Game


class CartPole(Game):
    GRAVITY, MASSCART, MASSPOLE, LENGTH, FORCE, TAU = 9.8, 1.0, 0.1, 0.5, 10.0, 0.02
    X_LIMIT, THETA_LIMIT = 2.4, 12 * 2 * math.pi / 360

    def __init__(self, render: bool = False, seed=None) -> None:
        super().__init__(1, [1, 1, 4], 2)
        self.rng = np.random.default_rng(seed)
        self.state = None

    def reset(self):
        self.state = self.rng.uniform(-0.05, 0.05, size=4)
        return np.array([[self.state.astype(np.float32)]])

    def legal_actions(self) -> np.ndarray:
        return np.ones(2)

    def step(self, action: int):
        x, x_dot, th, th_dot = self.state
        force = self.FORCE if action == 1 else -self.FORCE
        total, pml = self.MASSPOLE + self.MASSCART, self.MASSPOLE * self.LENGTH
        c, s = math.cos(th), math.sin(th)
        temp = (force + pml * th_dot ** 2 * s) / total
        th_acc = (self.GRAVITY * s - c * temp) / (self.LENGTH * (4.0 / 3.0 - self.MASSPOLE * c ** 2 / total))
        x_acc = temp - pml * th_acc * c / total
        x, x_dot = x + self.TAU * x_dot, x_dot + self.TAU * x_acc
        th, th_dot = th + self.TAU * th_dot, th_dot + self.TAU * th_acc
        self.state = np.array([x, x_dot, th, th_dot])
        done = bool(x < -self.X_LIMIT or x > self.X_LIMIT or th < -self.THETA_LIMIT or th > self.THETA_LIMIT)
        return np.array([[self.state.astype(np.float32)]]), 1.0, done

    def action_encoder(self, action: int):
        one_hot = np.zeros(self.n_actions)
        one_hot[action] = 1
        return one_hot

    def visit_softmax_temperature_func(self, training_steps: int, training_step: int) -> float:
        if training_step < 0.5 * training_steps:
            return 1.0
        return 0.5 if training_step < 0.75 * training_steps else 0.25
