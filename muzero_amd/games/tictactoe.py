"""TicTacToe with the reference's interface and scoring quirks (games/tictactoe.py:9-81)."""
from __future__ import annotations

import numpy as np

from .game import BoardGame


class TicTacToe(BoardGame):

    def __init__(self, size: int = 3) -> None:
        super().__init__(size, [3, size, size])

    def terminated(self) -> bool:
        """First complete line in the order columns, rows, diagonal, anti-diagonal names the winner;
        a full board then overrides it with a tie (so a board-filling win scores as a tie)."""
        b, n = self.board, self.size
        sums = list(b.sum(axis=0)) + list(b.sum(axis=1)) + [np.trace(b), np.trace(np.fliplr(b))]
        done = False
        for s in sums:
            if abs(s) == n:
                done, self.winner = True, int(s / n)
                break
        if np.abs(b).sum() == b.size:
            done, self.winner = True, 0
        return done

    def legal_actions(self) -> np.ndarray:
        return (self.board.reshape(-1) == 0).astype(np.float64)

    def step(self, action: int):
        self.board[action // self.size, action % self.size] = self._to_play  # no occupancy check
        self._to_play *= -1
        done = self.terminated()
        reward = 1 if (self.winner is not None and self.winner != 0) else 0
        return self.observation(), reward, done

    def observation(self):
        return np.array([np.where(self.board == -1, 1, 0), np.where(self.board == 1, 1, 0),
                         np.full_like(self.board, self._to_play)])

    def action_encoder(self, action: int):
        plane = np.zeros((self.size, self.size))
        plane[action // self.size, action % self.size] = 1
        return plane

    def visit_softmax_temperature_func(self, training_steps: int, training_step: int) -> float:
        return 1.0

    def to_play(self) -> int:
        return 0 if self._to_play == -1 else 1
