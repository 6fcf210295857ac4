"""Host view of one search tree: the field set, constants and accessors of the reference's
`Tree` (mcts/tree.py:7-68).  The engine keeps trees in HBM; a `Tree` is what
`SearchEngine.export_tree` / `MCTS.last_tree` hand back for inspection."""
from dataclasses import dataclass
from typing import ClassVar, Optional, Tuple

import numpy as np


@dataclass(frozen=True)
class Tree:
    node_visits: np.ndarray            # (N,) int32
    node_values: np.ndarray            # (N,) float64
    node_raw_values: np.ndarray        # (N,) float64
    parents: np.ndarray                # (N,) int32
    action_from_parent: np.ndarray     # (N,) int32
    children_index: np.ndarray         # (N, A) int32
    children_prior_logits: np.ndarray  # (N, A) float64 (PUCT) / float32 (Gumbel)
    children_visits: np.ndarray        # (N, A) int32
    children_rewards: np.ndarray       # (N, A) float64
    children_values: np.ndarray        # (N, A) float64
    hidden_states: np.ndarray          # (N, ...) float32
    to_play: np.ndarray                # (N,) int32
    root_legal_actions: np.ndarray     # (A,)
    discount: float
    gumbel_noise: Optional[np.ndarray]  # (A,) or None

    ROOT_INDEX: ClassVar[int] = 0
    NO_PARENT: ClassVar[int] = -1
    UNVISITED: ClassVar[int] = -1
    P1: ClassVar[int] = 0

    @property
    def n_actions(self) -> int:
        return self.children_index.shape[-1]

    @property
    def n_simulations(self) -> int:
        return self.node_visits.shape[-1] - 1

    def qvalues(self, node_index: int) -> np.ndarray:
        """Q(s, a) = r(s, a) + discount * V(child)  (mcts/tree.py:55-58)."""
        return self.children_rewards[node_index] + self.discount * self.children_values[node_index]

    def summary(self) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
        """Root visit counts, visit distribution (uniform when nothing was visited) and root
        Q-values (mcts/tree.py:60-68)."""
        counts = self.children_visits[Tree.ROOT_INDEX]
        total = np.sum(counts, keepdims=True)
        probs = np.where(total > 0, counts / np.maximum(total, 1), 1 / self.n_actions)
        return counts, probs, self.qvalues(Tree.ROOT_INDEX)
