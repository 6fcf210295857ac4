"""Value transforms with the reference's signatures (mcts/qtransforms.py), evaluated on the GPU."""
from __future__ import annotations

import numpy as np

from .tree import Tree
from . import _node_engine

_EPS = 1e-8


def _check_eps(epsilon):
    if epsilon != _EPS:
        raise NotImplementedError('the device kernels implement the reference default epsilon=1e-8')


def qtransform_by_parent_and_siblings(tree: Tree, node_index: int, epsilon: float = _EPS) -> np.ndarray:
    """Min-max normalised Q-values, unvisited actions completed by the minimum (mcts/qtransforms.py:7-22)."""
    _check_eps(epsilon)
    gumbel = tree.children_prior_logits.dtype == np.float32
    return _node_engine.load(tree, gumbel).node_qtransform(0, node_index, 0)


def qtransform_completed_by_mix_value(tree: Tree, node_index: int, c_visit: int, c_scale: float,
                                      epsilon: float = _EPS) -> np.ndarray:
    """Completed Q-values scaled by (c_visit + max visit) * c_scale (mcts/qtransforms.py:25-48)."""
    _check_eps(epsilon)
    return _node_engine.load(tree, True, c_visit=c_visit, c_scale=c_scale).node_qtransform(0, node_index, 1)


def compute_mixed_value(raw_value, qvalues: np.ndarray, visit_counts: np.ndarray,
                        prior_logits: np.ndarray) -> np.ndarray:
    """The per-action mixed-value vector of mcts/qtransforms.py:51-77 for explicit inputs."""
    A = len(qvalues)
    z = lambda dt: np.zeros((2, A), dt)  # noqa: E731
    t = Tree(node_visits=np.array([1, 0], np.int32), node_values=np.zeros(2), node_raw_values=np.array([float(np.asarray(raw_value).reshape(-1)[0]), 0.0]),
             parents=np.full(2, -1, np.int32), action_from_parent=np.full(2, -1, np.int32),
             children_index=np.full((2, A), -1, np.int32), children_prior_logits=z(np.float32),
             children_visits=z(np.int32), children_rewards=z(np.float64), children_values=z(np.float64),
             hidden_states=None, to_play=np.zeros(2, np.int32), root_legal_actions=np.ones(A), discount=0.0,
             gumbel_noise=np.zeros(A))
    t.children_prior_logits[0] = np.asarray(prior_logits, np.float32)
    t.children_visits[0] = visit_counts
    t.children_rewards[0] = qvalues  # with discount 0 the node's Q-values are exactly these
    return _node_engine.load(t, True).node_qtransform(0, 0, 2)
