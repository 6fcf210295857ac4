"""Action-selection rules with the reference's signatures (mcts/action_selection.py), evaluated
on the GPU by the same device code as the batched search."""
from __future__ import annotations

from functools import partial
from typing import Callable, Tuple

import numpy as np

from .tree import Tree
from . import _node_engine, qtransforms
from .mcts import _engine


def switching_action_selection_wrapper(root_action_selection_func: Callable,
                                       interior_action_selection_func: Callable) -> Callable:
    """depth 0 -> root rule, otherwise interior rule (mcts/action_selection.py:11-22)."""
    def switching_action_selection(tree: Tree, node_index: int, depth: int) -> int:
        fn = root_action_selection_func if depth == 0 else interior_action_selection_func
        return fn(tree, node_index, depth)
    return switching_action_selection


def _mix_params(qtransform) -> dict:
    """c_visit / c_scale bound into a `partial(qtransform_completed_by_mix_value, ...)`."""
    if isinstance(qtransform, partial) and qtransform.func is qtransforms.qtransform_completed_by_mix_value:
        return dict(c_visit=qtransform.keywords['c_visit'], c_scale=qtransform.keywords['c_scale'])
    raise NotImplementedError('device selection supports qtransform_completed_by_mix_value (as a partial '
                              'with c_visit and c_scale) only')


def muzero_action_selection(tree: Tree, node_index: int, depth: int, discount: float, c_base: float,
                            c_init: float, qtransform: Callable) -> int:
    """PUCT selection (mcts/action_selection.py:25-49); draws the tie-break uniforms from np.random."""
    if qtransform is not qtransforms.qtransform_by_parent_and_siblings:
        raise NotImplementedError('device selection supports qtransform_by_parent_and_siblings only')
    eng = _node_engine.load(tree, False, c_base=c_base, c_init=c_init)
    u = np.random.uniform(size=tree.n_actions)
    return eng.node_select(0, node_index, depth, u)


def muzero_action_selection_to_act(tree: Tree, temperature: float) -> Tuple[int, np.ndarray, np.ndarray]:
    """Visit-count policy with temperature (mcts/action_selection.py:52-70)."""
    eng = _node_engine.load(tree, False)
    u = np.random.random_sample()
    action, _, probs, q = eng.to_act(float(temperature), np.array([u]))
    return int(action[0]), probs[0].cpu().numpy(), q[0].cpu().numpy()


def gumbel_muzero_root_action_selection(tree: Tree, node_index: int, depth: int, discount: float,
                                        simulations: int, max_considered_actions: int, gumbel_scale: float,
                                        qtransform: Callable) -> int:
    """Sequential-halving root rule (mcts/action_selection.py:73-96); `gumbel_scale` is accepted and
    unused, as in the reference."""
    del depth, gumbel_scale
    if simulations != tree.n_simulations:
        raise ValueError('simulations must equal the tree capacity (tree.n_simulations)')
    eng = _node_engine.load(tree, True, max_considered_actions=max_considered_actions, **_mix_params(qtransform))
    return eng.node_select(0, node_index, 0)


def gumbel_muzero_interior_action_selection(tree: Tree, node_index: int, depth: int, discount: float,
                                            qtransform: Callable) -> int:
    """argmax(pi' - N/(1+sum N)) (mcts/action_selection.py:99-118)."""
    del depth
    eng = _node_engine.load(tree, True, **_mix_params(qtransform))
    return eng.node_select(0, node_index, 1)


def gumbel_muzero_action_selection_to_act(tree: Tree, qtransform: Callable
                                          ) -> Tuple[int, np.ndarray, np.ndarray]:
    """Final Gumbel action + improved policy (mcts/action_selection.py:121-138)."""
    eng = _node_engine.load(tree, True, **_mix_params(qtransform))
    action, _, probs, q = eng.to_act(0.0, None)
    return int(action[0]), probs[0].cpu().numpy(), q[0].cpu().numpy()


def masked_argmax(to_argmax: np.ndarray, legal_actions: np.ndarray) -> int:
    """mcts/action_selection.py:141-144."""
    if legal_actions is not None:
        to_argmax = np.where(legal_actions == 0, -np.inf, to_argmax)
    return np.argmax(to_argmax, axis=-1)


def gumbel_score(considered_visit: int, gumbel: np.ndarray, logits: np.ndarray,
                 normalized_qvalues: np.ndarray, visit_counts: np.ndarray) -> np.ndarray:
    """g + logits + q with -inf for actions off the considered visit count (mcts/action_selection.py:147-157)."""
    logits = logits - np.max(logits, keepdims=True)
    penalty = np.where(visit_counts == considered_visit, 0.0, -np.inf)
    return np.maximum(-1e9, gumbel + logits + normalized_qvalues) + penalty


def get_considered_visits(max_considered_actions: int, simulations: int):
    """Sequential-halving schedule (mcts/action_selection.py:160-175), from the library's host entry point."""
    return _engine.considered_visits(int(max_considered_actions), int(simulations))
