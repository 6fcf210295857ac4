"""Runs single-node rules of a host `Tree` through the engine (import tree -> node kernel).

The module-level functions of `action_selection` / `qtransforms` keep the reference's per-node
signatures; each call uploads the tree it is given (a few KB) into a cached one-tree engine
context and evaluates the rule with the same device code the batched search uses.
"""
from __future__ import annotations

from types import SimpleNamespace

import numpy as np

from .mcts import _engine

_cache = {}


def _ctx(tree, gumbel: bool, discount, **params):
    A, S = tree.n_actions, tree.n_simulations
    H = int(np.prod(tree.hidden_states.shape[1:])) if tree.hidden_states is not None else 0
    key = (A, S, H, gumbel, float(discount), tuple(sorted(params.items())))
    eng = _cache.get(key)
    if eng is None:
        cfg = SimpleNamespace(n_actions=A, simulations=S, players=1, gumbel=gumbel, gamma=discount,
                              c_base=params.get('c_base', 19625), c_init=params.get('c_init', 1.25),
                              max_considered_actions=params.get('max_considered_actions', A),
                              c_visit=params.get('c_visit', 50), c_scale=params.get('c_scale', 1.0))
        eng = _engine.SearchEngine(cfg, H, 1, noise_mode=_engine.NOISE_NONE)
        _cache[key] = eng
    return eng


def load(tree, gumbel: bool, **params):
    """Upload `tree` (host arrays) as tree 0 of a matching one-tree context."""
    eng = _ctx(tree, gumbel, tree.discount, **params)
    want = np.float32 if gumbel else np.float64
    if tree.children_prior_logits.dtype != want:
        raise TypeError(f'this rule expects {want.__name__} prior logits (the dtype the '
                        f'{"Gumbel" if gumbel else "PUCT"} search stores), got {tree.children_prior_logits.dtype}')
    arrays = {k: getattr(tree, k) for k in ('node_visits', 'node_values', 'node_raw_values', 'parents',
                                           'action_from_parent', 'to_play', 'children_index',
                                           'children_prior_logits', 'children_visits', 'children_rewards',
                                           'children_values')}
    arrays['hidden_states'] = None
    eng.import_tree(0, arrays, tree.root_legal_actions, tree.gumbel_noise if gumbel else None)
    return eng
