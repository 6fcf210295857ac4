"""Batched Reanalyse: every position of the sampled games is searched again in ONE batch.

Mirrors the body of the reference's `Reanalyser.reanalyse` loop (reanalyse.py:44-79), which walks
the positions of one game serially: stacked observations for all t, target-network values when
`config.mcts_target_value` is false, a fresh search per position from a NEW game's legal mask
(the actor's own game object is never stepped, reanalyse.py:69) with the recorded `to_plays[t]`,
then `save_reanalysed_stats`.  Positions are independent, so they form one search batch.
"""
from __future__ import annotations

from typing import List, Sequence

import numpy as np
import torch

from . import noise as _noise
from .mcts.mcts import MCTS, network_device
from .targets import TrajectoryStore


class BatchedReanalyser:
    def __init__(self, game, config, target_network, seed: int = 0) -> None:
        self.game = game            # only its fresh legal_actions() / action_encoder are used
        self.config = config
        self.target_network = target_network
        self.mcts = MCTS(config)
        self.seed = seed
        self.calls = 0             # batches searched so far: each gets tie-break uniforms of its own

    def set_weights(self, network) -> None:
        """Target-network refresh (reanalyse.py:38-42); the cadence is the caller's."""
        self.target_network = network

    def reanalyse_games(self, histories: Sequence, tree_id0: int = 0) -> List:
        """Recomputes policies and values of every position of `histories` in place
        (`save_reanalysed_stats`) and returns them; positions are batched in game order, and position
        k of the batch uses tie-break noise key `tree_id0 + k`."""
        cfg = self.config
        lengths = np.array([len(h) for h in histories], dtype=np.int64)
        B = int(lengths.sum())
        if B == 0:
            return []
        to_play = np.concatenate([np.asarray(h.to_plays[:n], dtype=np.int64) for h, n in zip(histories, lengths) if n])
        # stacked observations of every position (games/game.py:120-160) by the kernel that builds training batches:
        # one launch over all positions instead of a Python call per position
        played = [h for h, n in zip(histories, lengths) if n]
        store = TrajectoryStore(played, cfg.n_actions)
        gi = np.repeat(np.arange(len(played)), lengths[lengths > 0])
        ts = np.concatenate([np.arange(n) for n in lengths if n])
        obs = store.make_targets(gi, ts, np.full((B, 1), -1), 0, float(getattr(cfg, 'gamma', 1.0)), 0,
                                 int(cfg.n_stacked_observations), bool(cfg.stack_action))[0]
        legal = np.repeat(np.asarray(self.game.legal_actions(), dtype=np.float64)[None], B, axis=0)
        values = None
        if not getattr(cfg, 'mcts_target_value', True):
            with torch.no_grad():
                values = self.target_network.initial_inference(obs.to(network_device(self.target_network)))[2]
            values = np.asarray(values.detach().cpu().numpy(), dtype=np.float64).reshape(B)
        _, root_values, probs = self.mcts.search_batch(self.target_network, obs, legal, np.asarray(to_play), 0,
                                                       action_encoder=self.game.action_encoder,
                                                       seed=_noise.move_seed(self.seed, self.calls),
                                                       tree_id0=tree_id0)
        self.calls += 1
        out, k = [], 0
        for h in histories:
            n = len(h)
            p = [probs[k + i] for i in range(n)]
            v = [float(x) for x in (root_values if values is None else values)[k:k + n]]
            h.save_reanalysed_stats(p, v)
            out.append((p, v))
            k += n
        return out
