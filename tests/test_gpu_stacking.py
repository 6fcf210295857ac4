"""GPU: stacked observations kept on the device by the action + environment kernel.  Before every move the search
input of every environment must be GameHistory.stack_n_observations(-1, n, A, stack_action) of that game so far
(games/game.py:120-160 as called at self_play.py:86-91) - the checker's restatement of it, which
tests/test_targets_oracle.py pins on the reference's own outputs (tests/golden/stacking.npz), is applied to the
moves the environments actually played."""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from helpers import make_config

pytestmark = pytest.mark.gpu


def _selfplay(game, B, n, stack_action, seed):
    from muzero_amd.self_play import BatchedSelfPlay
    from muzero_amd.network import MuZeroNetwork
    planes = 2 if stack_action else 1
    if game == 'cartpole':
        OS, A, cfg = 4, 2, make_config(False, 2, 12, 1, 0.997)
        cfg.max_moves = 14
    else:
        OS, A, cfg = 27, 9, make_config(True, 9, 12, 2, 1)
        cfg.max_moves = 9
    cfg.n_stacked_observations, cfg.stack_action = n, stack_action
    # the network reads the whole stack as one flat observation (the reference's networks have no input for action
    # planes, network.py:299-308: a network that does is given the doubled plane count here)
    ncfg = SimpleNamespace(network='mlp', observation_dim=[1, 1, OS], n_actions=A, n_stacked_observations=n * planes,
                           encoding_size=8, fc_reward_layers=[16], fc_policy_layers=[16], fc_value_layers=[16],
                           fc_representation_layers=[], fc_hidden_state_layers=[16], support_limit=10)
    sp = BatchedSelfPlay(cfg, game, MuZeroNetwork(ncfg, seed=0), B, seed=seed, stale_observation=True)
    assert sp.mode == 'fused'
    return sp, A, OS


@pytest.mark.parametrize('game', ['tictactoe', 'cartpole'])
@pytest.mark.parametrize('n,stack_action', [(3, False), (2, True), (5, True), (1, True)])
def test_device_stack_is_the_history_stack(game, n, stack_action):
    from oracle import oracle as orc
    B, n_moves = 24, 40
    sp, A, OS = _selfplay(game, B, n, stack_action, seed=11)
    planes = 2 if stack_action else 1
    first = sp.envs.cur_obs.cpu().numpy().astype(np.float64)         # observation every game's history starts from
    hist_obs, hist_act = [[] for _ in range(B)], [[] for _ in range(B)]
    games_ended = 0
    for _ in range(n_moves):
        stack = sp.envs.search_obs.cpu().numpy().reshape(B, n * planes, OS)
        for b in range(B):
            want = orc.stack_before_move(first[b], np.array(hist_obs[b]).reshape(len(hist_obs[b]), OS),
                                         np.array(hist_act[b], dtype=np.int64), n, A, stack_action)
            np.testing.assert_array_equal(stack[b], want.astype(np.float32), err_msg=f'env {b} move {len(hist_act[b])}')
        out = sp.step(1.0)
        act, before, done = (out[k].cpu().numpy() for k in ('action', 'obs_before', 'done'))
        cur = sp.envs.cur_obs.cpu().numpy().astype(np.float64)
        for b in range(B):
            if done[b]:
                hist_obs[b], hist_act[b], first[b] = [], [], cur[b]
                games_ended += 1
            else:
                hist_obs[b].append(before[b].astype(np.float64))
                hist_act[b].append(int(act[b]))
    assert games_ended > B                                            # restarts were exercised
    sp.engine.close()


def test_stacking_needs_the_stale_rule_and_a_matching_network():
    from muzero_amd.self_play import BatchedSelfPlay, DeviceEnvs
    from muzero_amd.network import MuZeroNetwork
    with pytest.raises(ValueError, match='stale'):
        DeviceEnvs('cartpole', 4, torch.device('cuda'), 10, stale_observation=False, n_stacked_observations=2)
    cfg = make_config(False, 2, 4, 1, 0.997)
    cfg.max_moves, cfg.n_stacked_observations, cfg.stack_action = 10, 3, False
    ncfg = SimpleNamespace(network='mlp', observation_dim=[1, 1, 4], n_actions=2, n_stacked_observations=1,
                           encoding_size=8, fc_reward_layers=[16], fc_policy_layers=[16], fc_value_layers=[16],
                           fc_representation_layers=[], fc_hidden_state_layers=[16], support_limit=10)
    with pytest.raises(ValueError, match='stack'):
        BatchedSelfPlay(cfg, 'cartpole', MuZeroNetwork(ncfg, seed=0), 4)
    # the library refuses a stack without the stale rule as well
    sp, A, OS = _selfplay('cartpole', 8, 2, False, seed=1)
    sp.step(1.0)
    sp.envs.struct.stale_observation = 0
    with pytest.raises(ValueError, match='stale'):
        sp.step(1.0)
    sp.engine.close()
