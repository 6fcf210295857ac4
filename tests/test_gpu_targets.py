"""GPU: training batches built on the device (mz_make_targets) against the batches the reference's
GameHistory / ReplayBuffer.sample body produced, and batched Reanalyse against the reference's
per-position searches."""
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from test_targets_oracle import GOLD, history_of, load_games
from muzero_amd import hashnet

pytestmark = pytest.mark.gpu


def test_device_batches_equal_reference_batches():
    from muzero_amd.targets import TrajectoryStore
    z = np.load(os.path.join(GOLD, 'targets.npz'))
    for ci in range(int(z['n_cases'])):
        p = f'c{ci}_'
        players, A, td, unroll, reanalysed, n_games, Bs = (int(x) for x in z[p + 'meta'])
        gamma = float(z[p + 'gamma'])
        hs = [history_of(d, A) for d in load_games(z, p, n_games, reanalysed)]
        store = TrajectoryStore(hs, A)
        obs, act, val, rew, pol = store.make_targets(z[p + 'sel'], z[p + 't'], z[p + 'absorbed'], td, gamma, unroll)
        np.testing.assert_array_equal(val.cpu().numpy(), z[p + 'batch_value'])
        np.testing.assert_array_equal(rew.cpu().numpy(), z[p + 'batch_reward'])
        np.testing.assert_array_equal(pol.cpu().numpy(), z[p + 'batch_policy'])
        np.testing.assert_array_equal(obs.cpu().numpy().reshape(Bs, -1), z[p + 'batch_obs'].reshape(Bs, -1))
        np.testing.assert_array_equal(act.cpu().numpy().reshape(Bs, unroll + 1, -1),
                                      z[p + 'batch_actions'].reshape(Bs, unroll + 1, -1))
        # categorical targets as the trainer builds them from the batch
        from muzero_amd.targets import scalar_to_support
        L = int(z[p + 'support_limit'])
        np.testing.assert_array_equal(scalar_to_support(val, L).cpu().numpy(), z[p + 'value_support'])
        np.testing.assert_array_equal(scalar_to_support(rew, L).cpu().numpy(), z[p + 'reward_support'])
        # sampling consumes np.random exactly like ReplayBuffer.sample
        np.random.seed(11 + ci)
        obs2, act2, val2, rew2, pol2 = store.sample(Bs, td, gamma, unroll)
        np.testing.assert_array_equal(val2.cpu().numpy(), z[p + 'batch_value'])
        np.testing.assert_array_equal(act2.cpu().numpy().reshape(Bs, unroll + 1, -1),
                                      z[p + 'batch_actions'].reshape(Bs, unroll + 1, -1))


def test_device_targets_vs_checker_on_long_games():
    from muzero_amd.targets import TrajectoryStore
    from oracle import oracle as orc
    rng = np.random.default_rng(5)
    A, G = 6, 40
    games = []
    for g in range(G):
        T = int(rng.integers(1, 400))
        games.append(dict(observations=rng.normal(size=(T, 1, 1, 4)), actions=rng.integers(0, A, size=T),
                          rewards=rng.normal(size=T), to_plays=np.arange(T) % 2, probs=rng.dirichlet([1] * A, size=T),
                          root_values=rng.normal(size=T)))
    store = TrajectoryStore([history_of(d, A) for d in games], A)
    n, td, unroll, gamma = 512, 10, 5, 0.997
    sel = rng.integers(0, G, size=n)
    ts = np.array([rng.integers(0, len(games[g]['actions'])) for g in sel])
    ab = np.array([[rng.integers(0, len(games[g]['actions'])) for _ in range(unroll + 1)] for g in sel])
    for n_stack, stack_action in ((1, False), (3, True)):
        obs, act, val, rew, pol = store.make_targets(sel, ts, ab, td, gamma, unroll, n_stack, stack_action)
        val, rew, pol, obs = val.cpu().numpy(), rew.cpu().numpy(), pol.cpu().numpy(), obs.cpu().numpy()
        for i in range(0, n, 7):
            d = games[sel[i]]
            v, r, p = orc.make_target(d['rewards'], d['to_plays'], d['root_values'], d['probs'], ts[i], td, gamma, unroll)
            np.testing.assert_array_equal(val[i], v.astype(np.float32))
            np.testing.assert_array_equal(rew[i], r.astype(np.float32))
            np.testing.assert_array_equal(pol[i], p.astype(np.float32))
            o = orc.stack_observations(d['observations'], d['actions'], ts[i], n_stack, A, stack_action)
            np.testing.assert_array_equal(obs[i].reshape(-1), o.astype(np.float32).reshape(-1))
    with pytest.raises(ValueError):
        store.make_targets([G], [0], np.zeros((1, unroll + 1)), td, gamma, unroll)
    with pytest.raises(ValueError):
        store.make_targets([0], [10 ** 6], np.zeros((1, unroll + 1)), td, gamma, unroll)


@pytest.mark.parametrize('mcts_target_value', [True, False])
def test_batched_reanalyse_equals_reference_loop(mcts_target_value):
    from muzero_amd.reanalyse import BatchedReanalyser
    from muzero_amd.games.tictactoe import TicTacToe
    z = np.load(os.path.join(GOLD, 'reanalyse_tictactoe.npz'))
    for gi in range(int(z['n_games'])):
        p = f'g{gi}_'
        gumbel, seed, _, T = (int(x) for x in z[p + 'meta'])
        cfg = SimpleNamespace(simulations=20, n_actions=9, gamma=1, players=2, gumbel=bool(gumbel),
                              max_considered_actions=16, gumbel_scale=1.0, c_visit=50, c_scale=1.0, c_base=19625,
                              c_init=1.25, dirichlet_alpha=0.2, exploration_frac=0.25, n_stacked_observations=1,
                              stack_action=False, max_moves=9, mcts_target_value=mcts_target_value)
        d = {k: z[p + k] for k in ('observations', 'actions', 'rewards', 'to_plays', 'probs', 'root_values')}
        h = history_of(d, 9)
        target_net = hashnet.HashNetwork(hashnet.HashNetParams(9, 8, 4, 1, 1), obs_hash=True)
        re = BatchedReanalyser(TicTacToe(), cfg, target_net, seed=100 + seed)
        np.random.seed(100 + seed)
        (probs, values), = re.reanalyse_games([h])
        np.testing.assert_allclose(np.asarray(probs), z[p + 're_probs'], rtol=1e-12, atol=0)
        want_v = z[p + 're_root_values'] if mcts_target_value else z[p + 'target_net_values']
        np.testing.assert_array_equal(np.asarray(values), want_v)
        assert np.random.random_sample() == float(z[p + 'next_random_sample'])   # same draws consumed
        assert h.reanalysed_root_values == values
        if mcts_target_value:
            vt, rt, pt = h.make_target(1, 5, 1, 5, 9)
            np.testing.assert_array_equal(np.asarray(vt), z[p + 'target_value'])
            np.testing.assert_allclose(np.asarray(pt), z[p + 'target_policy'], rtol=1e-12, atol=0)
