"""CPU: the checker's restatement of GameHistory.make_target / stack_n_observations and the host
GameHistory mirror against vectors produced by the reference's own GameHistory."""
import os

import numpy as np

from oracle import oracle as orc
from muzero_amd.games.game import GameHistory

GOLD = os.path.join(os.path.dirname(__file__), 'golden')


def load_games(z, p, n_games, reanalysed):
    games = []
    for g in range(n_games):
        q = f'{p}g{g}_'
        d = {k: z[q + k] for k in ('observations', 'actions', 'rewards', 'to_plays', 'probs', 'root_values')}
        if reanalysed:
            d['re_probs'], d['re_root_values'] = z[q + 're_probs'], z[q + 're_root_values']
        games.append(d)
    return games


def history_of(d, A):
    h = GameHistory(lambda a: np.eye(A)[a][None], np.zeros_like(d['observations'][0]))
    for t in range(len(d['actions'])):
        h.save(d['observations'][t], int(d['actions'][t]), float(d['rewards'][t]), int(d['to_plays'][t]),
               [float(x) for x in d['probs'][t]], float(d['root_values'][t]))
    if 're_probs' in d:
        h.save_reanalysed_stats([list(map(float, r)) for r in d['re_probs']], [float(x) for x in d['re_root_values']])
    return h


def test_make_target_and_stacking_match_reference_batches():
    z = np.load(os.path.join(GOLD, 'targets.npz'))
    for ci in range(int(z['n_cases'])):
        p = f'c{ci}_'
        players, A, td, unroll, reanalysed, n_games, Bs = (int(x) for x in z[p + 'meta'])
        gamma = float(z[p + 'gamma'])
        games = load_games(z, p, n_games, reanalysed)
        for i in range(Bs):
            d, t = games[int(z[p + 'sel'][i])], int(z[p + 't'][i])
            rv = d['re_root_values'] if reanalysed else d['root_values']
            pr = d['re_probs'] if reanalysed else d['probs']
            v, r, pol = orc.make_target(d['rewards'], d['to_plays'], rv, pr, t, td, gamma, unroll)
            np.testing.assert_array_equal(v.astype(np.float32), z[p + 'batch_value'][i])
            np.testing.assert_array_equal(r.astype(np.float32), z[p + 'batch_reward'][i])
            np.testing.assert_array_equal(pol.astype(np.float32), z[p + 'batch_policy'][i])
            obs = orc.stack_observations(d['observations'], d['actions'], t, 1, A, False)
            np.testing.assert_array_equal(obs.astype(np.float32).reshape(-1), z[p + 'batch_obs'][i].reshape(-1))
            # the host mirror of GameHistory gives the same lists
            h = history_of(d, A)
            hv, hr, hp = h.make_target(t, td, gamma, unroll, A)
            np.testing.assert_array_equal(np.asarray(hv, np.float32), z[p + 'batch_value'][i])
            np.testing.assert_array_equal(np.asarray(hr, np.float32), z[p + 'batch_reward'][i])
            np.testing.assert_array_equal(np.asarray(hp, np.float32), z[p + 'batch_policy'][i])


def test_make_target_after_reanalyse_matches_reference():
    z = np.load(os.path.join(GOLD, 'reanalyse_tictactoe.npz'))
    for gi in range(int(z['n_games'])):
        p = f'g{gi}_'
        v, r, pol = orc.make_target(z[p + 'rewards'], z[p + 'to_plays'], z[p + 're_root_values'], z[p + 're_probs'],
                                    1, 5, 1, 5)
        np.testing.assert_array_equal(v, z[p + 'target_value'])
        np.testing.assert_array_equal(r, z[p + 'target_reward'])
        np.testing.assert_array_equal(pol, z[p + 'target_policy'])


def test_stacking_with_action_planes_matches_host_mirror():
    rng = np.random.default_rng(0)
    A, T = 4, 7
    d = dict(observations=rng.normal(size=(T, 1, 2, 3)), actions=rng.integers(0, A, size=T), rewards=np.zeros(T),
             to_plays=np.zeros(T, np.int64), probs=np.full((T, A), 0.25), root_values=np.zeros(T))
    h = history_of(d, A)
    for t in (0, 2, 6, -1):
        for n in (1, 3, 9):
            want = h.stack_n_observations(t, n, A, True)
            got = orc.stack_observations(d['observations'], d['actions'], t, n, A, True)
            np.testing.assert_array_equal(got.reshape(want.shape), want)


def test_stacking_matches_the_reference_fixture():
    """tests/golden/stacking.npz: the reference's GameHistory.stack_n_observations for 1..6 stacked observations with
    and without action planes - at every position of a game (replay buffer) and on the history so far before every
    move, the empty history included (self-play).  The checker and the host GameHistory must both reproduce it."""
    z = np.load(os.path.join(GOLD, 'stacking.npz'))
    for ci in range(int(z['n_cases'])):
        p = f'c{ci}_'
        A, T = (int(x) for x in z[p + 'meta'])
        obs, acts = z[p + 'observations'], z[p + 'actions']
        d = dict(observations=obs, actions=acts, rewards=np.zeros(T), to_plays=np.zeros(T, np.int64),
                 probs=np.full((T, A), 1.0 / A), root_values=np.zeros(T))
        h = history_of(d, A)
        for n, sa in ((1, True), (2, False), (3, True), (4, False), (6, True)):
            q = f'{p}n{n}a{int(sa)}_'
            at, before = z[q + 'at'], z[q + 'before_move']
            for t in range(T):
                got = orc.stack_observations(obs, acts, t, n, A, sa)
                np.testing.assert_array_equal(got.reshape(at[t].shape), at[t])
                np.testing.assert_array_equal(h.stack_n_observations(t, n, A, sa), at[t])
            for t in range(T + 1):
                got = orc.stack_before_move(obs[0], obs[:t], acts[:t], n, A, sa)
                np.testing.assert_array_equal(got.reshape(before[t].shape), before[t])


def test_scalar_to_support_matches_reference():
    z = np.load(os.path.join(GOLD, 'targets.npz'))
    for ci in range(int(z['n_cases'])):
        p = f'c{ci}_'
        L = int(z[p + 'support_limit'])
        for n in ('value', 'reward'):
            np.testing.assert_array_equal(orc.scalar_to_support(z[p + 'batch_' + n], L), z[p + n + '_support'])
    # clamping at both ends and exact integers
    x = np.array([1e9, -1e9, 0.0, 3.0, -3.0], np.float32)
    out = orc.scalar_to_support(x, 2)
    assert out[0].tolist() == [0, 0, 0, 0, 1] and out[1].tolist() == [1, 0, 0, 0, 0] and out[2].tolist() == [0, 0, 1, 0, 0]
    np.testing.assert_allclose(out.sum(-1), 1.0, rtol=0, atol=1e-6)
