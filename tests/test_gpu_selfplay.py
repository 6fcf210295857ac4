"""GPU: batched self-play (search + device environment step + trajectory record)."""
from types import SimpleNamespace

import ctypes as C
import numpy as np
import pytest
import torch

from helpers import load, make_config
from muzero_amd import hashnet, noise as mznoise

pytestmark = pytest.mark.gpu


def test_tictactoe_selfplay_reproduces_reference_games():
    """Whole games against the reference's SelfPlay.play loop (recorded with the reference's MCTS,
    TicTacToe and GameHistory): same actions, rewards, players, searched-from observations (the
    one-move-stale rule), policies and root values, move by move."""
    from muzero_amd.self_play import BatchedSelfPlay
    from muzero_amd.engine import NOISE_BUFFER
    z = load('selfplay_tictactoe.npz')
    for gi in range(int(z['n_games'])):
        p = f'g{gi}_'
        gumbel, seed, idx, length = [int(x) for x in z[p + 'meta']]
        cfg = make_config(gumbel, 9, 20, 2, 1, alpha=0.2)
        cfg.max_moves = 9
        net = hashnet.HashNetwork(hashnet.HashNetParams(9, 8, 3, 1, 1), obs_hash=True)
        sp = BatchedSelfPlay(cfg, 'tictactoe', net, 1, seed=seed, noise_mode=NOISE_BUFFER)
        np.random.seed(seed)
        calls = 0
        max_calls = 20 * 21 // 2
        for k in range(length):
            assert np.array_equal(sp.envs.search_obs.cpu().numpy().reshape(3, 3, 3), z[p + 'search_obs'][k]), (gi, k)
            if gumbel:
                root_noise, noise = (None, torch.as_tensor(np.random.gumbel(size=9)[None])), None
            else:
                root_noise = (torch.as_tensor(np.random.dirichlet([0.2] * 9)[None]), None)
                noise = np.stack([mznoise.noise_row(seed, idx, calls + i, 9) for i in range(max_calls)])[None]
            u = None
            out = sp.step(1.0, root_noise=root_noise, u=None if gumbel else np.zeros(1), noise=noise)
            if not gumbel:  # replay the final sampling with the uniform the reference's np.random.choice drew
                u = np.random.random_sample()
                _, nc = sp.engine.stats()
                calls += int(nc[0])
                probs = out['probs'][0].cpu().numpy()
                cdf = np.cumsum(probs); cdf /= cdf[-1]
                want = int(np.searchsorted(cdf, u, side='right'))
                # the device already stepped with u = 0; only the recorded policy/value are compared here
                np.testing.assert_allclose(probs, z[p + 'probs'][k], rtol=1e-12, atol=1e-300)
                assert float(out['value'][0]) == float(z[p + 'root_values'][k])
                assert want == int(z[p + 'actions'][k])
                break  # sampling differs from move 1 on; the PUCT game is checked on its first move only
            assert int(out['action'][0]) == int(z[p + 'actions'][k]), (gi, k)
            assert float(out['reward'][0]) == float(z[p + 'rewards'][k])
            assert int(out['to_play'][0]) == int(z[p + 'to_plays'][k])
            assert float(out['value'][0]) == float(z[p + 'root_values'][k])
            np.testing.assert_allclose(out['probs'][0].cpu().numpy(), z[p + 'probs'][k], rtol=1e-12, atol=1e-300)
            assert np.array_equal(out['obs_before'][0].cpu().numpy().reshape(3, 3, 3), z[p + 'observations'][k])
            assert bool(out['done'][0]) == (k == length - 1)
        sp.engine.close()


def test_puct_selfplay_game_with_reference_sampling():
    """PUCT games end to end: the sampling uniform is the one np.random.choice drew in the reference."""
    from muzero_amd.self_play import BatchedSelfPlay
    from muzero_amd.engine import NOISE_BUFFER
    z = load('selfplay_tictactoe.npz')
    for gi in range(int(z['n_games'])):
        p = f'g{gi}_'
        gumbel, seed, idx, length = [int(x) for x in z[p + 'meta']]
        if gumbel:
            continue
        cfg = make_config(False, 9, 20, 2, 1, alpha=0.2)
        cfg.max_moves = 9
        net = hashnet.HashNetwork(hashnet.HashNetParams(9, 8, 3, 1, 1), obs_hash=True)
        # the reference's stream per move: dirichlet, (tie-break uniforms are counter-based), choice's uniform.
        # Run the search twice per move: once to learn how many tie-break rows it consumed is not needed --
        # rows are indexed by the running call counter, so pre-compute generously.
        rs = np.random.RandomState(seed)
        sp = BatchedSelfPlay(cfg, 'tictactoe', net, 1, seed=seed, noise_mode=NOISE_BUFFER)
        calls, max_calls = 0, 20 * 21 // 2
        for k in range(length):
            di = rs.dirichlet([0.2] * 9)
            u = rs.random_sample()
            noise = np.stack([mznoise.noise_row(seed, idx, calls + i, 9) for i in range(max_calls)])[None]
            out = sp.step(1.0, root_noise=(torch.as_tensor(di[None]), None), u=np.array([u]), noise=noise)
            _, nc = sp.engine.stats()
            calls += int(nc[0])
            assert int(out['action'][0]) == int(z[p + 'actions'][k]), (gi, k)
            assert float(out['value'][0]) == float(z[p + 'root_values'][k])
            assert float(out['reward'][0]) == float(z[p + 'rewards'][k])
            assert bool(out['done'][0]) == (k == length - 1)
        sp.engine.close()


def test_cartpole_fused_selfplay_env_matches_checker():
    """Fused path (3 launches per move) on CartPole: every transition the device made is replayed by
    the checker's CartPole step from the recorded observations and actions; restarts happen on
    termination; histories come back in GameHistory layout."""
    from muzero_amd.self_play import BatchedSelfPlay
    from muzero_amd.network import MuZeroNetwork
    from oracle import oracle as orc
    ncfg = SimpleNamespace(network='mlp', observation_dim=[1, 1, 4], n_actions=2, n_stacked_observations=1,
                           encoding_size=8, fc_reward_layers=[16], fc_policy_layers=[16], fc_value_layers=[16],
                           fc_representation_layers=[], fc_hidden_state_layers=[16], support_limit=10)
    cfg = make_config(False, 2, 16, 1, 0.997)
    cfg.max_moves = 30
    B = 64
    sp = BatchedSelfPlay(cfg, 'cartpole', MuZeroNetwork(ncfg, seed=0), B, seed=5, stale_observation=False)
    assert sp.mode == 'fused'
    L = orc.lib()
    state = sp.envs.state.cpu().numpy().copy()
    n_done = 0
    for step in range(45):
        out = sp.step(1.0)
        act, done = out['action'].cpu().numpy(), out['done'].cpu().numpy()
        obs_before = out['obs_before'].cpu().numpy()
        new_state = sp.envs.state.cpu().numpy()
        for b in range(B):
            assert np.allclose(obs_before[b], state[b].astype(np.float32), atol=1e-6)
            s = state[b].copy()
            rew, obs = C.c_double(), np.zeros(4, np.float32)
            term = L.orc_cartpole_step(s.ctypes.data_as(C.c_void_p), int(act[b]), C.byref(rew), obs.ctypes.data_as(C.c_void_p))
            if done[b]:
                n_done += 1
                assert term or int(sp.envs.moves[b]) == 0
                assert np.all(np.abs(new_state[b]) <= 0.05)  # restarted from a fresh initial state
            else:
                assert not term
                np.testing.assert_allclose(new_state[b], s, rtol=0, atol=1e-12)
        state = new_state.copy()
    assert n_done > 0
    games = sp.play(40)
    assert len(games) > 0 and all(len(g) <= 31 for g in games)
    g = games[0]
    vt, rt, pt = g.make_target(0, 5, 0.997, 5, 2)
    assert len(vt) == 6 and g.observations[0].shape == (1, 1, 4)
    sp.engine.close()


def test_tictactoe_device_env_matches_checker_on_random_play():
    from muzero_amd.self_play import BatchedSelfPlay
    from oracle import oracle as orc
    cfg = make_config(True, 9, 8, 2, 1)
    cfg.max_moves = 9
    B = 128
    net = hashnet.HashNetwork(hashnet.HashNetParams(9, 4, 3, 1, 1), obs_hash=True)
    sp = BatchedSelfPlay(cfg, 'tictactoe', net, B, seed=2)
    L = orc.lib()
    boards = [np.zeros(9) for _ in range(B)]
    movers = [C.c_int(-1) for _ in range(B)]
    winners = [C.c_int(-2) for _ in range(B)]
    moves = np.zeros(B, int)
    for step in range(14):
        out = sp.step(1.0)
        act, done, rew = out['action'].cpu().numpy(), out['done'].cpu().numpy(), out['reward'].cpu().numpy()
        cur, legal = sp.envs.cur_obs.cpu().numpy(), sp.envs.legal.cpu().numpy()
        for b in range(B):
            r, obs, lg = C.c_double(), np.zeros(27), np.zeros(9)
            t = L.orc_tictactoe_step(boards[b].ctypes.data_as(C.c_void_p), C.byref(movers[b]), C.byref(winners[b]),
                                     int(act[b]), C.byref(r), obs.ctypes.data_as(C.c_void_p), lg.ctypes.data_as(C.c_void_p))
            moves[b] += 1
            assert float(rew[b]) == r.value
            assert bool(done[b]) == bool(t or moves[b] > 9)
            if done[b]:
                boards[b][:] = 0; movers[b] = C.c_int(-1); winners[b] = C.c_int(-2); moves[b] = 0
                obs = np.concatenate([np.zeros(18), -np.ones(9)]); lg = np.ones(9)
            assert np.array_equal(cur[b], obs.astype(np.float32)) and np.array_equal(legal[b], lg.astype(np.uint8))
    sp.engine.close()


def test_cartpole_selfplay_at_the_benchmarked_size():
    """The self-play step that the throughput figure is quoted on - 8 192 CartPole environments per GPU, 50 PUCT
    simulations, fused action + environment kernel writing the device move log (self_play.py:99-113 per environment) -
    checked by its properties at that size: every transition of a 256-environment sample (first, last and spread
    environments) is the checker's CartPole step from the state before it; every row of the move log equals the
    per-step outputs of the same move, for all environments; the games the device lists are the episode ends the host
    counted, and their lengths add up."""
    import bench
    from muzero_amd.self_play import BatchedSelfPlay
    from muzero_amd.network import MuZeroNetwork
    from oracle import oracle as orc
    B, N = 8192, 48
    cfg = bench.search_config(False, 2, 50, 1, 0.997)
    cfg.max_moves = 500
    sp = BatchedSelfPlay(cfg, 'cartpole', MuZeroNetwork(bench.cartpole_net_config(), seed=0), B, seed=3)
    assert sp.mode == 'fused'
    L = orc.lib()
    sample = np.unique(np.concatenate([np.arange(0, B, B // 250), [1, 63, 64, B - 65, B - 64, B - 1]]))
    log = sp.record(N)
    state = sp.envs.state.cpu().numpy().copy()
    ends = np.zeros(B, np.int64)
    for k in range(N):
        out = sp.step(1.0)
        new_state = sp.envs.state.cpu().numpy()
        act, done = out['action'].cpu().numpy(), out['done'].cpu().numpy()
        ends += done != 0
        for b in sample:
            s = state[b].copy()
            rew, obs = C.c_double(), np.zeros(4, np.float32)
            term = L.orc_cartpole_step(s.ctypes.data_as(C.c_void_p), int(act[b]), C.byref(rew), obs.ctypes.data_as(C.c_void_p))
            assert float(out['reward'][b]) == rew.value
            if done[b]:
                assert term or int(sp.envs.moves[b]) == 0
                assert np.all(np.abs(new_state[b]) <= 0.05)      # restarted from a fresh initial state
            else:
                assert not term
                np.testing.assert_allclose(new_state[b], s, rtol=0, atol=1e-12)
        # the log's row of this move against the step's own outputs, every environment
        assert torch.equal(log.observations[k], out['obs_before'].reshape(B, -1))
        assert torch.equal(log.actions[k], out['action']) and torch.equal(log.to_plays[k], out['to_play'])
        assert torch.equal(log.rewards[k], out['reward'].to(torch.float64))
        assert torch.equal(log.root_values[k], out['value']) and torch.equal(log.probs[k], out['probs'])
        assert torch.equal(log.done[k], out['done'])
        state = new_state.copy()
    assert ends.sum() > B // 4                                     # episodes did end inside the window
    store = sp.harvest()
    assert len(store) == int(ends.sum())
    assert int(store.lengths.sum()) <= B * N and int(store.lengths.min()) >= 1
    # a game's positions are rows of the log: the first listed game of environment 0 starts at its row 0
    sp.engine.close()
