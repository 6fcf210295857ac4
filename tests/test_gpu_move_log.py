"""GPU: the device move log.  The action + environment kernel appends every move's record to it, the game
table is built on the device, and training batches are made from the log in place - compared bit for bit with
batches made from the same games gathered on the host move by move (the reference's GameHistory route)."""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from helpers import make_config

pytestmark = pytest.mark.gpu


def _selfplay(game, B, seed):
    from muzero_amd.self_play import BatchedSelfPlay
    from muzero_amd.network import MuZeroNetwork
    if game == 'cartpole':
        dim, A, cfg = [1, 1, 4], 2, make_config(False, 2, 16, 1, 0.997)
        cfg.max_moves = 30
    else:
        dim, A, cfg = [3, 3, 3], 9, make_config(True, 9, 16, 2, 1)
        cfg.max_moves = 9
    ncfg = SimpleNamespace(network='mlp', observation_dim=dim, n_actions=A, n_stacked_observations=1,
                           encoding_size=8, fc_reward_layers=[16], fc_policy_layers=[16], fc_value_layers=[16],
                           fc_representation_layers=[], fc_hidden_state_layers=[16], support_limit=10)
    sp = BatchedSelfPlay(cfg, game, MuZeroNetwork(ncfg, seed=0), B, seed=seed, stale_observation=(game != 'cartpole'))
    assert sp.mode == 'fused'
    return sp, cfg, A, tuple(dim)


def _host_games(records, B):
    """Finished games from per-move host copies of the step records, environment by environment."""
    games = []
    for b in range(B):
        start = 0
        for k, rec in enumerate(records):
            if rec['done'][b]:
                rows = records[start:k + 1]
                games.append(dict(observations=np.stack([r['obs_before'][b] for r in rows]),
                                  actions=np.array([r['action'][b] for r in rows]),
                                  rewards=np.array([float(r['reward'][b]) for r in rows]),
                                  to_plays=np.array([r['to_play'][b] for r in rows]),
                                  probs=np.stack([r['probs'][b] for r in rows]),
                                  root_values=np.array([r['value'][b] for r in rows])))
                start = k + 1
    return games


@pytest.mark.parametrize('game', ['cartpole', 'tictactoe'])
def test_batches_from_the_log_equal_batches_from_host_gathered_games(game):
    from muzero_amd.targets import TrajectoryStore
    from test_targets_oracle import history_of
    from oracle import oracle as orc
    B, n_steps = 48, 50
    sp, cfg, A, shape = _selfplay(game, B, seed=3)
    sp.record(n_steps)
    records = []
    for _ in range(n_steps):
        out = sp.step(1.0)
        records.append({k: v.cpu().numpy().copy() for k, v in out.items()})
    games = _host_games(records, B)
    assert len(games) > B   # several games per environment, some still open at the end
    for d in games:
        d['observations'] = d['observations'].reshape((-1,) + shape)
    host_store = TrajectoryStore([history_of(d, A) for d in games], A)
    log_store = sp.harvest()
    np.testing.assert_array_equal(log_store.lengths, host_store.lengths)
    rng = np.random.default_rng(1)
    n, td, unroll, gamma = 256, 5, 4, float(cfg.gamma)
    sel = rng.integers(0, len(games), size=n)
    ts = np.array([rng.integers(0, len(games[g]['actions'])) for g in sel])
    ab = np.array([[rng.integers(0, len(games[g]['actions'])) for _ in range(unroll + 1)] for g in sel])
    for n_stack, stack_action in ((1, False), (3, True)):
        a = log_store.make_targets(sel, ts, ab, td, gamma, unroll, n_stack, stack_action)
        b = host_store.make_targets(sel, ts, ab, td, gamma, unroll, n_stack, stack_action)
        for x, y in zip(a, b):
            np.testing.assert_array_equal(x.cpu().numpy(), y.cpu().numpy())
    # ... and the checker's GameHistory.make_target on the host-gathered games
    _, _, val, rew, pol = log_store.make_targets(sel, ts, ab, td, gamma, unroll)
    val, rew, pol = val.cpu().numpy(), rew.cpu().numpy(), pol.cpu().numpy()
    for i in range(0, n, 5):
        d = games[sel[i]]
        v, r, p = orc.make_target(d['rewards'], d['to_plays'], d['root_values'], d['probs'], ts[i], td, gamma, unroll)
        np.testing.assert_array_equal(val[i], v.astype(np.float32))
        np.testing.assert_array_equal(rew[i], r.astype(np.float32))
        np.testing.assert_array_equal(pol[i], p.astype(np.float32))
    # sampling consumes np.random like ReplayBuffer.sample whichever store holds the games
    np.random.seed(4)
    s1 = log_store.sample(64, td, gamma, unroll)
    np.random.seed(4)
    s2 = host_store.sample(64, td, gamma, unroll)
    for x, y in zip(s1, s2):
        np.testing.assert_array_equal(x.cpu().numpy(), y.cpu().numpy())
    # reanalysed statistics written to a log-backed store land on the game's own positions
    g = int(sel[0])
    T = int(log_store.lengths[g])
    new_p, new_v = rng.dirichlet([1] * A, size=T), rng.normal(size=T)
    log_store.save_reanalysed_stats(g, new_p, new_v)
    host_store.save_reanalysed_stats(g, new_p, new_v)
    a = log_store.make_targets(sel, ts, ab, td, gamma, unroll)
    b = host_store.make_targets(sel, ts, ab, td, gamma, unroll)
    for x, y in zip(a, b):
        np.testing.assert_array_equal(x.cpu().numpy(), y.cpu().numpy())
    # the GameHistory view of the log (one copy, no per-move transfers)
    enc = (lambda a: np.eye(9)[a].reshape(3, 3)) if game == 'tictactoe' else (lambda a: np.eye(2)[a])
    hs = log_store.histories(enc)
    assert len(hs) == len(games)
    for h, d in zip(hs, games):
        assert h.actions == [int(x) for x in d['actions']] and h.to_plays == [int(x) for x in d['to_plays']]
        np.testing.assert_array_equal(np.stack(h.observations), d['observations'].astype(np.float32))
        np.testing.assert_array_equal(np.stack(h.action_probabilities), d['probs'])
        assert h.root_values == [float(x) for x in d['root_values']] and h.rewards == [float(x) for x in d['rewards']]
    sp.engine.close()


@pytest.mark.parametrize('game', ['cartpole', 'tictactoe'])
def test_consecutive_windows_list_whole_games_only(game):
    """SelfPlay.play stores a game when it has ended, from its first move (self_play.py:77-115).  A window that
    opens in the middle of a game does not list that game's tail as a game: every listed game begins where an
    environment was reset, and the games of a second window are the end-of-game flags minus one per environment that
    was mid-game at its start and finished that game inside it."""
    B, n_steps = 48, 23
    sp, cfg, A, shape = _selfplay(game, B, seed=5)
    sp.play(n_steps)                                   # leaves most environments in the middle of a game
    open_before = (sp.envs.moves != 0).cpu().numpy()
    assert open_before.sum() > B // 4
    sp.record(n_steps)
    records = []
    for _ in range(n_steps):
        out = sp.step(1.0)
        records.append({k: v.cpu().numpy().copy() for k, v in out.items()})
    np.testing.assert_array_equal(sp.log.open_at_start.cpu().numpy().astype(bool), open_before)
    done = np.stack([r['done'] for r in records]).astype(bool)            # [row][env]
    base, length = (t.cpu().numpy() for t in sp.log.games())
    host = []                                                             # (first row, env, length) of whole games
    for b in range(B):
        start, whole = 0, not open_before[b]
        for k in range(n_steps):
            if done[k, b]:
                if whole:
                    host.append((start, b, k - start + 1))
                whole, start = True, k + 1
    assert len(host) == done.sum() - (open_before & done.any(0)).sum()
    np.testing.assert_array_equal(base, [r * B + b for r, b, _ in host])
    np.testing.assert_array_equal(length, [n for _, _, n in host])
    # a listed game that begins after row 0 begins on the row after an end-of-game flag; one at row 0 in an
    # environment that had just been reset
    for r, b, _ in host:
        assert (r == 0 and not open_before[b]) or done[r - 1, b]
    if game == 'tictactoe':   # every stored game starts from the empty board
        store = sp.harvest()
        first = store.observations[torch.as_tensor(base, device=store.device)].cpu().numpy().reshape(-1, 3, 3, 3)
        assert not first[:, :2].any()
    sp.engine.close()


def test_play_returns_the_games_of_its_window_and_a_full_log_is_refused():
    from muzero_amd.engine import EngineError
    sp, cfg, A, shape = _selfplay('cartpole', 16, seed=9)
    games = sp.play(45)
    assert len(games) >= 16 and all(1 <= len(g) <= 31 for g in games)
    assert games[0].observations[0].shape == shape
    vt, rt, pt = games[0].make_target(0, 5, 0.997, 5, 2)
    assert len(vt) == 6
    assert sp.play(3, record=False) == [] and sp.log is None
    sp.record(2)
    sp.step(1.0)
    sp.step(1.0)
    with pytest.raises(EngineError, match='full'):
        sp.step(1.0)
    # the library refuses a row outside the log as well
    with pytest.raises(ValueError, match='full'):
        sp.engine.act_env_step(sp.envs.struct, 1.0, sp._u_block[0], sp.out, sp.log.struct, 2)
    sp.engine.close()


def test_move_log_arguments_are_checked():
    """A log shaped for another batch, an incomplete log and bad row counts are refused with MZ_ERR_ARG before anything
    is launched."""
    import ctypes as C
    from muzero_amd.engine import MzMoveLog, lib
    from muzero_amd.targets import MoveLog
    sp, cfg, A, shape = _selfplay('cartpole', 16, seed=2)
    sp.step(1.0)   # creates the trees
    dev = sp.device
    other = MoveLog(4, 8, shape, A, dev)          # 8 environments, the engine has 16
    with pytest.raises(ValueError, match='another batch'):
        sp.engine.act_env_step(sp.envs.struct, 1.0, sp._u_block[0], sp.out, other.struct, 0)
    log = MoveLog(4, 16, shape, A, dev)
    broken = MzMoveLog(4, 16, log.obs_size, A, log.observations.data_ptr(), log.actions.data_ptr(), None,
                       log.to_plays.data_ptr(), log.root_values.data_ptr(), log.probs.data_ptr(), log.done.data_ptr())
    with pytest.raises(ValueError, match='incomplete'):
        sp.engine.act_env_step(sp.envs.struct, 1.0, sp._u_block[0], sp.out, broken, 0)
    with pytest.raises(ValueError, match='full'):
        sp.engine.act_env_step(sp.envs.struct, 1.0, sp._u_block[0], sp.out, log.struct, -1)
    count = torch.zeros(16, dtype=torch.int32, device=dev)
    assert lib().mz_log_count_games(C.byref(log.struct), 5, C.c_void_p(count.data_ptr()), None) == -1   # more rows than the log has
    assert lib().mz_log_count_games(C.byref(log.struct), 2, None, None) == -1
    assert lib().mz_log_list_games(C.byref(log.struct), 2, None, None, None, None) == -1
    # an empty window lists no games
    log.rows = 0
    base, length = log.games()
    assert base.numel() == 0 and length.numel() == 0
    sp.engine.close()
