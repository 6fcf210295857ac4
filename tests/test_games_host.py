"""CPU: host game classes, GameHistory and the checker's environment steps against traces and targets
recorded from the reference's own games/ classes."""
import ctypes as C

import numpy as np

from helpers import load
from muzero_amd.games import TicTacToe, CartPole, GameHistory
from oracle import oracle as orc


def test_tictactoe_host_and_checker_match_reference_traces():
    z = load('misc.npz')
    L = orc.lib()
    for i in range(int(z['ttt_n'])):
        g = TicTacToe()
        board, mover, winner = np.zeros(9), C.c_int(-1), C.c_int(-2)
        for k, a in enumerate(z[f'ttt{i}_moves']):
            assert np.array_equal(g.legal_actions(), z[f'ttt{i}_legal'][k]) and g.to_play() == z[f'ttt{i}_tp'][k]
            o, r, t = g.step(int(a))
            assert np.array_equal(o, z[f'ttt{i}_obs'][k]) and r == z[f'ttt{i}_rew'][k] and t == bool(z[f'ttt{i}_term'][k])
            rew, obs, legal = C.c_double(), np.zeros(27), np.zeros(9)
            t2 = L.orc_tictactoe_step(board.ctypes.data_as(C.c_void_p), C.byref(mover), C.byref(winner), int(a),
                                      C.byref(rew), obs.ctypes.data_as(C.c_void_p), legal.ctypes.data_as(C.c_void_p))
            assert np.array_equal(obs.reshape(3, 3, 3), z[f'ttt{i}_obs'][k]) and rew.value == r and bool(t2) == t
    # trace 1 ends with a move that completes a line AND fills the board: the reference scores a tie
    assert z['ttt1_term'][-1] and z['ttt1_rew'][-1] == 0


def test_game_history_targets_match_reference():
    z = load('selfplay_tictactoe.npz')
    for gi in range(int(z['n_games'])):
        p = f'g{gi}_'
        g = TicTacToe()
        h = GameHistory(g.action_encoder, g.reset())
        for k in range(len(z[p + 'actions'])):
            assert np.array_equal(h.stack_n_observations(-1, 1, 9, False), z[p + 'search_obs'][k])
            h.save(z[p + 'observations'][k], int(z[p + 'actions'][k]), float(z[p + 'rewards'][k]),
                   int(z[p + 'to_plays'][k]), list(z[p + 'probs'][k]), float(z[p + 'root_values'][k]))
        vt, rt, pt = h.make_target(0, 5, 1, 5, 9)
        assert np.array_equal(np.array(vt, float), z[p + 'target_value'])
        assert np.array_equal(np.array(rt, float), z[p + 'target_reward'])
        assert np.array_equal(np.array(pt, float), z[p + 'target_policy'])
        assert h.compute_return(1, 0, 2) == z[p + 'return_p0'] and h.compute_return(1, 1, 2) == z[p + 'return_p1']
    # n-step TD branch and stacking with action planes (no reference fixture: structural checks)
    h = GameHistory(lambda a: a, np.zeros((1, 1, 4)))
    for t in range(6):
        h.save(np.full((1, 1, 4), float(t)), t % 2, 1.0, 0, [0.5, 0.5], 10.0 + t)
    vt, rt, pt = h.make_target(2, 2, 0.5, 3, 2)
    assert vt[0] == 1.0 + 0.5 * 1.0 + 0.25 * 14.0 and len(vt) == 4 and rt[0] == 1.0
    s = h.stack_n_observations(3, 2, 2, True)
    assert s.shape == (4, 1, 4) and s[0, 0, 0] == 3.0 and s[1, 0, 0] == 0.5 and s[2, 0, 0] == 2.0


def test_cartpole_host_matches_checker_spec():
    """CartPole is a specification restatement (gymnasium is unpinned and absent: parity unpinned);
    the host class and the checker agree with each other and with the spec's known first step."""
    g = CartPole(seed=3)
    g.reset()
    s = g.state.copy()
    L = orc.lib()
    for k in range(40):
        a = k % 2
        o, r, t = g.step(a)
        rew, obs = C.c_double(), np.zeros(4, np.float32)
        t2 = L.orc_cartpole_step(s.ctypes.data_as(C.c_void_p), a, C.byref(rew), obs.ctypes.data_as(C.c_void_p))
        assert np.allclose(o.reshape(-1), obs, atol=1e-7) and r == 1.0 and bool(t2) == t
    # from rest, pushing right: xacc = F/m_total corrected by the pole term -> documented values
    s = np.zeros(4)
    L.orc_cartpole_step(s.ctypes.data_as(C.c_void_p), 1, C.byref(rew), obs.ctypes.data_as(C.c_void_p))
    assert np.allclose(s, [0.0, 0.02 * 9.7560975609756, 0.0, 0.02 * -14.6341463414634], atol=1e-12)
