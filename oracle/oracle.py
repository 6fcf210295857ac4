"""ctypes front end of the CPU checker (oracle/mcts_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, `__graft_entry__.smoke()` and
bench.py's cpu_baseline leg.  The product package (muzero_amd/) never imports it.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from types import SimpleNamespace

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, 'liboracle.so')
    src = os.path.join(_HERE, 'mcts_oracle.c')
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(['make', '-C', _HERE] + (['-B'] if force else []) + ['liboracle.so'])
    return so


class _Cfg(C.Structure):
    _fields_ = [('A', C.c_int32), ('S', C.c_int32), ('players', C.c_int32), ('gumbel', C.c_int32),
                ('discount', C.c_double), ('c_base', C.c_double), ('c_init', C.c_double),
                ('max_considered', C.c_int32), ('_pad', C.c_int32),
                ('c_visit', C.c_double), ('c_scale', C.c_double), ('c_table', C.c_void_p)]


class _Tree(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in (
        'node_visits', 'node_values', 'node_raw_values', 'parents', 'action_from_parent', 'to_play',
        'children_index', 'children_prior_logits', 'children_visits', 'children_rewards',
        'children_values', 'hidden_states', 'root_legal', 'gumbel_noise')] + [
        ('H', C.c_int32), ('_pad', C.c_int32)]


class _Hash(C.Structure):
    _fields_ = [('logit_shift', C.c_int32), ('reward_on', C.c_int32),
                ('parent_predicts', C.c_int32), ('_pad', C.c_int32)]


class _Net(C.Structure):
    _fields_ = [('mode', C.c_int32), ('_pad', C.c_int32), ('hash', _Hash),
                ('logits', C.c_void_p), ('value', C.c_void_p), ('reward', C.c_void_p),
                ('hidden', C.c_void_p)]


class _Noise(C.Structure):
    _fields_ = [('mode', C.c_int32), ('max_calls', C.c_int32), ('seed', C.c_uint64),
                ('tree_id', C.c_uint64), ('buf', C.c_void_p), ('calls', C.c_int64)]


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.orc_np_sum_f64.restype = C.c_double
        _LIB.orc_np_sum_f32.restype = C.c_float
        _LIB.orc_np_expf.restype = C.c_float
        _LIB.orc_np_expf.argtypes = [C.c_float]
        _LIB.orc_noise_u.restype = C.c_double
        _LIB.orc_noise_u.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32]
        _LIB.orc_hash_value.restype = C.c_double
        _LIB.orc_hash_reward.restype = C.c_double
        _LIB.orc_hash_child.restype = C.c_uint32
        _LIB.orc_puct_to_act.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.c_double,
                                         C.c_void_p, C.c_void_p]
    return _LIB


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def default_c_table(cfg) -> np.ndarray:
    """c(n) = c_init + log((n + c_base + 1) / c_base) evaluated by NumPy exactly as
    reference mcts/action_selection.py:43 does (np.int32 visit count)."""
    n = np.arange(cfg.simulations + 2, dtype=np.int32)
    return np.array([cfg.c_init + np.log((k + cfg.c_base + 1) / cfg.c_base) for k in n],
                    dtype=np.float64)


class Oracle:
    """One search tree held as the reference's struct-of-arrays (mcts/tree.py:26-40)."""

    def __init__(self, config, hidden_size: int = 0, c_table: np.ndarray | None = None):
        self.config = config
        self.A, self.S = int(config.n_actions), int(config.simulations)
        self.N = self.S + 1
        self.H = int(hidden_size)
        self.gumbel = bool(config.gumbel)
        self.c_table = None
        if not self.gumbel:
            self.c_table = np.ascontiguousarray(
                default_c_table(config) if c_table is None else c_table, dtype=np.float64)
        g = self.gumbel
        self.cfg = _Cfg(self.A, self.S, int(config.players), int(g), float(config.gamma),
                        0.0 if g else float(config.c_base), 0.0 if g else float(config.c_init),
                        int(config.max_considered_actions) if g else 0, 0,
                        float(config.c_visit) if g else 0.0, float(config.c_scale) if g else 0.0,
                        _p(self.c_table))
        N, A = self.N, self.A
        pd = np.float32 if g else np.float64
        self.arrays = dict(
            node_visits=np.zeros(N, np.int32), node_values=np.zeros(N, np.float64),
            node_raw_values=np.zeros(N, np.float64), parents=np.zeros(N, np.int32),
            action_from_parent=np.zeros(N, np.int32), to_play=np.zeros(N, np.int32),
            children_index=np.zeros((N, A), np.int32), children_prior_logits=np.zeros((N, A), pd),
            children_visits=np.zeros((N, A), np.int32), children_rewards=np.zeros((N, A), np.float64),
            children_values=np.zeros((N, A), np.float64),
            hidden_states=np.zeros((N, max(self.H, 1)), np.float32))
        self.root_legal = np.ones(A, np.float64)
        self.gumbel_noise = None
        self.tree = _Tree()
        self._bind()

    def _bind(self):
        for k, v in self.arrays.items():
            setattr(self.tree, k, _p(v))
        if self.H == 0:
            self.tree.hidden_states = None
        self.tree.root_legal = _p(self.root_legal)
        self.tree.gumbel_noise = _p(self.gumbel_noise)
        self.tree.H = self.H

    def __getattr__(self, name):
        arrays = self.__dict__.get('arrays', {})
        if name in arrays:
            return arrays[name]
        raise AttributeError(name)

    # -- building blocks --------------------------------------------------
    def init(self, root_prior, root_value, root_hidden, to_play, legal, gumbel_noise=None):
        self.root_legal = np.ascontiguousarray(legal, dtype=np.float64)
        self.gumbel_noise = None if gumbel_noise is None else np.ascontiguousarray(gumbel_noise, np.float64)
        self._bind()
        pd = np.float32 if self.gumbel else np.float64
        rp = np.ascontiguousarray(root_prior, dtype=pd)
        rh = None if root_hidden is None or self.H == 0 else np.ascontiguousarray(
            np.asarray(root_hidden, np.float32).reshape(-1))
        lib().orc_tree_init(C.byref(self.cfg), C.byref(self.tree), _p(rp), C.c_double(float(root_value)),
                            _p(rh), C.c_int(int(to_play)))

    def search(self, net, noise):
        sel = [np.zeros(self.S, np.int32) for _ in range(3)]
        lib().orc_search(C.byref(self.cfg), C.byref(self.tree), C.byref(net), C.byref(noise),
                         _p(sel[0]), _p(sel[1]), _p(sel[2]))
        return SimpleNamespace(parent=sel[0], action=sel[1], depth=sel[2], noise_calls=int(noise.calls))

    def hash_net(self, logit_shift=5, reward_on=1, parent_predicts=1):
        return _Net(0, 0, _Hash(logit_shift, reward_on, parent_predicts, 0), None, None, None, None)

    def table_net(self, logits, value, reward, hidden=None):
        self._keep = [np.ascontiguousarray(logits, np.float32), np.ascontiguousarray(value, np.float64),
                      np.ascontiguousarray(reward, np.float64),
                      None if hidden is None else np.ascontiguousarray(hidden, np.float32)]
        return _Net(1, 0, _Hash(0, 0, 0, 0), *[_p(a) for a in self._keep])

    def counter_noise(self, seed, tree_id):
        return _Noise(0, 0, seed, tree_id, None, 0)

    def buffer_noise(self, buf):
        self._nbuf = np.ascontiguousarray(buf, np.float64).reshape(-1, self.A)
        return _Noise(1, self._nbuf.shape[0], 0, 0, _p(self._nbuf), 0)

    def puct_select(self, node, depth, uniform):
        u = np.ascontiguousarray(uniform, np.float64)
        return int(lib().orc_puct_select(C.byref(self.cfg), C.byref(self.tree), C.c_int(node),
                                         C.c_int(depth), _p(u)))

    def gumbel_root_select(self, node=0):
        return int(lib().orc_gumbel_root_select(C.byref(self.cfg), C.byref(self.tree), C.c_int(node)))

    def gumbel_interior_select(self, node):
        return int(lib().orc_gumbel_interior_select(C.byref(self.cfg), C.byref(self.tree), C.c_int(node)))

    def qtransform_by_parent_and_siblings(self, node):
        out = np.zeros(self.A)
        lib().orc_qtransform_by_parent_and_siblings(C.byref(self.cfg), C.byref(self.tree), C.c_int(node), _p(out))
        return out

    def qtransform_completed_by_mix_value(self, node):
        out = np.zeros(self.A)
        lib().orc_qtransform_completed_by_mix_value(C.byref(self.cfg), C.byref(self.tree), C.c_int(node), _p(out))
        return out

    def to_act(self, temperature=0.0, u=0.0):
        probs, q = np.zeros(self.A), np.zeros(self.A)
        if self.gumbel:
            a = lib().orc_gumbel_to_act(C.byref(self.cfg), C.byref(self.tree), _p(probs), _p(q))
        else:
            a = lib().orc_puct_to_act(C.byref(self.cfg), C.byref(self.tree), float(temperature), float(u),
                                      _p(probs), _p(q))
        return int(a), probs, q


def puct_root_prior(logits, dirichlet, frac, legal):
    A = len(logits)
    out = np.zeros(A)
    lib().orc_puct_root_prior(C.c_int(A), _p(np.ascontiguousarray(logits, np.float32)),
                              _p(np.ascontiguousarray(dirichlet, np.float64)), C.c_double(frac),
                              _p(np.ascontiguousarray(legal, np.float64)), _p(out))
    return out


def gumbel_root_prior(logits, legal):
    A = len(logits)
    out = np.zeros(A, np.float32)
    lib().orc_gumbel_root_prior(C.c_int(A), _p(np.ascontiguousarray(logits, np.float32)),
                                _p(np.ascontiguousarray(legal, np.float64)), _p(out))
    return out


def considered_visits(m, S):
    out = np.zeros(S, np.int32)
    lib().orc_get_considered_visits(C.c_int(m), C.c_int(S), _p(out))
    return out


def np_expf(x):
    x = np.ascontiguousarray(x, np.float32)
    y = np.empty_like(x)
    lib().orc_np_expf_array(_p(x), _p(y), C.c_int64(x.size))
    return y


def search_batch_hash(config, hash_params, H, root_codes, legal, to_play, dirichlet=None, frac=0.0,
                      gumbel=None, seed=0, tree_id0=0, n_threads=1, c_table=None):
    """B independent hash-network searches on host threads (the CPU baseline)."""
    o = Oracle(config, 0, c_table)
    B, A = len(root_codes), o.A
    rc = np.ascontiguousarray(root_codes, np.uint32)
    lg = np.ascontiguousarray(legal, np.float64)
    tp = np.ascontiguousarray(to_play, np.int32)
    di = None if dirichlet is None else np.ascontiguousarray(dirichlet, np.float64)
    gu = None if gumbel is None else np.ascontiguousarray(gumbel, np.float64)
    rv, val = np.zeros((B, A), np.int32), np.zeros(B)
    nv, act = np.zeros(B, np.int64), np.zeros(B, np.int32)
    hn = _Hash(hash_params.logit_shift, hash_params.reward_on, hash_params.parent_predicts, 0)
    lib().orc_search_batch_hash(C.byref(o.cfg), C.byref(hn), C.c_int(H), C.c_int64(B), _p(rc), _p(lg),
                                _p(tp), _p(di), C.c_double(frac), _p(gu), C.c_uint64(seed),
                                C.c_uint64(tree_id0), _p(rv), _p(val), _p(nv), _p(act), C.c_int(n_threads))
    return SimpleNamespace(root_visits=rv, root_value=val, node_visits=nv, action=act)


def net_recurrent_batch(packed, hidden, actions):
    """Leaf evaluation of a packed network (muzero_amd.packing.PackedNetwork) on the host."""
    r = packed.recurrent
    hidden = np.ascontiguousarray(hidden, np.float32)
    B = hidden.shape[0]
    act = np.ascontiguousarray(actions, np.int32)
    io = packed.io()
    lg, nh = np.zeros((B, packed.A), np.float32), np.zeros((B, packed.H), np.float32)
    val, rew = np.zeros(B), np.zeros(B)
    lib().orc_net_recurrent_batch(_p(r.meta), _p(r.data), _p(io), C.c_int64(B), _p(hidden), _p(act), _p(lg),
                                  _p(nh), _p(val), _p(rew))
    return lg, nh, val, rew


def search_batch_net(config, packed, root_logits, root_hidden, root_value, legal, to_play, dirichlet=None,
                     frac=0.0, gumbel=None, seed=0, tree_id0=0, n_threads=1, c_table=None):
    """B independent searches with a packed network as leaf evaluator, on host threads."""
    o = Oracle(config, 0, c_table)
    rl = np.ascontiguousarray(root_logits, np.float32)
    B, A = rl.shape
    rh = np.ascontiguousarray(root_hidden, np.float32).reshape(B, -1)
    rv0 = np.ascontiguousarray(root_value, np.float64)
    lg = np.ascontiguousarray(legal, np.float64)
    tp = np.ascontiguousarray(to_play, np.int32)
    di = None if dirichlet is None else np.ascontiguousarray(dirichlet, np.float64)
    gu = None if gumbel is None else np.ascontiguousarray(gumbel, np.float64)
    rv, val = np.zeros((B, A), np.int32), np.zeros(B)
    nv, act = np.zeros(B, np.int64), np.zeros(B, np.int32)
    r, io = packed.recurrent, packed.io()
    lib().orc_search_batch_net(C.byref(o.cfg), _p(r.meta), _p(r.data), _p(io), C.c_int64(B), _p(rl), _p(rh),
                               _p(rv0), _p(lg), _p(tp), _p(di), C.c_double(frac), _p(gu), C.c_uint64(seed),
                               C.c_uint64(tree_id0), _p(rv), _p(val), _p(nv), _p(act), C.c_int(n_threads))
    return SimpleNamespace(root_visits=rv, root_value=val, node_visits=nv, action=act)


def make_target(rewards, to_plays, root_values, probs, t, td_steps, gamma, unroll_steps):
    """GameHistory.make_target on one game's lists -> (value[U+1], reward[U+1], policy[U+1][A]) float64."""
    rewards = np.ascontiguousarray(rewards, np.float64)
    tp = np.ascontiguousarray(to_plays, np.int32)
    rv = np.ascontiguousarray(root_values, np.float64)
    pr = np.ascontiguousarray(probs, np.float64)
    T, A = pr.shape
    gp = np.array([float(gamma) ** k for k in range(td_steps + 1)], np.float64)
    v, r, p = np.zeros(unroll_steps + 1), np.zeros(unroll_steps + 1), np.zeros((unroll_steps + 1, A))
    f = lib().orc_make_target
    f.restype = None
    f(C.c_int(T), C.c_int(A), _p(rewards), _p(tp), _p(rv), _p(pr), C.c_int(int(t)), C.c_int(int(td_steps)),
      C.c_double(float(gamma)), _p(gp), C.c_int(int(unroll_steps)), _p(v), _p(r), _p(p))
    return v, r, p


def stack_observations(observations, actions, t, n, n_actions, stack_action):
    """GameHistory.stack_n_observations on one non-empty game -> [n * (1 + stack_action), obs] float64."""
    obs = np.ascontiguousarray(observations, np.float64)
    T = obs.shape[0]
    flat = obs.reshape(T, -1)
    act = np.ascontiguousarray(actions, np.int32)
    out = np.zeros((n * (2 if stack_action else 1), flat.shape[1]))
    f = lib().orc_stack_observations
    f.restype = None
    f(C.c_int(T), C.c_int(flat.shape[1]), _p(flat), _p(act), C.c_int(int(t)), C.c_int(int(n)), C.c_int(int(n_actions)),
      C.c_int(int(bool(stack_action))), _p(out))
    return out


def stack_before_move(initial_observation, observations, actions, n, n_actions, stack_action):
    """What self-play hands to the search before a move (self_play.py:86-91): stack_n_observations(-1, ...) on the
    history so far.  An empty history gives the initial observation followed by zero planes (games/game.py:134-141)."""
    if len(observations) == 0:
        first = np.asarray(initial_observation, np.float64).reshape(-1)
        out = np.zeros((n * (2 if stack_action else 1), first.size))
        out[0] = first
        return out
    return stack_observations(observations, actions, -1, n, n_actions, stack_action)


def scalar_to_support(x, support_limit):
    """utils/network_utils.scalar_to_support on a float32 array -> (..., 2L+1) float32."""
    xf = np.ascontiguousarray(x, np.float32)
    out = np.zeros(xf.shape + (2 * int(support_limit) + 1,), np.float32)
    f = lib().orc_scalar_to_support
    f.restype = None
    f(_p(xf), C.c_int64(xf.size), C.c_int(int(support_limit)), _p(out))
    return out
