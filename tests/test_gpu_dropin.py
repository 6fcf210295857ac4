"""GPU: the drop-in `mcts` package (reference call signatures) against the reference's outputs."""
import os
import sys
from functools import partial

import numpy as np
import pytest
import torch

from helpers import load, hash_cases, assert_tree_equal, TREE_FIELDS, make_config
from muzero_amd import hashnet, noise as mznoise

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def mcts_pkg():
    """Import the package the way a user of the reference would: `from mcts.mcts import MCTS`."""
    sys.path.insert(0, os.path.join(REPO, 'muzero_amd'))
    import mcts.mcts as m
    import mcts.action_selection as asel
    import mcts.qtransforms as qt
    import mcts.tree as tr
    yield m, asel, qt, tr
    sys.path.remove(os.path.join(REPO, 'muzero_amd'))


def test_search_consumes_numpy_stream_like_reference(mcts_pkg):
    """Seed NumPy, call MCTS.search as self_play.py:92-99 does: same action, root value, policy,
    same tree, and the global stream is left exactly where the reference leaves it."""
    m = mcts_pkg[0]
    for case in hash_cases('search_stream_puct.npz', prefix='s'):
        hp = hashnet.HashNetParams(case.A, case.H, case.logit_shift, case.reward_on, case.parent_predicts)
        net = hashnet.HashNetwork(hp)
        np.random.seed(case.seed)
        search = m.MCTS(case.cfg)
        action, root_value, probs = search.search(net, hashnet.observation_of(case.root_code), case['legal'],
                                                  lambda a: np.eye(case.A)[a], case.to_play,
                                                  float(case['temperature']))
        assert np.random.random_sample() == float(case['next_random_sample'])
        assert int(action) == int(case['action'])
        assert float(root_value) == float(case['root_value'])
        np.testing.assert_allclose(probs, case['probs'], rtol=1e-12, atol=1e-300)
        tree = search.last_tree(0)
        assert_tree_equal(tree, case, fields=[f for f in TREE_FIELDS if f != 'children_prior_logits'])
        # prior rows: leaves are the network's logits (exact); the root row went through log() on
        # the device (<= 1 ulp from NumPy's)
        got, ref = tree.children_prior_logits, case['children_prior_logits']
        assert np.array_equal(got[1:], ref[1:])
        legal = case['legal'] != 0
        np.testing.assert_allclose(got[0][legal], ref[0][legal], rtol=0, atol=1e-15)
        assert np.array_equal(got[0][~legal], ref[0][~legal])


def test_gumbel_search_drop_in(mcts_pkg):
    m = mcts_pkg[0]
    for case in hash_cases('search_hash_gumbel.npz')[::4]:
        hp = hashnet.HashNetParams(case.A, case.H, case.logit_shift, case.reward_on, case.parent_predicts)
        net = hashnet.HashNetwork(hp)
        np.random.seed(case.seed)  # the fixture drew its Gumbel noise from this seed
        search = m.MCTS(case.cfg)
        action, root_value, probs = search.search(net, hashnet.observation_of(case.root_code), case['legal'],
                                                  lambda a: np.eye(case.A)[a], case.to_play)
        assert int(action) == int(case['action'])
        assert float(root_value) == float(case['root_value'])
        np.testing.assert_allclose(probs, case['probs'], rtol=1e-12, atol=1e-300)
        assert_tree_equal(search.last_tree(0), case)


def _step_by_step(m, asel, qt, case):
    """The reference's own search loop (mcts/mcts.py:181-252) spelled with the drop-in's functions."""
    cfg = case.cfg
    hp = hashnet.HashNetParams(case.A, case.H, case.logit_shift, case.reward_on, case.parent_predicts)
    net = hashnet.HashNetwork(hp)
    search = m.MCTS(cfg)
    logits, hidden, value = net.initial_inference(torch.as_tensor(hashnet.observation_of(case.root_code)[None]))
    logits, hidden, value = logits.numpy().squeeze(), hidden.numpy(), value.numpy().squeeze()
    if cfg.gumbel:
        qtf = partial(qt.qtransform_completed_by_mix_value, c_visit=cfg.c_visit, c_scale=cfg.c_scale)
        root_fn = partial(asel.gumbel_muzero_root_action_selection, discount=cfg.gamma, simulations=cfg.simulations,
                          max_considered_actions=cfg.max_considered_actions, gumbel_scale=cfg.gumbel_scale,
                          qtransform=qtf)
        inner_fn = partial(asel.gumbel_muzero_interior_action_selection, discount=cfg.gamma, qtransform=qtf)
        prior = case['children_prior_logits'][0]
        gnoise = case['gumbel_noise']
    else:
        root_fn = partial(asel.muzero_action_selection, discount=cfg.gamma, c_base=cfg.c_base, c_init=cfg.c_init,
                          qtransform=qt.qtransform_by_parent_and_siblings)
        inner_fn = root_fn
        prior = case['children_prior_logits'][0]
        gnoise = None
    fn = asel.switching_action_selection_wrapper(root_fn, inner_fn)
    tree = search.initialize_tree(prior, hidden, value, case.to_play, case['legal'], gnoise)
    for sim in range(cfg.simulations):
        node, action, ntp = search.simulate(tree, fn)
        lg, nh, val, rew = net.recurrent_inference(torch.as_tensor(tree.hidden_states[node]),
                                                   torch.as_tensor(np.eye(case.A)[action][None]))
        tree = search.expand(tree, node, action, sim + 1, lg.numpy(), nh.numpy(), rew.numpy(), val.numpy(), ntp)
        tree = search.backward(tree, sim + 1, ntp)
    return tree


def test_step_by_step_api_reproduces_reference_tree(mcts_pkg):
    m, asel, qt, tr = mcts_pkg
    real_uniform = np.random.uniform
    for name in ('search_hash_puct.npz', 'search_hash_gumbel.npz'):
        for case in [c for c in hash_cases(name) if c.S == 10][:3]:
            calls = {'n': 0}

            def counter_uniform(low=0.0, high=1.0, size=None):
                row = mznoise.noise_row(case.seed, case.idx, calls['n'], int(size))
                calls['n'] += 1
                return row
            np.random.uniform = counter_uniform
            try:
                tree = _step_by_step(m, asel, qt, case)
            finally:
                np.random.uniform = real_uniform
            assert_tree_equal(tree, case)
            counts, probs, q = tree.summary()
            assert np.array_equal(counts, case['children_visits'][0])


def test_module_level_helpers(mcts_pkg):
    m, asel, qt, tr = mcts_pkg
    z = load('misc.npz')
    assert asel.get_considered_visits(9, 25) == tuple(int(x) for x in z['cv_9_25'])
    assert tr.Tree.ROOT_INDEX == 0 and tr.Tree.NO_PARENT == -1 and tr.Tree.UNVISITED == -1 and tr.Tree.P1 == 0
    x = np.array([0.1, 0.9, 0.5])
    assert asel.masked_argmax(x, np.array([1.0, 0.0, 1.0])) == 2
    assert asel.masked_argmax(x, None) == 1
    case = [c for c in hash_cases('search_hash_gumbel.npz') if c.S == 10][0]
    from oracle import oracle as orc
    cv = case['children_visits'][3]
    qv = case['children_rewards'][3] + case.gamma * case['children_values'][3]
    got = qt.compute_mixed_value(case['node_raw_values'][3], qv, cv, case['children_prior_logits'][3])
    import ctypes as C
    ref = np.zeros(case.A)
    p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    orc.lib().orc_compute_mixed_value(C.c_int(case.A), C.c_double(float(case['node_raw_values'][3])),
                                      p(np.ascontiguousarray(qv)), p(np.ascontiguousarray(cv)),
                                      p(np.ascontiguousarray(case['children_prior_logits'][3])), p(ref))
    assert np.array_equal(got, ref)


def test_dense_network_batched_search_on_gpu(mcts_pkg):
    """Real (lowered) networks evaluated on the GPU inside search_batch: structural invariants,
    and agreement with the reference's recorded search up to fp32 network rounding."""
    m = mcts_pkg[0]
    from test_network_lowering import load_net
    for name, obs_shape in (('mlp_cartpole', (1, 1, 4)), ('resnet_tictactoe', (3, 3, 3))):
        z, net = load_net(name)
        net.to('cuda')
        p = 'r1_'  # a Gumbel case: no tie-break noise involved
        gumbel, A, S, players, H, to_play, seed, idx, calls = [int(x) for x in z[p + 'meta']]
        cfg = make_config(gumbel, A, S, players, float(z[p + 'gamma']))
        B = 8
        np.random.seed(seed)  # first gumbel draw == the fixture's
        search = m.MCTS(cfg)
        obs = np.repeat(z[p + 'obs'][None], B, 0)
        actions, values, probs = search.search_batch(net, obs, np.repeat(z[p + 'legal'][None], B, 0),
                                                     np.full(B, to_play))
        tree = search.last_tree(0)
        assert tree.node_visits[0] == S + 1 and tree.children_visits[0].sum() == S
        assert np.array_equal(tree.children_visits[0], z[p + 'children_visits'][0])
        assert int(actions[0]) == int(z[p + 'action'])
        assert abs(float(values[0]) - float(z[p + 'search_value'])) < 1e-5
        np.testing.assert_allclose(probs[0], z[p + 'probs'], atol=1e-4)
        np.testing.assert_allclose(tree.node_values, z[p + 'node_values'], atol=1e-5)


def test_downsampled_network_searches_on_the_matrix_cores(mcts_pkg):
    """`downsample` on (network.py:228-325): roots from the torch-evaluated representation, every simulation's leaf on
    the matrix cores, against the same search with torch evaluating every leaf (the generic path, taken for a network
    object that only has the reference's two inference calls).  Same Gumbel noise for both; the leaf evaluators agree
    to float32 rounding, so nearly all trees are identical and the rest differ by a late visit."""
    m = mcts_pkg[0]
    from test_network_lowering import load_net

    class TwoCalls:   # what a reference network offers, nothing the lowering recognises
        def __init__(self, net):
            self.net, self.device = net, net.device

        def initial_inference(self, obs):
            return self.net.initial_inference(obs)

        def recurrent_inference(self, h, a):
            return self.net.recurrent_inference(h, a)

    z, net = load_net('resnet_downsample')
    net.to('cuda')
    A, S, B = 9, 30, 64
    cfg = make_config(True, A, S, 1, 0.997)
    rng = np.random.RandomState(3)
    obs = rng.uniform(0.0, 1.0, size=(B, 3, 48, 48)).astype(np.float32)
    legal, tp = np.ones((B, A)), np.zeros(B, np.int32)
    out = []
    for network in (net, TwoCalls(net)):
        np.random.seed(17)
        search = m.MCTS(cfg)
        actions, values, probs = search.search_batch(network, obs, legal, tp, action_encoder=lambda a: np.eye(9)[a].reshape(3, 3))
        trees = [search.last_tree(b) for b in range(B)]
        out.append((np.asarray(actions), np.asarray(values), np.asarray(probs), np.stack([t.children_visits[0] for t in trees])))
    assert m.MCTS._lowered_plan(net, A)[0] == 'mfma' and m.MCTS._lowered_plan(net, A)[2] is None
    same = (out[0][3] == out[1][3]).all(axis=1)
    assert same.mean() >= 0.9, same.mean()
    assert np.array_equal(out[0][0][same], out[1][0][same])
    np.testing.assert_allclose(out[0][1][same], out[1][1][same], atol=1e-5, rtol=0)
    assert (out[0][3].sum(axis=1) == S).all()


def test_reference_style_module_is_lowered_once_per_weight_version(mcts_pkg):
    """An eval-mode nn.Module with the reference's state_dict layout takes the single-launch path: it is
    lowered on first use, reused while its weights are untouched, lowered again after an in-place update."""
    import torch
    m = mcts_pkg[0]
    from test_network_lowering import load_net, net_config

    class Module(torch.nn.Module):          # only what MCTS looks at: parameters in the reference's layout
        def __init__(self, sd):
            super().__init__()
            self.keys = list(sd)
            self.store = torch.nn.ParameterList([torch.nn.Parameter(v.clone().float(), requires_grad=False)
                                                 for v in sd.values()])

        def state_dict(self, *a, **k):
            return {key: p.data for key, p in zip(self.keys, self.store)}

    z, net = load_net('mlp_cartpole')
    ncfg = net_config('mlp_cartpole')
    cfg = make_config(True, 2, 30, 1, 0.997)
    for k, v in vars(ncfg).items():
        setattr(cfg, k, v)
    mod = Module(net.state_dict()).eval()
    search = m.MCTS(cfg)
    obs = np.random.default_rng(0).uniform(-0.05, 0.05, size=(4, 1, 1, 4))
    np.random.seed(3)
    a1, v1, p1 = search.search_batch(mod, obs, np.ones((4, 2)), np.zeros(4, np.int32))
    low1 = search._lowered[id(mod)][1]
    assert low1 is not None and hasattr(low1, 'dyn')
    np.random.seed(3)
    a2, v2, p2 = search.search_batch(net, obs, np.ones((4, 2)), np.zeros(4, np.int32))   # the lowered network itself
    np.testing.assert_array_equal(v1, v2)
    np.testing.assert_array_equal(p1, p2)
    np.random.seed(3)
    search.search_batch(mod, obs, np.ones((4, 2)), np.zeros(4, np.int32))
    assert search._lowered[id(mod)][1] is low1                      # same weights: no new lowering
    with torch.no_grad():
        mod.store[0].mul_(1.5)
    np.random.seed(3)
    a3, v3, p3 = search.search_batch(mod, obs, np.ones((4, 2)), np.zeros(4, np.int32))
    assert search._lowered[id(mod)][1] is not low1
    assert not np.array_equal(v3, v1)


@pytest.mark.parametrize('name,obs_shape', [('mlp_cartpole', (1, 1, 4)), ('resnet_tictactoe', (3, 3, 3))])
def test_single_search_with_stream_noise_same_on_every_path(mcts_pkg, monkeypatch, name, obs_shape):
    """MCTS.search (PUCT, tie-break rows taken from the global NumPy stream) with a lowered network: the
    single-launch paths and the select / torch network / expand loop consume the same draws and agree."""
    m = mcts_pkg[0]
    from test_network_lowering import load_net
    z, net = load_net(name)
    net.to('cuda')
    A = net.A
    cfg = make_config(False, A, 20, 2 if A == 9 else 1, 1 if A == 9 else 0.997)
    obs = np.asarray(z['obs'][0], dtype=np.float64).reshape(obs_shape)
    legal = np.ones(A)
    enc = lambda a: np.eye(A)[a]  # noqa: E731
    agree = 0
    for seed in range(6):
        out = []
        for one_launch in (True, False):
            search = m.MCTS(cfg)
            if not one_launch:
                monkeypatch.setattr(m.MCTS, '_lowered_plan', staticmethod(lambda network, A: None))
            np.random.seed(seed)
            a, v, p = search.search(net, obs, legal, enc, 0, 1.0)
            out.append((int(a), float(v), np.asarray(p), np.random.random_sample()))
            monkeypatch.undo()
        assert abs(out[0][1] - out[1][1]) < 1e-5
        np.testing.assert_allclose(out[0][2], out[1][2], atol=1e-6)
        agree += out[0][0] == out[1][0] and out[0][3] == out[1][3]   # same action, stream left at the same place
    assert agree >= 5


@pytest.mark.parametrize('gumbel', [True, False])
def test_search_batch_leaves_the_numpy_stream_where_consecutive_searches_would(mcts_pkg, gumbel):
    """search_batch draws the root noise of all its trees in one call; the global stream must end where B consecutive
    reference searches (mcts/mcts.py:211 / :27, mcts/action_selection.py:69; tie-break uniforms aside, which the
    batched search takes from its counter function) leave it, and tree b must have been given the b-th draw."""
    m = mcts_pkg[0]
    A, S, B, H = 9, 12, 37, 8
    cfg = make_config(gumbel, A, S, 2, 1.0)
    hp = hashnet.HashNetParams(A, H, 4, 1, 1)
    net = hashnet.HashNetwork(hp)
    obs = np.stack([hashnet.observation_of(1000 + b) for b in range(B)])
    legal, tp = np.ones((B, A)), np.zeros(B, np.int32)
    np.random.seed(99)
    if gumbel:
        want_noise = np.stack([np.random.gumbel(size=A) for _ in range(B)])
    else:
        for _ in range(B):
            np.random.dirichlet([cfg.dirichlet_alpha] * A)
            np.random.random_sample()
    want_next = np.random.random_sample(4)
    np.random.seed(99)
    search = m.MCTS(cfg)
    search.search_batch(net, obs, legal, tp, 1.0, action_encoder=lambda a: np.eye(A)[a])
    np.testing.assert_array_equal(np.random.random_sample(4), want_next)
    if gumbel:
        for b in (0, B // 2, B - 1):
            np.testing.assert_array_equal(search.last_tree(b).gumbel_noise, want_noise[b])
