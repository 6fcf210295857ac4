"""GPU parity: the HIP engine against golden vectors of the reference and against the CPU checker.

Bit-exact on every tree field, visit count, action and root value; the returned policy is compared
at rtol 1e-12 because it ends in a float64 exp whose last bit differs between math libraries.
"""
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from helpers import load, hash_cases, assert_tree_equal, make_config
from muzero_amd import hashnet

pytestmark = pytest.mark.gpu


def _engine(case_cfg, H, B, **kw):
    from muzero_amd.engine import SearchEngine
    return SearchEngine(case_cfg, H, B, **kw)


def _as_ns(arrs):
    return SimpleNamespace(**arrs)


def _lanes_options(A):
    l0 = 2
    while l0 < A:
        l0 *= 2
    return sorted({l0, max(l0, 16), 64})


@pytest.mark.parametrize('name', ['search_hash_puct.npz', 'search_hash_gumbel.npz'])
def test_fused_hash_search_vs_reference_golden(name):
    from muzero_amd.engine import NOISE_COUNTER
    for case in hash_cases(name):
        hp = hashnet.HashNetParams(case.A, case.H, case.logit_shift, case.reward_on, case.parent_predicts)
        for lanes in _lanes_options(case.A):
            eng = _engine(case.cfg, case.H, 4, noise_mode=NOISE_COUNTER, seed=case.seed, lanes_per_tree=lanes,
                          c_table=None if case.gumbel else case['c_table'])
            B = 3  # the same root three times at different batch slots; slot 1 carries the fixture's tree id
            prior = np.repeat(case['children_prior_logits'][0][None], B, 0)
            eng.reset(prior, np.full(B, hashnet.value_of(case.root_code)),
                      np.repeat(hashnet.hidden_of(case.root_code, hp)[None], B, 0),
                      np.full(B, case.to_play), np.repeat(case['legal'][None], B, 0),
                      np.repeat(case['gumbel_noise'][None], B, 0) if case.gumbel else None,
                      tree_id0=case.idx - 1 if case.idx > 0 else 0)
            slot = 1 if case.idx > 0 else 0
            eng.search_hash(case.logit_shift, case.reward_on, case.parent_predicts)
            tree = _as_ns(eng.export_tree(slot))
            assert_tree_equal(tree, case)
            action, value, probs, q = eng.to_act(float(case['temperature']), np.full(B, float(case['choice_u'])))
            assert int(action[slot]) == int(case['action']), (case.idx, lanes)
            assert float(value[slot]) == float(case['root_value'])
            np.testing.assert_allclose(probs[slot].cpu().numpy(), case['probs'], rtol=1e-12, atol=1e-300)
            nv, nc = eng.stats()
            assert int(nv[slot]) == int(tree.node_visits.sum()) - 1
            if not case.gumbel:
                assert int(nc[slot]) == case.calls
            eng.close()


def test_noise_buffer_mode_vs_reference_stream():
    """PUCT with the uniforms the reference drew from the global np.random stream."""
    from muzero_amd.engine import NOISE_BUFFER
    for case in hash_cases('search_stream_puct.npz', prefix='s'):
        hp = hashnet.HashNetParams(case.A, case.H, case.logit_shift, case.reward_on, case.parent_predicts)
        eng = _engine(case.cfg, case.H, 1, noise_mode=NOISE_BUFFER)
        eng.reset(case['children_prior_logits'][0][None], [hashnet.value_of(case.root_code)],
                  hashnet.hidden_of(case.root_code, hp)[None], [case.to_play], case['legal'][None])
        rows = case['uniform_rows']
        eng.search_hash(case.logit_shift, case.reward_on, case.parent_predicts, noise=rows[None])
        tree = _as_ns(eng.export_tree(0))
        assert_tree_equal(tree, case)
        _, nc = eng.stats()
        assert int(nc[0]) == len(rows)
        eng.close()


@pytest.mark.parametrize('fname', ['net_mlp_cartpole.npz', 'net_resnet_tictactoe.npz', 'net_mlp_synthetic.npz'])
def test_step_mode_with_recorded_network_outputs(fname):
    """select / expand+backup one simulation at a time, leaves fed with what the reference's real
    network returned; the (parent, action) chosen at every simulation must match too."""
    z = load(fname)
    for ci in range(int(z['n_search_cases'])):
        p = f'r{ci}_'
        gumbel, A, S, players, H, to_play, seed, idx, calls = [int(x) for x in z[p + 'meta']]
        cfg = make_config(gumbel, A, S, players, float(z[p + 'gamma']), alpha=float(z[p + 'alpha']))
        eng = _engine(cfg, H, 2, seed=seed, c_table=None if gumbel else z[p + 'c_table'])
        B = 2
        rep = lambda x: np.repeat(np.asarray(x)[None], B, 0)  # noqa: E731
        eng.reset(rep(z[p + 'children_prior_logits'][0]), rep(z[p + 'root_value']), rep(z[p + 'root_hidden'].reshape(-1)),
                  np.full(B, to_play), rep(z[p + 'legal']), rep(z[p + 'gumbel_noise']) if gumbel else None,
                  tree_id0=idx)
        ref_parent, ref_action = z[p + 'parents'], z[p + 'action_from_parent']
        for sim in range(S):
            parent, action, ph = eng.select()
            assert int(parent[0]) == int(ref_parent[sim + 1]) and int(action[0]) == int(ref_action[sim + 1]), (ci, sim)
            assert np.array_equal(ph[0].cpu().numpy(), z[p + 'hidden_states'][int(parent[0])].reshape(-1))
            eng.expand_backup(rep(z[p + 'rec_logits'][sim]), rep(z[p + 'rec_value'][sim]), rep(z[p + 'rec_reward'][sim]),
                              rep(z[p + 'rec_hidden'][sim].reshape(-1)))
        tree = _as_ns(eng.export_tree(0))

        class Case:
            def __getitem__(self, k):
                return z[p + k]
        assert_tree_equal(tree, Case())
        action, value, probs, _ = eng.to_act(1.0, np.full(B, float(z[p + 'choice_u'])))
        assert int(action[0]) == int(z[p + 'action'])
        assert float(value[0]) == float(z[p + 'search_value'])
        np.testing.assert_allclose(probs[0].cpu().numpy(), z[p + 'probs'], rtol=1e-12, atol=1e-300)
        eng.close()


def _batch_inputs(A, B, rng, partial=True):
    legal = np.ones((B, A))
    if partial and A > 2:
        for b in range(0, B, 3):
            k = rng.integers(1, A - 1)
            legal[b, rng.choice(A, size=k, replace=False)] = 0
    codes = rng.integers(0, 2**32, size=B, dtype=np.uint64).astype(np.uint32)
    return codes, legal


@pytest.mark.parametrize('gumbel,A,S,players,gamma,H,B', [
    (False, 2, 50, 1, 0.997, 8, 2048),     # CartPole-shaped PUCT
    (True, 9, 100, 2, 1, 16, 1024),        # TicTacToe-shaped Gumbel
    (False, 9, 50, 2, 1, 16, 1024),
    (True, 18, 200, 1, 0.997, 8, 256),     # Atari-shaped action count, deep schedule
    (False, 18, 200, 1, 0.997, 8, 256),
    (True, 2, 50, 1, 0.997, 8, 512),
    (True, 64, 40, 1, 0.997, 4, 64),       # maximum action count: one whole wavefront per tree
    (False, 33, 30, 2, 1, 4, 64),
])
def test_batched_search_vs_checker_at_scale(gumbel, A, S, players, gamma, H, B):
    """Thousands of independent trees: root visit counts, actions and root values bit-exact vs the
    CPU checker, and batch slot / tree id independence of the result."""
    from oracle import oracle as orc
    rng = np.random.default_rng(7 + A + S)
    cfg = make_config(gumbel, A, S, players, gamma)
    hp = hashnet.HashNetParams(A, H, 4, 1, 1)
    codes, legal = _batch_inputs(A, B, rng)
    to_play = rng.integers(0, players, size=B).astype(np.int32)
    gnoise = rng.gumbel(size=(B, A)) if gumbel else None
    dirichlet = None if gumbel else rng.dirichlet([0.25] * A, size=B)
    ref = orc.search_batch_hash(cfg, hp, H, codes, legal, to_play, dirichlet=dirichlet, frac=0.25, gumbel=gnoise,
                                seed=11, tree_id0=1000, n_threads=8)
    eng = _engine(cfg, H, B, seed=11)
    logits = np.stack([hashnet.logits_of(int(c), hp) for c in codes])
    prior = eng.root_prepare(logits, legal, dirichlet, 0.25)
    eng.reset(prior, [hashnet.value_of(int(c)) for c in codes], np.stack([hashnet.hidden_of(int(c), hp) for c in codes]),
              to_play, legal, gnoise, tree_id0=1000)
    eng.search_hash(4, 1, 1)
    action, value, probs, _ = eng.to_act(0.0)
    nv, _ = eng.stats()
    assert np.array_equal(nv.cpu().numpy(), ref.node_visits)
    assert np.array_equal(value.cpu().numpy(), ref.root_value)
    assert np.array_equal(action.cpu().numpy(), ref.action)
    for b in rng.choice(B, size=min(B, 16), replace=False):
        t = eng.export_tree(int(b))
        assert np.array_equal(t['children_visits'][0], ref.root_visits[b]), b
    eng.close()


@pytest.mark.parametrize('gumbel,A,S,players,gamma,H,B', [
    (False, 3, 1, 1, 0.9, 2, 1),       # a single simulation of a single tree
    (True, 5, 7, 2, 1, 3, 3),
    (False, 5, 12, 2, 1, 2, 5),
    (True, 3, 33, 1, 0.99, 2, 7),
    (False, 1, 6, 1, 0.99, 2, 4),      # one action: every tree is a chain
    (True, 1, 6, 2, 1, 2, 4),
    (False, 17, 9, 2, 1, 2, 65),       # ragged: 65 trees do not fill the last block
])
def test_edge_shapes_every_group_width(gumbel, A, S, players, gamma, H, B):
    """Odd action counts, single trees / simulations and ragged batches, for every lane-group width the
    library offers for the action count; whole trees compared with the CPU checker."""
    from oracle import oracle as orc
    rng = np.random.default_rng(100 + A * S)
    cfg = make_config(gumbel, A, S, players, gamma)
    hp = hashnet.HashNetParams(A, H, 4, 1, 1)
    codes, legal = _batch_inputs(A, B, rng)
    to_play = rng.integers(0, players, size=B).astype(np.int32)
    gnoise = rng.gumbel(size=(B, A)) if gumbel else None
    dirichlet = None if gumbel else rng.dirichlet([0.25] * A, size=B)
    ref = orc.search_batch_hash(cfg, hp, H, codes, legal, to_play, dirichlet=dirichlet, frac=0.25, gumbel=gnoise,
                                seed=3, tree_id0=5, n_threads=2)
    rl = 2
    while rl < A:
        rl *= 2
    for lanes in sorted({rl, max(rl, 16), 64}):
        eng = _engine(cfg, H, B, seed=3, lanes_per_tree=lanes)
        logits = np.stack([hashnet.logits_of(int(c), hp) for c in codes])
        prior = eng.root_prepare(logits, legal, dirichlet, 0.25)
        eng.reset(prior, [hashnet.value_of(int(c)) for c in codes],
                  np.stack([hashnet.hidden_of(int(c), hp) for c in codes]), to_play, legal, gnoise, tree_id0=5)
        eng.search_hash(4, 1, 1)
        action, value, probs, _ = eng.to_act(0.0)
        nv, _ = eng.stats()
        assert np.array_equal(nv.cpu().numpy(), ref.node_visits), lanes
        assert np.array_equal(value.cpu().numpy(), ref.root_value), lanes
        assert np.array_equal(action.cpu().numpy(), ref.action), lanes
        for b in range(B):
            assert np.array_equal(eng.export_tree(b)['children_visits'][0], ref.root_visits[b]), (lanes, b)
        eng.close()


def test_root_prepare_vs_reference():
    for name in ('search_hash_puct.npz', 'search_hash_gumbel.npz'):
        for case in hash_cases(name)[::3]:
            eng = _engine(case.cfg, case.H, 2)
            out = eng.root_prepare(case['root_logits'][None], case['legal'][None],
                                   None if case.gumbel else case['dirichlet'][None], 0.25)[0].cpu().numpy()
            ref = case['children_prior_logits'][0]
            assert out.dtype == ref.dtype
            if case.gumbel:
                assert np.array_equal(out, ref)
            else:
                legal = case['legal'] != 0
                assert np.array_equal(out[~legal], ref[~legal])
                np.testing.assert_allclose(out[legal], ref[legal], rtol=0, atol=4e-16 * max(1.0, np.abs(ref[legal]).max()))
            eng.close()


@pytest.mark.parametrize('name', ['search_hash_puct.npz', 'search_hash_gumbel.npz'])
def test_per_node_entry_points_vs_reference(name):
    for case in hash_cases(name):
        if not case.has('node_qtransform') or case.S > 10:
            continue
        eng = _engine(case.cfg, case.H, 1, noise_mode=2, c_table=None if case.gumbel else case['c_table'])
        arrays = {k: case[k] for k in ['node_visits', 'node_values', 'node_raw_values', 'parents', 'action_from_parent',
                                      'to_play', 'children_index', 'children_prior_logits', 'children_visits',
                                      'children_rewards', 'children_values']}
        arrays['hidden_states'] = case['hidden_states'].reshape(case.S + 1, -1)
        eng.import_tree(0, arrays, case['legal'], case['gumbel_noise'] if case.gumbel else None)
        back = eng.export_tree(0)
        for k in arrays:
            assert np.array_equal(back[k].reshape(arrays[k].shape), arrays[k]), k
        for n in range(case.S + 1):
            qt = eng.node_qtransform(0, n, 1 if case.gumbel else 0)
            assert np.array_equal(qt, case['node_qtransform'][n]), (case.idx, n)
            sel = eng.node_select(0, n, 1, None if case.gumbel else np.zeros(case.A))
            assert sel == int(case['node_select'][n]), (case.idx, n)
        eng.close()


def test_error_behaviour():
    from muzero_amd.engine import EngineError
    cfg = make_config(False, 4, 8, 1, 0.99)
    with pytest.raises(ValueError):
        _engine(cfg, 4, 4, lanes_per_tree=3)
    eng = _engine(cfg, 4, 4)
    with pytest.raises(EngineError):
        eng.select()  # no reset yet
    eng.reset(np.zeros((2, 4)), np.zeros(2), np.zeros((2, 4), np.float32), np.zeros(2, np.int32), np.ones((2, 4)))
    with pytest.raises(EngineError):
        eng.expand_backup(np.zeros((2, 4), np.float32), np.zeros(2), np.zeros(2))  # before select
    with pytest.raises(ValueError):
        eng.reset(np.zeros((5, 4)), np.zeros(5), None, np.zeros(5, np.int32), np.ones((5, 4)))  # B > max_batch
    eng.close()


def test_result_is_independent_of_the_shard_layout():
    """One context over all trees vs. the same trees split over 1, 2, 3 and 8 'ranks' (block partition of
    muzero_amd.distributed.shard_range, tree_id0 = first global tree of the shard): the noise is keyed by the
    global tree id, so every shard layout returns the same actions, values and visit counts."""
    from muzero_amd.distributed import shard_range
    rng = np.random.default_rng(21)
    A, S, H, B = 2, 30, 4, 203
    cfg = make_config(False, A, S, 1, 0.997)
    hp = hashnet.HashNetParams(A, H, 4, 1, 1)
    codes, legal = _batch_inputs(A, B, rng, partial=False)
    dirichlet = rng.dirichlet([0.25] * A, size=B)
    logits = np.stack([hashnet.logits_of(int(c), hp) for c in codes])
    values = np.array([hashnet.value_of(int(c)) for c in codes])
    hidden = np.stack([hashnet.hidden_of(int(c), hp) for c in codes])
    to_play = np.zeros(B, np.int32)

    def run(lo, hi):
        eng = _engine(cfg, H, hi - lo, seed=17)
        prior = eng.root_prepare(logits[lo:hi], legal[lo:hi], dirichlet[lo:hi], 0.25)
        eng.reset(prior, values[lo:hi], hidden[lo:hi], to_play[lo:hi], legal[lo:hi], None, tree_id0=lo)
        eng.search_hash(4, 1, 1)
        action, value, probs, _ = eng.to_act(0.0)
        nv, calls = eng.stats()
        out = [x.cpu().numpy().copy() for x in (action, value, probs, nv, calls)]
        eng.close()
        return out

    whole = run(0, B)
    for world in (2, 3, 8):
        parts = [run(*shard_range(B, r, world)) for r in range(world)]
        for k in range(5):
            np.testing.assert_array_equal(np.concatenate([p[k] for p in parts]), whole[k])


def test_random_configurations_vs_checker():
    """Forty random search configurations (rule, action count, simulations, players, discount, legality,
    group width, hash-network shape): whole batches against the CPU checker, bit for bit."""
    from oracle import oracle as orc
    rng = np.random.default_rng(2024)
    for trial in range(40):
        gumbel = bool(rng.integers(2))
        A = int(rng.choice([2, 3, 4, 6, 7, 9, 12, 16, 18, 25, 40, 64]))
        S = int(rng.integers(1, 60))
        players = int(rng.integers(1, 3))
        gamma = float(rng.choice([1.0, 0.997, 0.9]))
        H = int(rng.integers(2, 9))
        B = int(rng.integers(1, 70))
        cfg = make_config(gumbel, A, S, players, gamma)
        if gumbel:
            cfg.max_considered_actions = int(rng.choice([2, 4, 16]))
        hp = hashnet.HashNetParams(A, H, int(rng.integers(2, 7)), int(rng.integers(2)), int(rng.integers(2)))
        codes, legal = _batch_inputs(A, B, rng, partial=bool(rng.integers(2)))
        to_play = rng.integers(0, players, size=B).astype(np.int32)
        gnoise = rng.gumbel(size=(B, A)) if gumbel else None
        dirichlet = None if gumbel else rng.dirichlet([0.25] * A, size=B)
        ref = orc.search_batch_hash(cfg, hp, H, codes, legal, to_play, dirichlet=dirichlet, frac=0.25, gumbel=gnoise,
                                    seed=trial, tree_id0=trial * 7, n_threads=4)
        rl = 2
        while rl < A:
            rl *= 2
        lanes = int(rng.choice(sorted({rl, max(rl, 16), 64})))
        eng = _engine(cfg, H, B, seed=trial, lanes_per_tree=lanes)
        logits = np.stack([hashnet.logits_of(int(c), hp) for c in codes])
        prior = eng.root_prepare(logits, legal, dirichlet, 0.25)
        eng.reset(prior, [hashnet.value_of(int(c)) for c in codes],
                  np.stack([hashnet.hidden_of(int(c), hp) for c in codes]), to_play, legal, gnoise, tree_id0=trial * 7)
        eng.search_hash(hp.logit_shift, hp.reward_on, hp.parent_predicts)
        action, value, probs, _ = eng.to_act(0.0)
        nv, _ = eng.stats()
        tag = (trial, gumbel, A, S, players, gamma, lanes)
        assert np.array_equal(nv.cpu().numpy(), ref.node_visits), tag
        assert np.array_equal(value.cpu().numpy(), ref.root_value), tag
        assert np.array_equal(action.cpu().numpy(), ref.action), tag
        for b in range(0, B, 9):
            assert np.array_equal(eng.export_tree(b)['children_visits'][0], ref.root_visits[b]), tag
        eng.close()


def test_nan_regime_matches_reference():
    """Roots whose float32 prior softmax underflows entirely (the reference's scores are all NaN there and it
    plays the first legal action): actions, root values, root visit counts and the NaN policies equal the
    reference's, for every group width."""
    z = load('nan_regime.npz')
    A, S, H, B, m = (int(x) for x in z['meta'])
    cfg = make_config(True, A, S, 1, 0.5)
    cfg.max_considered_actions = m
    hp = hashnet.HashNetParams(A, H, 1, 0, 0)
    codes, legal, g = z['codes'], z['legal'], z['gumbel']
    for lanes in (32, 64):
        eng = _engine(cfg, H, B, seed=0, lanes_per_tree=lanes)
        logits = np.stack([hashnet.logits_of(int(c), hp) for c in codes])
        prior = eng.root_prepare(logits, legal)
        eng.reset(prior, [hashnet.value_of(int(c)) for c in codes],
                  np.stack([hashnet.hidden_of(int(c), hp) for c in codes]), np.zeros(B, np.int32), legal, g)
        eng.search_hash(1, 0, 0)
        action, value, probs, _ = eng.to_act(0.0)
        np.testing.assert_array_equal(action.cpu().numpy(), z['action'])
        np.testing.assert_array_equal(value.cpu().numpy(), z['root_value'])
        p = probs.cpu().numpy()
        assert np.array_equal(np.isnan(p), np.isnan(z['probs']))
        np.testing.assert_allclose(p[~np.isnan(p)], z['probs'][~np.isnan(z['probs'])], rtol=1e-12, atol=1e-300)
        for b in range(B):
            assert np.array_equal(eng.export_tree(b)['children_visits'][0], z['root_visits'][b]), (lanes, b)
        eng.close()


def test_noise_key_follows_the_seed_of_each_search():
    """One engine, two searches with different noise keys (mz_set_seed): each equals the checker run with its
    own key, and the keys give different tie-break draws.  MCTS.search_batch re-keys its cached engine."""
    from oracle import oracle as orc
    from muzero_amd import noise as mznoise
    rng = np.random.default_rng(5)
    A, S, H, B = 2, 40, 4, 256      # two actions with near-equal priors: ties are frequent
    cfg = make_config(False, A, S, 1, 0.997)
    hp = hashnet.HashNetParams(A, H, 30, 0, 1)   # logits of order 1e-7: the 1e-7 * u tie-break decides often
    codes, legal = _batch_inputs(A, B, rng)
    to_play = np.zeros(B, np.int32)
    dirichlet = np.full((B, A), 0.5)
    logits = np.stack([hashnet.logits_of(int(c), hp) for c in codes])
    eng = _engine(cfg, H, B, seed=3)
    seen = {}
    for seed in (3, 11):
        ref = orc.search_batch_hash(cfg, hp, H, codes, legal, to_play, dirichlet=dirichlet, frac=0.25, seed=seed,
                                    tree_id0=7, n_threads=4)
        if seed != 3:
            eng.set_seed(seed)
        prior = eng.root_prepare(logits, legal, dirichlet, 0.25)
        eng.reset(prior, [hashnet.value_of(int(c)) for c in codes], np.stack([hashnet.hidden_of(int(c), hp) for c in codes]),
                  to_play, legal, None, tree_id0=7)
        eng.search_hash(30, 0, 1)
        nv, _ = eng.stats()
        assert np.array_equal(nv.cpu().numpy(), ref.node_visits), seed
        seen[seed] = np.stack([eng.export_tree(b)['children_visits'][0] for b in range(32)])
        assert np.array_equal(seen[seed], ref.root_visits[:32]), seed
    assert not np.array_equal(seen[3], seen[11])
    eng.close()
    assert mznoise.noise_u(3, 7, 0, 0) != mznoise.noise_u(11, 7, 0, 0)
    # the drop-in's engine cache: same engine object, new key
    sys_path_mcts = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'muzero_amd')
    import sys
    sys.path.insert(0, sys_path_mcts)
    from mcts.mcts import MCTS
    from muzero_amd.engine import NOISE_COUNTER
    m = MCTS(cfg)
    e1 = m._engine_for(H, 4, NOISE_COUNTER, seed=1)
    e2 = m._engine_for(H, 4, NOISE_COUNTER, seed=2)
    assert e1 is e2 and e2.seed == 2
    e1.close()


def test_short_noise_buffer_is_reported():
    """A PUCT search that makes more selection calls than the caller's noise buffer has rows raises at the next
    check instead of silently scoring with zero noise."""
    from muzero_amd.engine import NOISE_BUFFER, EngineError
    A, S, H, B = 3, 12, 2, 4
    cfg = make_config(False, A, S, 1, 0.99)
    hp = hashnet.HashNetParams(A, H, 4, 1, 1)
    codes = np.arange(1, B + 1, dtype=np.uint32)
    for rows, ok in ((S * (S + 1) // 2, True), (3, False)):
        eng = _engine(cfg, H, B, noise_mode=NOISE_BUFFER)
        prior = eng.root_prepare(np.zeros((B, A), np.float32), np.ones((B, A)), np.full((B, A), 1 / 3), 0.25)
        eng.reset(prior, np.zeros(B), np.stack([hashnet.hidden_of(int(c), hp) for c in codes]), np.zeros(B, np.int32),
                  np.ones((B, A)))
        eng.search_hash(4, 1, 1, noise=np.random.default_rng(0).random((B, rows, A)))
        if ok:
            eng.check()
        else:
            with pytest.raises(EngineError, match='noise buffer'):
                eng.check()
            eng.check()   # the flag is cleared by the report
        eng.close()
