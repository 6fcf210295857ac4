"""The CPU checker (oracle/) against golden vectors produced by the reference itself."""
import numpy as np
import pytest

from helpers import load, hash_cases, assert_tree_equal
from muzero_amd import hashnet
from oracle import oracle as orc


def _run_hash_case(case, noise_rows=None):
    o = orc.Oracle(case.cfg, case.H, None if case.gumbel else
                   (case['c_table'] if case.has('c_table') else None))
    hp = hashnet.HashNetParams(case.A, case.H, case.logit_shift, case.reward_on, case.parent_predicts)
    logits = hashnet.logits_of(case.root_code, hp)
    legal = case['legal']
    if case.gumbel:
        prior = orc.gumbel_root_prior(logits, legal)
    else:
        prior = case['children_prior_logits'][0]  # root prior as the reference prepared it
    o.init(prior, hashnet.value_of(case.root_code), hashnet.hidden_of(case.root_code, hp),
           case.to_play, legal, case['gumbel_noise'] if case.gumbel else None)
    net = o.hash_net(case.logit_shift, case.reward_on, case.parent_predicts)
    nz = o.counter_noise(case.seed, case.idx) if noise_rows is None else o.buffer_noise(noise_rows)
    sel = o.search(net, nz)
    return o, sel


@pytest.mark.parametrize('name', ['search_hash_puct.npz', 'search_hash_gumbel.npz'])
def test_search_trees_bit_exact(name):
    for case in hash_cases(name):
        o, sel = _run_hash_case(case)
        assert_tree_equal(o, case)
        if not case.gumbel:
            assert sel.noise_calls == case.calls
            assert sel.noise_calls == int(o.node_visits.sum()) - 1 - case.S
        a, probs, q = o.to_act(float(case['temperature']), float(case['choice_u']))
        assert a == int(case['action']), (case.idx, a, int(case['action']))
        assert o.node_values[0] == float(case['root_value'])
        np.testing.assert_allclose(probs, case['probs'], rtol=1e-12, atol=1e-300)


def test_gumbel_root_prior_exact():
    for case in hash_cases('search_hash_gumbel.npz'):
        got = orc.gumbel_root_prior(case['root_logits'], case['legal'])
        ref = case['children_prior_logits'][0]
        assert got.dtype == ref.dtype and np.array_equal(got, ref)


def test_puct_root_prior():
    """Dirichlet mixing: f32 softmax is exact; the f64 log may differ by 1 ulp from NumPy's."""
    for case in hash_cases('search_hash_puct.npz'):
        got = orc.puct_root_prior(case['root_logits'], case['dirichlet'], 0.25, case['legal'])
        ref = case['children_prior_logits'][0]
        legal = case['legal'] != 0
        assert np.array_equal(got[~legal], ref[~legal])
        np.testing.assert_allclose(got[legal], ref[legal], rtol=0, atol=4e-16 * np.abs(ref[legal]).max() + 1e-300)


def test_stream_noise_cases():
    """PUCT with the uniforms the reference really drew from np.random (explicit rows)."""
    for case in hash_cases('search_stream_puct.npz', prefix='s'):
        o, sel = _run_hash_case(case, noise_rows=case['uniform_rows'])
        assert sel.noise_calls == case.calls
        assert_tree_equal(o, case)
        assert o.node_values[0] == float(case['root_value'])


@pytest.mark.parametrize('name', ['search_hash_puct.npz', 'search_hash_gumbel.npz'])
def test_per_node_functions(name):
    for case in hash_cases(name):
        if not case.has('node_qtransform'):
            continue
        o, _ = _run_hash_case(case)
        for n in range(case.S + 1):
            if case.gumbel:
                qt = o.qtransform_completed_by_mix_value(n)
                sel = o.gumbel_interior_select(n)
            else:
                qt = o.qtransform_by_parent_and_siblings(n)
                sel = o.puct_select(n, 1, np.zeros(case.A))
            assert np.array_equal(qt, case['node_qtransform'][n]), (case.idx, n)
            assert sel == int(case['node_select'][n]), (case.idx, n)


@pytest.mark.parametrize('fname', ['net_mlp_cartpole.npz', 'net_resnet_tictactoe.npz', 'net_mlp_synthetic.npz'])
def test_recorded_network_replay(fname):
    """Searches driven by recorded outputs of the reference's real networks."""
    from helpers import make_config
    z = load(fname)
    for ci in range(int(z['n_search_cases'])):
        p = f'r{ci}_'
        gumbel, A, S, players, H, to_play, seed, idx, calls = [int(x) for x in z[p + 'meta']]
        cfg = make_config(gumbel, A, S, players, float(z[p + 'gamma']), alpha=float(z[p + 'alpha']))
        o = orc.Oracle(cfg, H, None if gumbel else z[p + 'c_table'])
        prior = z[p + 'children_prior_logits'][0]
        if gumbel:
            assert np.array_equal(orc.gumbel_root_prior(z[p + 'root_logits'], z[p + 'legal']), prior)
        o.init(prior, z[p + 'root_value'], z[p + 'root_hidden'], to_play, z[p + 'legal'],
               z[p + 'gumbel_noise'] if gumbel else None)
        net = o.table_net(z[p + 'rec_logits'], z[p + 'rec_value'], z[p + 'rec_reward'],
                          z[p + 'rec_hidden'].reshape(S, -1))
        sel = o.search(net, o.counter_noise(seed, idx))

        class Case:
            def __getitem__(self, k):
                return z[p + k]
        assert_tree_equal(o, Case())
        a, probs, _ = o.to_act(1.0, float(z[p + 'choice_u']))
        assert a == int(z[p + 'action'])
        np.testing.assert_allclose(probs, z[p + 'probs'], rtol=1e-12, atol=1e-300)
        assert o.node_values[0] == float(z[p + 'search_value'])


def test_considered_visits_tables():
    z = load('misc.npz')
    for m, S in z['cv_pairs']:
        assert np.array_equal(orc.considered_visits(int(m), int(S)), z[f'cv_{m}_{S}']), (m, S)
    # known answers quoted from the reference's behaviour
    assert list(orc.considered_visits(1, 5)) == [0, 1, 2, 3, 4]
    assert list(orc.considered_visits(2, 6)) == [0, 0, 1, 1, 2, 2]


def test_numpy_kernels():
    z = load('misc.npz')
    assert np.array_equal(orc.np_expf(z['expf_in']).view(np.uint32), z['expf_out'].view(np.uint32))
    import ctypes as C
    L = orc.lib()
    for n in (2, 7, 8, 9, 16, 18, 33, 64):
        a = z[f'sum64_in_{n}']
        got = np.array([L.orc_np_sum_f64(r.ctypes.data_as(C.c_void_p), C.c_int(n)) for r in np.ascontiguousarray(a)])
        assert np.array_equal(got, z[f'sum64_out_{n}'])
        af = np.ascontiguousarray(a.astype(np.float32))
        got = np.array([L.orc_np_sum_f32(r.ctypes.data_as(C.c_void_p), C.c_int(n)) for r in af], dtype=np.float32)
        assert np.array_equal(got, z[f'sum32_out_{n}'])


def test_counter_noise_matches_host_definition():
    from muzero_amd import noise
    for seed, tree, call, a in [(0, 0, 0, 0), (7, 3, 11, 8), (2**40 + 5, 16383, 4000, 17)]:
        assert orc.lib().orc_noise_u(seed, tree, call, a) == noise.noise_u(seed, tree, call, a)


def test_nan_regime_matches_reference():
    """Roots whose float32 prior softmax underflows entirely: the reference's scores are NaN and NumPy's NaN
    rules decide (maximum / min / max keep a NaN, argmax returns the first one)."""
    import os
    from muzero_amd import hashnet
    from helpers import make_config
    z = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'nan_regime.npz'))
    A, S, H, B, m = (int(x) for x in z['meta'])
    cfg = make_config(True, A, S, 1, 0.5)
    cfg.max_considered_actions = m
    hp = hashnet.HashNetParams(A, H, 1, 0, 0)
    ref = orc.search_batch_hash(cfg, hp, H, z['codes'], z['legal'], np.zeros(B, np.int32), gumbel=z['gumbel'], seed=0,
                                n_threads=2)
    assert np.isnan(z['probs']).any(axis=1).sum() >= 5          # the fixture does contain the regime
    np.testing.assert_array_equal(ref.action, z['action'])
    np.testing.assert_array_equal(ref.root_value, z['root_value'])
    np.testing.assert_array_equal(ref.root_visits, z['root_visits'])
