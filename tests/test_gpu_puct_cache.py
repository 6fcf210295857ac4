"""GPU: the level-by-level PUCT search with cached decisions (csrc/tree_ops.h puct_decide: the choice at a node is kept
while no tie-break draw of 1e-7 * uniform[0, 1), mcts/action_selection.py:46, can change it) builds, bit for bit, the
trees of the same search that scores every node of every descent - same noise stream positions, same paths, same
statistics.  MZ_PUCT_CACHE=0 switches the table off."""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from helpers import make_config, assert_tree_equal
from muzero_amd.network import MuZeroNetwork
from muzero_amd.packing import pack_network_mfma
from test_gpu_chain_search import WIDE, ODD, _network

pytestmark = pytest.mark.gpu


def _search(monkeypatch, shape, peaked, B, S, cached, noise_mode='counter', stepped=0, tied=False):
    from muzero_amd import engine as E
    A, seed = shape.n_actions, 5
    net = _network(shape, peaked)
    if tied:   # a policy head without preferences: every fresh node scores all its actions alike but for the draws
        sd = {k: v.clone() for k, v in net.state_dict().items()}
        for k in sorted(k for k in sd if 'policy_network' in k)[-2:]:
            sd[k].zero_()
        net = MuZeroNetwork(shape, sd)
    gnet = MuZeroNetwork(shape, net.state_dict()).to('cuda')
    rng = np.random.default_rng(8)
    cfg = make_config(False, A, S, 1, 0.997)
    obs = torch.as_tensor(rng.standard_normal((B, shape.observation_dim[-1])).astype(np.float32), device='cuda')
    legal, tp = np.ones((B, A)), np.zeros(B, np.int32)
    legal[::4, min(2, A - 1)] = 0
    legal[1::7, :max(A - 2, 1)] = 0
    di = rng.dirichlet([0.25] * A, size=B)
    with torch.no_grad():
        logits, hidden, value = gnet.initial_flat(obs)
    monkeypatch.setenv('MZ_PUCT_CACHE', '1' if cached else '0')
    monkeypatch.setenv('MZ_SEARCH_CHAIN', '0')
    buffer = noise_mode == 'buffer'
    eng = E.SearchEngine(cfg, net.H, B, seed=seed, noise_mode=E.NOISE_BUFFER if buffer else E.NOISE_COUNTER)
    eng.set_network_mfma(*pack_network_mfma(gnet))
    noise = np.random.default_rng(2).random((B, (S + stepped + 1) * (S + stepped + 1), A)) if buffer else None
    eng.reset(eng.root_prepare(logits, legal, di, 0.25), value, hidden, tp, legal, None, tree_id0=0)
    if stepped:
        eng.select(noise, want_hidden=False)
        for i in range(stepped):
            eng.expand_backup_net(select_next=i + 1 < stepped, noise=noise)
    eng.search_mfma(noise)
    eng.check()
    step = max(1, B // 40)
    trees = [eng.export_tree(b) for b in sorted(set(range(0, B, step)) | {B - 1})]
    nv, calls = eng.stats()
    act = eng.to_act(0.0)
    eng.close()
    return trees, nv.cpu().numpy(), calls.cpu().numpy(), [x.cpu().numpy() for x in act[:3]]


def _same(a, b):
    np.testing.assert_array_equal(a[1], b[1])
    np.testing.assert_array_equal(a[2], b[2])      # selection calls made: the noise stream is where it would be
    for x, y in zip(a[3], b[3]):
        np.testing.assert_array_equal(x, y)
    for x, y in zip(a[0], b[0]):
        assert_tree_equal(SimpleNamespace(**x), y, hidden=True)


@pytest.mark.parametrize('peaked', [False, True])
@pytest.mark.parametrize('noise_mode', ['counter', 'buffer'])
def test_cached_puct_decisions_give_the_same_trees(peaked, noise_mode, monkeypatch):
    B, S = (150, 70) if noise_mode == 'counter' else (40, 30)
    ref = _search(monkeypatch, WIDE, peaked, B, S, cached=False, noise_mode=noise_mode)
    got = _search(monkeypatch, WIDE, peaked, B, S, cached=True, noise_mode=noise_mode)
    if peaked:
        assert ref[1].max() > S * 8        # chains: deep paths
    _same(got, ref)


def test_cached_puct_decisions_other_shapes_and_resumed_searches(monkeypatch):
    """21 actions on a small network; a search that continues trees grown by stepped simulations (nodes the table has
    never seen are scored in full until a simulation passes through them)."""
    ref = _search(monkeypatch, ODD, True, 70, 40, cached=False, stepped=7)
    got = _search(monkeypatch, ODD, True, 70, 40, cached=True, stepped=7)
    _same(got, ref)


def test_cached_puct_decisions_with_exact_ties(monkeypatch):
    """A policy head that prefers nothing: the scores of a fresh node differ by the tie-break draws alone, nothing is
    certified there, and the draws of the right call index must decide."""
    ref = _search(monkeypatch, ODD, False, 64, 50, cached=False, tied=True)
    got = _search(monkeypatch, ODD, False, 64, 50, cached=True, tied=True)
    _same(got, ref)


@pytest.mark.parametrize('A', [2, 5, 11, 16])
def test_cached_puct_decisions_with_fewer_actions_than_lanes_per_node(A, monkeypatch):
    """Short lists are decided by groups of up to 16 lanes per node: with fewer actions than that some lanes of a group
    have no action at all and must not contribute to the node's value range (MZ_SEARCH_LEVELS=1 sends these small
    action counts down the level-by-level path)."""
    monkeypatch.setenv('MZ_SEARCH_LEVELS', '1')
    shape = SimpleNamespace(network='mlp', observation_dim=[1, 1, 24], n_actions=A, n_stacked_observations=1,
                            encoding_size=24, fc_reward_layers=[32], fc_policy_layers=[32], fc_value_layers=[32],
                            fc_representation_layers=[32], fc_hidden_state_layers=[32], support_limit=10)
    for peaked in (False, True):
        ref = _search(monkeypatch, shape, peaked and A > 7, 90, 60, cached=False)
        got = _search(monkeypatch, shape, peaked and A > 7, 90, 60, cached=True)
        _same(got, ref)


def test_cached_puct_at_the_benchmark_size(monkeypatch):
    ref = _search(monkeypatch, WIDE, False, 1024, 200, cached=False)
    got = _search(monkeypatch, WIDE, False, 1024, 200, cached=True)
    _same(got, ref)


def test_cached_puct_in_a_search_longer_than_a_block_has_threads(monkeypatch):
    """More simulations than the tree kernel's block has threads: the exploration constants are staged up front by
    strided loops instead of one entry per thread behind the backup, and the square-root factors replace the table."""
    ref = _search(monkeypatch, ODD, True, 48, 300, cached=False)
    got = _search(monkeypatch, ODD, True, 48, 300, cached=True)
    assert ref[1].max() > 255
    _same(got, ref)


def test_level_blocks_evened_out_over_the_compute_units_give_the_same_trees(monkeypatch):
    """Where a network level has more blocks per tile than the tile's group of compute units, but not a multiple of it,
    its steps are cut into more, smaller blocks (MZ_LVL_PLAIN=1: the plain cut of 8 column tiles per block).  An
    output's products are summed in the same order either way: the trees must not differ in a single bit."""
    monkeypatch.setenv('MZ_LVL_PLAIN', '1')
    ref = _search(monkeypatch, WIDE, False, 1024, 40, cached=True)
    monkeypatch.delenv('MZ_LVL_PLAIN')
    got = _search(monkeypatch, WIDE, False, 1024, 40, cached=True)
    _same(got, ref)
