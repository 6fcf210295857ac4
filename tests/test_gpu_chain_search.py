"""GPU: the single-launch search that chains the network levels and the tree phases across workgroups through tagged
data in global memory (mcts/mcts.py:236-248 with network.py:105-126 as the leaf evaluator) builds, bit for bit, the
trees of the level-by-level search that launches every phase separately: same k order per network output, same
epilogue, same tree code.  MZ_SEARCH_CHAIN=2 makes the engine refuse any other path, so the comparison cannot pass
vacuously; every hand-off in the chained kernel carries data another workgroup computed, so a stale or torn read
changes a tree."""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from helpers import make_config, assert_tree_equal
from muzero_amd.network import MuZeroNetwork
from muzero_amd.packing import pack_network_mfma

pytestmark = pytest.mark.gpu

WIDE = SimpleNamespace(network='mlp', observation_dim=[1, 1, 128], n_actions=18, n_stacked_observations=1,
                       encoding_size=256, fc_reward_layers=[256], fc_policy_layers=[256], fc_value_layers=[256],
                       fc_representation_layers=[256], fc_hidden_state_layers=[256], support_limit=300)
# a smaller shape of the same kind: supports of at most 64 bins (the other epilogue), a hidden state that is not a
# multiple of 16, three hidden layers per head (more levels)
ODD = SimpleNamespace(network='mlp', observation_dim=[1, 1, 24], n_actions=21, n_stacked_observations=1,
                      encoding_size=40, fc_reward_layers=[48, 32], fc_policy_layers=[24], fc_value_layers=[48, 32],
                      fc_representation_layers=[32], fc_hidden_state_layers=[64, 48], support_limit=20)


def _network(shape, peaked):
    net = MuZeroNetwork(shape, seed=0)
    if peaked:
        sd = {k: v.clone() for k, v in net.state_dict().items()}
        key = sorted(k for k in sd if 'policy_network' in k and k.endswith('bias'))[-1]
        sd[key][7] += 6.0
        net = MuZeroNetwork(shape, sd)
    return net


def _search(monkeypatch, shape, gumbel, peaked, B, S, chain, G=None, twice=False):
    from muzero_amd.engine import SearchEngine
    A, seed = shape.n_actions, 9
    net = _network(shape, peaked)
    gnet = MuZeroNetwork(shape, net.state_dict()).to('cuda')
    rng = np.random.default_rng(3)
    cfg = make_config(gumbel, A, S, 1, 0.997)
    obs = torch.as_tensor(rng.standard_normal((B, shape.observation_dim[-1])).astype(np.float32), device='cuda')
    legal, tp = np.ones((B, A)), np.zeros(B, np.int32)
    legal[::5, 3] = 0
    di = None if gumbel else rng.dirichlet([0.25] * A, size=B)
    g = rng.gumbel(size=(B, A)) if gumbel else None
    with torch.no_grad():
        logits, hidden, value = gnet.initial_flat(obs)
    monkeypatch.setenv('MZ_SEARCH_CHAIN', '2' if chain else '0')
    if G:
        monkeypatch.setenv('MZ_CHAIN_G', str(G))
    else:
        monkeypatch.delenv('MZ_CHAIN_G', raising=False)
    eng = SearchEngine(cfg, net.H, B, seed=seed)
    eng.set_network_mfma(*pack_network_mfma(gnet))
    prior = eng.root_prepare(logits, legal, di, 0.25)
    for rep in range(2 if twice else 1):
        eng.reset(prior, value, hidden, tp, legal, g, tree_id0=0)
        eng.search_mfma()
    eng.check()
    step = max(1, B // 40)
    trees = [eng.export_tree(b) for b in sorted(set(range(0, B, step)) | {B - 1})]
    nv, calls = eng.stats()
    act = eng.to_act(0.0)
    eng.close()
    return trees, nv.cpu().numpy(), calls.cpu().numpy(), [x.cpu().numpy() for x in act[:3]]


def _same(a, b):
    np.testing.assert_array_equal(a[1], b[1])
    np.testing.assert_array_equal(a[2], b[2])
    for x, y in zip(a[3], b[3]):
        np.testing.assert_array_equal(x, y)
    for x, y in zip(a[0], b[0]):
        assert_tree_equal(SimpleNamespace(**x), y, hidden=True)


@pytest.mark.parametrize('gumbel', [False, True])
@pytest.mark.parametrize('G', [8, 16, 32])
def test_chained_search_is_the_level_by_level_search(gumbel, G, monkeypatch):
    """200 trees (a last tile that is not full) x 60 simulations down long chains, for clusters of 8, 16 and 32
    workgroups; the engine is reset and searched twice (tags of the first search must not pass for data)."""
    ref = _search(monkeypatch, WIDE, gumbel, True, 200, 60, chain=False)
    got = _search(monkeypatch, WIDE, gumbel, True, 200, 60, chain=True, G=G, twice=True)
    _same(got, ref)


@pytest.mark.parametrize('B', [1, 33, 1024])
def test_chained_search_batch_sizes(B, monkeypatch):
    """One tree, one tile and a tree, and the benchmark's 1 024 trees (random-init network: shallow trees)."""
    S = 200 if B == 1024 else 50
    ref = _search(monkeypatch, WIDE, False, False, B, S, chain=False)
    got = _search(monkeypatch, WIDE, False, False, B, S, chain=True)
    _same(got, ref)


@pytest.mark.parametrize('gumbel', [False, True])
def test_chained_search_other_shapes(gumbel, monkeypatch):
    """21 actions, hidden state of 40 floats, 41-bin supports, heads of two hidden layers (three network levels)."""
    ref = _search(monkeypatch, ODD, gumbel, True, 70, 40, chain=False)
    got = _search(monkeypatch, ODD, gumbel, True, 70, 40, chain=True)
    _same(got, ref)


def test_chained_search_refuses_what_it_does_not_serve(monkeypatch):
    """A network packed with shared places in the activation row is served level by level; demanding the chained
    kernel for it is an error that says why."""
    from muzero_amd.engine import SearchEngine
    net = _network(WIDE, False)
    gnet = MuZeroNetwork(WIDE, net.state_dict()).to('cuda')
    cfg = make_config(False, 18, 8, 1, 0.997)
    B = 4
    obs = torch.zeros((B, 128), device='cuda')
    with torch.no_grad():
        logits, hidden, value = gnet.initial_flat(obs)
    legal, tp = np.ones((B, 18)), np.zeros(B, np.int32)
    di = np.full((B, 18), 1 / 18)
    eng = SearchEngine(cfg, net.H, B, seed=1)
    eng.set_network_mfma(*pack_network_mfma(gnet, reuse=True))
    eng.reset(eng.root_prepare(logits, legal, di, 0.25), value, hidden, tp, legal, tree_id0=0)
    monkeypatch.setenv('MZ_SEARCH_CHAIN', '2')
    with pytest.raises((ValueError, RuntimeError), match='share a place'):
        eng.search_mfma()
    monkeypatch.setenv('MZ_SEARCH_CHAIN', '1')
    eng.search_mfma()     # falls back to the level-by-level launches
    eng.check()
    eng.close()
