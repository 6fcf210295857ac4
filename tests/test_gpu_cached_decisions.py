"""GPU: the Gumbel whole-search kernel that walks cached decisions (csrc/search_cached.h) against the kernel that
scores every node on the way down (k_search_mfma).  Both evaluate leaves with the same matrix-core code, so every
tree field must agree bit for bit - fresh searches, searches resumed after stepped simulations, illegal root
actions, either player to move, ragged batches, both tile sizes, 2 to 16 actions, records with 16- and 32-bit
visit / index rows."""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from helpers import make_config
from test_network_lowering import load_net, net_config
from muzero_amd.network import MuZeroNetwork

pytestmark = pytest.mark.gpu


def _run(monkeypatch, cached, tm, cfg, net, pm_io, logits, hidden, value, legal, tp, g, stepped, B):
    from muzero_amd.engine import SearchEngine
    monkeypatch.setenv('MZ_SEARCH_CACHED', '2' if cached else '0')   # 2: an error if the kernel does not serve the case
    monkeypatch.setenv('MZ_SEARCH_TM', str(tm))
    eng = SearchEngine(cfg, net.H, B, seed=2)
    eng.set_network_mfma(*pm_io)
    prior = eng.root_prepare(logits, legal, None, 0.0)
    eng.reset(prior, value, hidden, tp, legal, g)
    if stepped:
        eng.select(want_hidden=False)
        for i in range(stepped):
            eng.expand_backup_net(select_next=i + 1 < stepped)
    eng.search_mfma()
    action, rv, probs, q = eng.to_act(0.0)
    nv, _ = eng.stats()
    out = dict(action=action.cpu().numpy().copy(), value=rv.cpu().numpy().copy(), probs=probs.cpu().numpy().copy(),
               q=q.cpu().numpy().copy(), nv=nv.cpu().numpy().copy(),
               trees=[eng.export_tree(b) for b in sorted({x for x in (0, 1, B // 3, B // 2, B - 2, B - 1) if 0 <= x < B})])
    eng.close()
    return out


@pytest.mark.parametrize('S,B,stepped', [(100, 70, 0), (40, 33, 5), (7, 1, 0), (25, 129, 24)])
def test_cached_decisions_give_the_same_trees(S, B, stepped, monkeypatch):
    from muzero_amd.packing import pack_network_mfma
    z, net = load_net('resnet_tictactoe')
    net = MuZeroNetwork(net_config('resnet_tictactoe'), net.state_dict()).to('cuda')
    pm_io = pack_network_mfma(net)
    rng = np.random.default_rng(S + B)
    cfg = make_config(True, 9, S, 2, 1)
    obs = z['obs'][rng.integers(0, len(z['obs']), size=B)].astype(np.float32)
    with torch.no_grad():
        logits, hidden, value = net.initial_flat(torch.as_tensor(obs.reshape(B, -1), device='cuda'))
    legal = np.ones((B, 9))
    for b in range(0, B, 3):
        legal[b, rng.choice(9, size=rng.integers(1, 8), replace=False)] = 0
    tp = rng.integers(0, 2, size=B).astype(np.int32)
    g = rng.gumbel(size=(B, 9))
    want = _run(monkeypatch, False, 32, cfg, net, pm_io, logits, hidden, value, legal, tp, g, stepped, B)
    assert want['nv'].sum() > B * S            # the reference kernel really searched
    for tm in (32, 16):
        got = _run(monkeypatch, True, tm, cfg, net, pm_io, logits, hidden, value, legal, tp, g, stepped, B)
        for k in ('action', 'value', 'probs', 'q', 'nv'):
            np.testing.assert_array_equal(want[k], got[k], err_msg=f'{k} tm={tm}')
        for tw, tg in zip(want['trees'], got['trees']):
            for k in tw:
                np.testing.assert_array_equal(tw[k], tg[k], err_msg=f'{k} tm={tm}')


def test_cached_decisions_on_degenerate_priors(monkeypatch):
    """Roots whose float32 prior softmax underflows entirely (the reference scores NaN everywhere and plays the first
    legal action), and chains: a policy head that always prefers one action."""
    from muzero_amd.packing import pack_network_mfma
    z, net0 = load_net('resnet_tictactoe')
    sd = {k: v.clone() for k, v in net0.state_dict().items()}
    key = [k for k in sd if 'policy' in k and k.endswith('bias')][-1]
    sd[key][4] += 40.0
    net = MuZeroNetwork(net_config('resnet_tictactoe'), sd).to('cuda')
    pm_io = pack_network_mfma(net)
    rng = np.random.default_rng(1)
    B, S = 48, 60
    cfg = make_config(True, 9, S, 2, 1)
    obs = z['obs'][rng.integers(0, len(z['obs']), size=B)].astype(np.float32)
    with torch.no_grad():
        logits, hidden, value = net.initial_flat(torch.as_tensor(obs.reshape(B, -1), device='cuda'))
    logits = logits.clone()
    logits[::4] -= 200.0                         # exp underflows for every action of these roots
    legal = np.ones((B, 9))
    legal[1::4, :3] = 0
    tp = np.zeros(B, np.int32)
    g = rng.gumbel(size=(B, 9))
    want = _run(monkeypatch, False, 32, cfg, net, pm_io, logits, hidden, value, legal, tp, g, 0, B)
    got = _run(monkeypatch, True, 32, cfg, net, pm_io, logits, hidden, value, legal, tp, g, 0, B)
    assert want['nv'].max() > S * 10             # chains: leaves far below the root
    for k in ('action', 'value', 'nv'):
        np.testing.assert_array_equal(want[k], got[k], err_msg=k)
    np.testing.assert_array_equal(np.isnan(want['probs']), np.isnan(got['probs']))
    for tw, tg in zip(want['trees'], got['trees']):
        for k in tw:
            np.testing.assert_array_equal(tw[k], tg[k], err_msg=k)


@pytest.mark.parametrize('mode', ['tied', 'near_tied', 'deep'])
def test_decisions_without_a_certified_lead(mode, monkeypatch):
    """The cached-decision kernel takes a cheap argmax only when its lead is certified.  Policy heads that leave it no
    lead: identical logits for every action (every leaf joins the work list, every interior score is an exact tie:
    the reference's first-maximum rule decides), logits a few float32 ulps apart (the lead exists but is far below
    what the cheap evaluation resolves), and a head that prefers one action so strongly that every path is a chain as
    deep as the search is long (work lists of several rounds per simulation)."""
    from muzero_amd.packing import pack_network_mfma
    z, net0 = load_net('resnet_tictactoe')
    sd = {k: v.clone() for k, v in net0.state_dict().items()}
    wkey = [k for k in sd if 'policy' in k and k.endswith('weight')][-1]
    bkey = [k for k in sd if 'policy' in k and k.endswith('bias')][-1]
    if mode == 'tied':
        sd[wkey].zero_(); sd[bkey].fill_(0.25)
    elif mode == 'near_tied':
        sd[wkey].zero_()
        sd[bkey].copy_(torch.tensor(np.float32(0.25) + np.arange(9, dtype=np.float32) * np.float32(3e-8)))
    else:
        sd[wkey].zero_(); sd[bkey].zero_(); sd[bkey][7] = 85.0      # exp(85) is still a finite float32
        for head in ('value', 'reward'):                             # and no value differences to argue with it
            sd[[k for k in sd if head in k and k.endswith('weight')][-1]].zero_()
    net = MuZeroNetwork(net_config('resnet_tictactoe'), sd).to('cuda')
    pm_io = pack_network_mfma(net)
    rng = np.random.default_rng(5)
    B, S = 40, 90
    cfg = make_config(True, 9, S, 2, 1)
    obs = z['obs'][rng.integers(0, len(z['obs']), size=B)].astype(np.float32)
    with torch.no_grad():
        logits, hidden, value = net.initial_flat(torch.as_tensor(obs.reshape(B, -1), device='cuda'))
    legal = np.ones((B, 9))
    legal[::5, 2:5] = 0
    if mode == 'deep':                           # roots with one legal move: the whole search is one chain
        legal[1::2] = 0
        legal[1::2, 7] = 1
    tp = rng.integers(0, 2, size=B).astype(np.int32)
    g = rng.gumbel(size=(B, 9))
    want = _run(monkeypatch, False, 32, cfg, net, pm_io, logits, hidden, value, legal, tp, g, 0, B)
    if mode == 'deep':
        assert want['nv'].max() > S * S // 3     # leaf depth grows with every simulation
    for tm in (32, 16):
        got = _run(monkeypatch, True, tm, cfg, net, pm_io, logits, hidden, value, legal, tp, g, 0, B)
        for k in ('action', 'value', 'probs', 'q', 'nv'):
            np.testing.assert_array_equal(want[k], got[k], err_msg=f'{k} tm={tm}')
        for tw, tg in zip(want['trees'], got['trees']):
            for k in tw:
                np.testing.assert_array_equal(tw[k], tg[k], err_msg=f'{k} tm={tm}')


def _mlp(A, seed):
    shape = SimpleNamespace(network='mlp', observation_dim=[1, 1, 12], n_actions=A, n_stacked_observations=1,
                            encoding_size=24, fc_reward_layers=[16], fc_policy_layers=[16], fc_value_layers=[16],
                            fc_representation_layers=[16], fc_hidden_state_layers=[32], support_limit=10)
    net = MuZeroNetwork(shape, seed=seed)
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    for k in [k for k in sd if 'policy' in k and (k.endswith('weight') or k.endswith('bias'))][-2:]:
        sd[k] *= 8.0                               # priors with leaders: trees with some depth
    return MuZeroNetwork(shape, sd).to('cuda')


@pytest.mark.parametrize('A,rows', [(2, None), (3, None), (4, None), (5, None), (8, None), (12, None), (16, None),
                                    (4, 32), (9, 32), (16, 32)])
def test_cached_decisions_for_other_action_counts_and_row_widths(A, rows, monkeypatch):
    """The kernel is built for every action count a 16-lane group serves and for both widths of the visit / index rows
    (mz_create narrows them to 16 bits only where that shortens the record; MZ_IDX16=0 keeps 32 bits everywhere).
    MZ_SEARCH_CACHED=2 makes the engine refuse any other kernel."""
    from muzero_amd.packing import pack_network_mfma
    if rows == 32:
        monkeypatch.setenv('MZ_IDX16', '0')
    else:
        monkeypatch.delenv('MZ_IDX16', raising=False)
    net = _mlp(A, seed=A)
    pm_io = pack_network_mfma(net)
    rng = np.random.default_rng(100 + A)
    B, S = 77, 64
    cfg = make_config(True, A, S, 2, 0.997)
    obs = rng.standard_normal((B, 12)).astype(np.float32)
    with torch.no_grad():
        logits, hidden, value = net.initial_flat(torch.as_tensor(obs, device='cuda'))
    legal = np.ones((B, A))
    for b in range(0, B, 3):
        legal[b, rng.choice(A, size=rng.integers(1, A), replace=False)] = 0
    tp = rng.integers(0, 2, size=B).astype(np.int32)
    g = rng.gumbel(size=(B, A))
    want = _run(monkeypatch, False, 32, cfg, net, pm_io, logits, hidden, value, legal, tp, g, 0, B)
    assert want['nv'].sum() > B * S * 2
    for tm in (32, 16):
        got = _run(monkeypatch, True, tm, cfg, net, pm_io, logits, hidden, value, legal, tp, g, 0, B)
        for k in ('action', 'value', 'probs', 'q', 'nv'):
            np.testing.assert_array_equal(want[k], got[k], err_msg=f'{k} tm={tm}')
        for tw, tg in zip(want['trees'], got['trees']):
            for k in tw:
                np.testing.assert_array_equal(tw[k], tg[k], err_msg=f'{k} tm={tm}')
