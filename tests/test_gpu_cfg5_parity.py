"""GPU: the Atari-shaped configuration (hidden 256, 601-bin supports, 18 actions, 200 simulations, PUCT) with its REAL
network leg on every matrix-core path (activation tiles in LDS; level by level through global memory, one simulation
per call; the whole search level by level in one call), against the CPU
checker - random-init network (shallow trees) and a peaked-prior variant (deep trees).

Network outputs of two float32 evaluators differ in the last bits, and a PUCT search amplifies a last-bit difference
into another tree whenever two scores are nearly tied.  So the comparison is split into two exact statements:
  1. given the network outputs the GPU search itself produced (read back from its tree), the checker's search builds
     the SAME tree, bit for bit in every field - for every tree examined, in particular every tree whose root visit
     counts differ from the all-CPU run;
  2. those network outputs agree with the checker's dense executor on identical inputs: logits within 1e-5, hidden
     states 1e-4, values and rewards (601-bin expectations in float32) 5e-5.
Together they say that every difference between the two runs is caused by network differences below the tolerance.
The share of trees that agree outright is reported and bounded as well."""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from helpers import make_config, assert_tree_equal, replay_on_checker
from muzero_amd.network import MuZeroNetwork
from muzero_amd.packing import pack_network, pack_network_mfma

pytestmark = pytest.mark.gpu

WIDE = SimpleNamespace(network='mlp', observation_dim=[1, 1, 128], n_actions=18, n_stacked_observations=1,
                       encoding_size=256, fc_reward_layers=[256], fc_policy_layers=[256], fc_value_layers=[256],
                       fc_representation_layers=[256], fc_hidden_state_layers=[256], support_limit=300)


def _network(peaked):
    net = MuZeroNetwork(WIDE, seed=0)
    if peaked:   # a policy head that prefers one action by a wide margin: searches run down long chains
        sd = {k: v.clone() for k, v in net.state_dict().items()}
        key = sorted(k for k in sd if 'policy_network' in k and k.endswith('bias'))[-1]
        sd[key][7] += 6.0
        net = MuZeroNetwork(WIDE, sd)
    return net


@pytest.mark.parametrize('peaked', [False, True])
@pytest.mark.parametrize('path', ['tile', 'level', 'search'])
def test_cfg5_network_leg_against_checker(path, peaked, monkeypatch):
    from muzero_amd.engine import SearchEngine
    from oracle import oracle as orc
    if path != 'search':
        monkeypatch.setenv('MZ_NET_PATH', path)
    A, S, B, seed, tid0 = 18, 200, 256, 4, 1000
    net = _network(peaked)
    packed = pack_network(net)
    gnet = MuZeroNetwork(WIDE, net.state_dict()).to('cuda')
    rng = np.random.default_rng(21)
    cfg = make_config(False, A, S, 1, 0.997)
    obs = rng.standard_normal((B, 128)).astype(np.float32)
    with torch.no_grad():
        logits, hidden, value = net.initial_flat(torch.as_tensor(obs))
    logits, hidden, value = logits.numpy(), hidden.numpy(), value.numpy()
    legal, tp = np.ones((B, A)), np.zeros(B, np.int32)
    legal[::5, 3] = 0
    di = rng.dirichlet([0.25] * A, size=B)
    ref = orc.search_batch_net(cfg, packed, logits, hidden, value, legal, tp, dirichlet=di, frac=0.25, seed=seed,
                               tree_id0=tid0, n_threads=16)
    eng = SearchEngine(cfg, net.H, B, seed=seed)
    eng.set_network_mfma(*pack_network_mfma(gnet))
    prior = eng.root_prepare(logits, legal, di, 0.25)
    eng.reset(prior, value, hidden, tp, legal, tree_id0=tid0)
    if path == 'search':
        eng.search_mfma()     # 18 actions: the whole search runs level by level
    else:
        eng.select(want_hidden=False)
        for _ in range(S):
            eng.expand_backup_net(select_next=True)
    action, root_value, probs, _ = eng.to_act(0.0)
    nv, _ = eng.stats()
    nv = nv.cpu().numpy()
    trees = [eng.export_tree(b) for b in range(B)]
    eng.close()
    visits = np.stack([t['children_visits'][0] for t in trees])
    same = (visits == ref.root_visits).all(axis=1)
    close = np.abs(root_value.cpu().numpy() - ref.root_value) < 5e-5
    print(f'{path} peaked={peaked}: identical root visits {same.mean():.3f}, root values within 5e-5 {close.mean():.3f}, '
          f'mean leaf depth {nv.mean() / S - 1:.2f}')
    assert same.mean() >= 0.90 and (close | ~same).mean() >= 0.99
    if peaked:
        assert nv.mean() / S - 1 > 6          # deep trees indeed
    # 1. every tree that differs, and a sample of those that agree: exact given its own network outputs
    examine = sorted(set(np.flatnonzero(~same).tolist()) | set(range(0, B, 16)))
    worst = dict(logits=0.0, value=0.0, reward=0.0, hidden=0.0)
    for b in examine:
        t = trees[b]
        o = replay_on_checker(cfg, t, net.H, legal[b], int(tp[b]), seed, tid0 + b)
        assert_tree_equal(o, t, hidden=False)
        # 2. the outputs stored in the tree against the checker's dense executor on the same (parent state, action)
        par, act = t['parents'][1:], t['action_from_parent'][1:]
        lg, nh, val, rew = orc.net_recurrent_batch(packed, t['hidden_states'][par].reshape(S, -1), act)
        worst['logits'] = max(worst['logits'], np.abs(lg - t['children_prior_logits'][1:]).max())
        worst['value'] = max(worst['value'], np.abs(val - t['node_raw_values'][1:]).max())
        worst['reward'] = max(worst['reward'], np.abs(rew - t['children_rewards'][par, act]).max())
        worst['hidden'] = max(worst['hidden'], np.abs(nh - t['hidden_states'][1:].reshape(S, -1)).max())
    print('   largest differences to the dense executor on identical inputs:', worst)
    # logits and hidden states at the 1e-5 / 1e-4 of the small networks; the scalars are expectations over 601 bins with
    # support values up to 300 taken with float32 probabilities (eps 6e-8), which bounds two float32 evaluators'
    # agreement at about 300 * 6e-8 * a few = 5e-5 - observed: 1.2e-5
    assert worst['logits'] <= 1e-5 and worst['hidden'] <= 1e-4 and worst['value'] <= 5e-5 and worst['reward'] <= 5e-5


@pytest.mark.parametrize('gumbel', [False, True])
def test_whole_search_by_level_is_the_stepped_search(gumbel, monkeypatch):
    """The level-by-level whole search builds, bit for bit, the trees of the one-call-per-simulation path through the
    per-level kernels (same k order per output, same epilogue lanes, same tree code), also when an engine is reset and
    searched again."""
    from muzero_amd.engine import SearchEngine
    A, S, B, seed = 18, 60, 200, 9       # 200 trees: a last row tile that is not full
    net = _network(True)
    gnet = MuZeroNetwork(WIDE, net.state_dict()).to('cuda')
    rng = np.random.default_rng(3)
    cfg = make_config(gumbel, A, S, 1, 0.997)
    obs = torch.as_tensor(rng.standard_normal((B, 128)).astype(np.float32), device='cuda')
    legal, tp = np.ones((B, A)), np.zeros(B, np.int32)
    di = None if gumbel else rng.dirichlet([0.25] * A, size=B)
    g = rng.gumbel(size=(B, A)) if gumbel else None
    with torch.no_grad():
        logits, hidden, value = gnet.initial_flat(obs)

    def run(mode):
        monkeypatch.setenv('MZ_NET_PATH', 'level')
        eng = SearchEngine(cfg, net.H, B, seed=seed)
        eng.set_network_mfma(*pack_network_mfma(gnet))
        prior = eng.root_prepare(logits, legal, di, 0.25)
        eng.reset(prior, value, hidden, tp, legal, g, tree_id0=0)
        if mode == 'search':
            for rep in range(2):
                if rep:
                    eng.reset(prior, value, hidden, tp, legal, g, tree_id0=0)
                eng.search_mfma()
        else:
            eng.select(want_hidden=False)
            for _ in range(S):
                eng.expand_backup_net(select_next=True)
        eng.check()
        trees = [eng.export_tree(b) for b in range(0, B, 7)]
        nv, _ = eng.stats()
        eng.close()
        return trees, nv.cpu().numpy()

    ref, nv_ref = run('step')
    got, nv = run('search')
    np.testing.assert_array_equal(nv, nv_ref)
    for x, y in zip(got, ref):
        assert_tree_equal(SimpleNamespace(**x), y, hidden=True)
