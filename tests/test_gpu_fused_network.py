"""GPU: the fused search whose leaves are evaluated in-kernel by a lowered network.

Network outputs on the GPU (fma chains, device expf/expm1f) differ from the host's in the last
bits, so trees are compared the way the north star states it: visit counts identical for
(nearly) every tree, values within 1e-5; the bit-exact checks given identical network outputs
live in test_gpu_tree_parity.py."""
import numpy as np
import pytest
import torch

from helpers import make_config, assert_tree_equal, replay_on_checker
from test_network_lowering import load_net, net_config
from types import SimpleNamespace

from muzero_amd.packing import pack_network, pack_network_mfma
from muzero_amd.network import MuZeroNetwork

pytestmark = pytest.mark.gpu


def _roots(net, B, obs_shape, rng):
    obs = rng.uniform(-0.05, 0.05, size=(B,) + obs_shape).astype(np.float32)
    logits, hidden, value = net.initial_inference(torch.as_tensor(obs))
    return obs, logits.numpy(), hidden.numpy().reshape(B, -1), value.numpy()


@pytest.mark.parametrize('gumbel', [False, True])
def test_fused_mlp_search_vs_checker(gumbel, monkeypatch):
    from muzero_amd.engine import SearchEngine
    from oracle import oracle as orc
    z, net = load_net('mlp_cartpole')
    packed = pack_network(net)
    B, A, S = 1024, 2, 50
    rng = np.random.default_rng(3)
    cfg = make_config(gumbel, A, S, 1, 0.997)
    obs, logits, hidden, value = _roots(net, B, (1, 1, 4), rng)
    legal = np.ones((B, A))
    tp = np.zeros(B, np.int32)
    g = rng.gumbel(size=(B, A)) if gumbel else None
    di = None if gumbel else rng.dirichlet([0.25] * A, size=B)
    ref = orc.search_batch_net(cfg, packed, logits, hidden, value, legal, tp, dirichlet=di, frac=0.25, gumbel=g,
                               seed=5, tree_id0=0, n_threads=8)
    for lanes, reg in ((2, '0'), (16, '0'), (64, '0'), (64, '1')):   # lane groups of every width, then trees in registers
        monkeypatch.setenv('MZ_TREE_REG', reg)
        eng = SearchEngine(cfg, packed.H, B, seed=5, lanes_per_tree=lanes)
        eng.set_network(packed)
        prior = eng.root_prepare(logits, legal, di, 0.25)
        eng.reset(prior, value, hidden, tp, legal, g)
        eng.search_net()
        action, root_value, probs, _ = eng.to_act(0.0)
        nv, _ = eng.stats()
        visits = np.stack([eng.export_tree(b)['children_visits'][0] for b in range(0, B, 8)])
        same = (visits == ref.root_visits[::8]).all(axis=1)
        assert same.mean() >= 0.98, (lanes, same.mean())
        rv = root_value.cpu().numpy()
        close = np.abs(rv - ref.root_value) < 1e-5
        assert close.mean() >= 0.98, (lanes, close.mean())
        assert (action.cpu().numpy() == ref.action).mean() >= 0.98
        assert abs(int(nv.sum()) - int(ref.node_visits.sum())) <= 0.01 * ref.node_visits.sum()
        # the trees that differ are explained, not just counted: with the leaf outputs the GPU search itself produced the
        # checker rebuilds the very same tree, and those outputs are within tolerance of the checker's own evaluation
        for k in np.flatnonzero(~same)[:24]:
            b = int(k) * 8
            t = eng.export_tree(b)
            o = replay_on_checker(cfg, t, packed.H, legal[b], int(tp[b]), seed=5, tree_id=b, gumbel_noise=None if g is None else g[b])
            assert_tree_equal(o, t, hidden=False)
            par, act = t['parents'][1:], t['action_from_parent'][1:]
            lg, nh, val, rew = orc.net_recurrent_batch(packed, t['hidden_states'][par].reshape(S, -1), act)
            assert np.abs(lg - t['children_prior_logits'][1:]).max() <= 1e-5 and np.abs(val - t['node_raw_values'][1:]).max() <= 1e-5
            assert np.abs(rew - t['children_rewards'][par, act]).max() <= 1e-5
            assert np.abs(nh - t['hidden_states'][1:].reshape(S, -1)).max() <= 1e-4
        eng.close()


def test_fused_equals_step_mode_with_torch_network():
    """Same trees whether leaves are evaluated inside the fused kernel or by torch between the
    select and expand kernels (both on the GPU)."""
    from muzero_amd.engine import SearchEngine
    z, net = load_net('mlp_cartpole')
    packed = pack_network(net)
    B, A, S = 256, 2, 50
    rng = np.random.default_rng(4)
    cfg = make_config(False, A, S, 1, 0.997)
    obs, logits, hidden, value = _roots(net, B, (1, 1, 4), rng)
    legal, tp = np.ones((B, A)), np.zeros(B, np.int32)
    di = rng.dirichlet([0.25] * A, size=B)
    gnet = MuZeroNetwork(net_config('mlp_cartpole'), net.state_dict()).to('cuda')
    eng = SearchEngine(cfg, packed.H, B, seed=9)
    eng.set_network(packed)
    prior = eng.root_prepare(logits, legal, di, 0.25)
    eng.reset(prior, value, hidden, tp, legal)
    eng.search_net()
    a1, v1, p1, _ = eng.to_act(0.0)
    eng.reset(prior, value, hidden, tp, legal)
    for _ in range(S):
        parent, action, phid = eng.select()
        lg, nh, val, rew = gnet.recurrent_flat(phid, gnet.encode_actions(action))
        eng.expand_backup(lg, val, rew, nh)
    a2, v2, p2, _ = eng.to_act(0.0)
    assert (a1 == a2).float().mean() >= 0.98
    assert ((v1 - v2).abs() < 1e-5).float().mean() >= 0.98
    eng.close()


@pytest.mark.parametrize('gumbel,deep', [(False, False), (True, False), (False, True)])
def test_lds_resident_tree_is_bit_identical(gumbel, deep, monkeypatch):
    """Small batches keep their trees in LDS and score every node of a 2-action tree at once, or (PUCT)
    keep the whole tree in registers, one node per lane; the arithmetic is the same as the global-memory,
    level-by-level kernel, so whole trees must be equal bit for bit - also when the fused search takes over
    after a few stepped simulations, with two players, and with some root actions illegal."""
    from muzero_amd.engine import SearchEngine
    z, net = load_net('mlp_cartpole')
    if deep:   # a policy head that all but always prefers action 0: trees degenerate into chains 40+ levels deep
        sd = {k: v.clone() for k, v in net.state_dict().items()}
        sd['prediction_network.policy_network.2.bias'][0] += 30.0
        net = MuZeroNetwork(net_config('mlp_cartpole'), sd)
    packed = pack_network(net)
    B, A, S = 96, 2, 50
    rng = np.random.default_rng(8)
    cfg = make_config(gumbel, A, S, 2, 0.997)
    obs, logits, hidden, value = _roots(net, B, (1, 1, 4), rng)
    legal, tp = np.ones((B, A)), rng.integers(0, 2, size=B).astype(np.int32)
    legal[1::7, 0] = 0
    legal[3::7, 1] = 0
    g = rng.gumbel(size=(B, A)) if gumbel else None
    di = None if gumbel else rng.dirichlet([0.25] * A, size=B)
    gnet = MuZeroNetwork(net_config('mlp_cartpole'), net.state_dict()).to('cuda')
    for stepped in (0, 7):
        out = []
        # trees in global memory, in LDS, in registers (hidden states in LDS), in registers with hidden states too
        # ... and the same with those hidden states in LDS again (what batches that fill the chip use)
        for flag, reg, tiny, hlds in (('0', '0', '0', '0'), ('1', '0', '0', '0'), ('1', '1', '0', '0'), ('1', '1', '1', '0'),
                                      ('1', '1', '1', '1')):
            monkeypatch.setenv('MZ_TREE_LDS', flag)
            monkeypatch.setenv('MZ_TREE_REG', reg)
            monkeypatch.setenv('MZ_NET_TINY', tiny)
            monkeypatch.setenv('MZ_HID_LDS', hlds)
            eng = SearchEngine(cfg, packed.H, B, seed=13, lanes_per_tree=64)
            eng.set_network(packed)
            prior = eng.root_prepare(logits, legal, di, 0.25)
            eng.reset(prior, value, hidden, tp, legal, g)
            for _ in range(stepped):
                parent, action, phid = eng.select()
                lg, nh, val, rew = gnet.recurrent_flat(phid, gnet.encode_actions(action))
                eng.expand_backup(lg, val, rew, nh)
            eng.search_net()
            nv, calls = eng.stats()
            out.append(([eng.export_tree(b) for b in range(0, B, 5)], nv.cpu().numpy().copy(), calls.cpu().numpy().copy()))
            eng.close()
        for other in out[1:]:
            for t0, t1 in zip(out[0][0], other[0]):
                for k in t0:
                    np.testing.assert_array_equal(t0[k], t1[k], err_msg=k)
            np.testing.assert_array_equal(out[0][1], other[1])
            np.testing.assert_array_equal(out[0][2], other[2])
        if deep:   # node visits per tree = sum of (leaf depth + 1): chains reach far beyond 32 levels
            assert out[0][1].max() > S * 20


def test_network_too_large_for_lds_is_rejected():
    from muzero_amd.engine import SearchEngine
    z, net = load_net('resnet_tictactoe')
    packed = pack_network(net)
    cfg = make_config(True, 9, 10, 2, 1)
    eng = SearchEngine(cfg, packed.H, 4)
    eng.set_network(packed)
    eng.reset(np.zeros((2, 9), np.float32), np.zeros(2), np.zeros((2, packed.H), np.float32), np.zeros(2, np.int32),
              np.ones((2, 9)), np.zeros((2, 9)))
    with pytest.raises(ValueError):
        eng.search_net()
    eng.close()


@pytest.mark.parametrize('name,B', [('resnet_tictactoe', 100), ('mlp_synthetic', 77), ('mlp_cartpole', 33),
                                    ('mlp_wide256', 40)])
def test_mfma_leaf_evaluation_matches_torch_network(name, B, monkeypatch):
    """Leaves evaluated by the matrix-core kernel vs the same lowered network run by torch:
    first-simulation outputs within 1e-5 (hidden 1e-4), whole searches agree tree by tree."""
    from muzero_amd.engine import SearchEngine
    from muzero_amd.packing import pack_network_mfma
    if name == 'mlp_wide256':   # hidden 256, 601-bin supports: only a 16-tree activation tile fits in LDS
        from types import SimpleNamespace
        wide = SimpleNamespace(network='mlp', observation_dim=[1, 1, 128], n_actions=18, n_stacked_observations=1,
                               encoding_size=256, fc_reward_layers=[256], fc_policy_layers=[256], fc_value_layers=[256],
                               fc_representation_layers=[256], fc_hidden_state_layers=[256], support_limit=300)
        net = MuZeroNetwork(wide, seed=0)
        z = {'obs': np.random.default_rng(0).normal(size=(64, 1, 1, 128))}
        gnet = MuZeroNetwork(wide, net.state_dict()).to('cuda')
    else:
        z, net = load_net(name)
        gnet = MuZeroNetwork(net_config(name), net.state_dict()).to('cuda')
    pm, io = pack_network_mfma(net)
    A, S = net.A, 24
    gumbel = name == 'resnet_tictactoe'
    cfg = make_config(gumbel, A, S, 2 if gumbel else 1, 1 if gumbel else 0.997)
    rng = np.random.default_rng(11)
    obs = z['obs'][rng.integers(0, len(z['obs']), size=B)].astype(np.float32)
    logits, hidden, value = net.initial_inference(torch.as_tensor(obs))
    logits, hidden, value = logits.numpy(), hidden.numpy().reshape(B, -1), value.numpy()
    legal, tp = np.ones((B, A)), np.zeros(B, np.int32)
    g = rng.gumbel(size=(B, A)) if gumbel else None
    di = None if gumbel else rng.dirichlet([0.25] * A, size=B)
    results = []
    for mode in ('mfma', 'mfma_fused', 'mfma_search', 'mfma_level', 'torch'):
        # 'mfma_level': layer by layer with the activations in global memory (what small batches use)
        monkeypatch.setenv('MZ_NET_PATH', 'level' if mode == 'mfma_level' else 'tile')
        eng = SearchEngine(cfg, net.H, B, seed=2)
        eng.set_network_mfma(pm, io)
        prior = eng.root_prepare(logits, legal, di, 0.25)
        eng.reset(prior, value, hidden, tp, legal, g)
        if mode == 'mfma_fused':   # selection of the next simulation in the same launch as the backup
            eng.select(want_hidden=False)
            for sim in range(S):
                eng.expand_backup_net(select_next=True)
        if mode == 'mfma_search':  # the whole simulation loop in one call (one launch up to 16 actions, level by level above)
            eng.search_mfma()
        for sim in range(S if mode in ('mfma', 'mfma_level', 'torch') else 0):
            parent, action, phid = eng.select()
            if mode in ('mfma', 'mfma_level'):
                if sim == 0:
                    lg, nh, val, rew = gnet.recurrent_flat(phid, gnet.encode_actions(action))
                eng.expand_backup_net()
                if sim == 0:
                    t0 = eng.export_tree(0), eng.export_tree(B - 1)
                    for b, t in zip((0, B - 1), t0):
                        np.testing.assert_allclose(t['children_prior_logits'][1], lg[b].cpu().numpy(), atol=1e-5, rtol=0)
                        np.testing.assert_allclose(t['node_values'][1], float(val[b]), atol=1e-5, rtol=0)
                        np.testing.assert_allclose(t['hidden_states'][1], nh[b].cpu().numpy(), atol=1e-4, rtol=0)
                        a0 = int(action[b])
                        np.testing.assert_allclose(t['children_rewards'][0][a0], float(rew[b]), atol=1e-5, rtol=0)
            else:
                lg, nh, val, rew = gnet.recurrent_flat(phid, gnet.encode_actions(action))
                eng.expand_backup(lg, val, rew, nh)
        a, v, p, _ = eng.to_act(0.0)
        results.append((a.cpu().numpy().copy(), v.cpu().numpy().copy()))
        if mode == 'mfma':
            mfma_trees = [eng.export_tree(b) for b in range(B)]
        eng.close()
    # trees on which the matrix-core leaves and torch's lead to different searches: given its own leaf outputs the
    # matrix-core search is exactly the checker's, and those outputs are within tolerance of the checker's dense executor
    from oracle import oracle as orc
    packed = pack_network(net)
    differ = np.flatnonzero((results[0][0] != results[4][0]) | (np.abs(results[0][1] - results[4][1]) >= 1e-5))
    tol = 1e-5 if net.support_limit <= 32 else 5e-5   # 601-bin float32 expectations: see test_gpu_cfg5_parity.py
    for b in list(differ[:16]) + [0]:
        t = mfma_trees[int(b)]
        o = replay_on_checker(cfg, t, net.H, legal[b], int(tp[b]), seed=2, tree_id=int(b), gumbel_noise=None if g is None else g[b])
        assert_tree_equal(o, t, hidden=False)
        par, act = t['parents'][1:], t['action_from_parent'][1:]
        lg, nh, val, rew = orc.net_recurrent_batch(packed, t['hidden_states'][par].reshape(S, -1), act)
        assert np.abs(lg - t['children_prior_logits'][1:]).max() <= 1e-5 and np.abs(val - t['node_raw_values'][1:]).max() <= tol
        assert np.abs(rew - t['children_rewards'][par, act]).max() <= tol
        if name != 'resnet_tictactoe':   # per-plane min-max over 9 cells amplifies last-bit differences of a flat plane
            assert np.abs(nh - t['hidden_states'][1:].reshape(S, -1)).max() <= 1e-4
        # ... so next hidden states are bounded where they are well conditioned, on every network: difference x range of
        # the scaled plane
        from helpers import hidden_conditioned_error
        assert hidden_conditioned_error(net, t['hidden_states'][par].reshape(S, -1), act,
                                        t['hidden_states'][1:].reshape(S, -1)) <= 2e-6
    for k in (1, 2, 3):   # fewer launches / other data placement, the very same search
        if name == 'mlp_wide256' and (k == 3 or (k == 2 and A > 16)):
            # long supports are reduced by 64 lanes instead of 16 on the level-by-level paths (with more than 16
            # actions the whole-search call is one of them): float32 sums in another order
            assert (results[0][0] == results[k][0]).mean() >= 0.97
            assert (np.abs(results[0][1] - results[k][1]) < 1e-5).mean() >= 0.97
            continue
        np.testing.assert_array_equal(results[0][0], results[k][0])
        np.testing.assert_array_equal(results[0][1], results[k][1])
    if A > 16:   # the two level-by-level paths (whole search in one call, one call per simulation): the very same search
        np.testing.assert_array_equal(results[2][0], results[3][0])
        np.testing.assert_array_equal(results[2][1], results[3][1])
    assert (results[0][0] == results[4][0]).mean() >= 0.97
    assert (np.abs(results[0][1] - results[4][1]) < 1e-5).mean() >= 0.97


@pytest.mark.parametrize('lanes', [16, 64])
def test_networks_near_the_lds_limit_fit_or_fail_cleanly(lanes):
    """Weights, per-group activations and the kernel's own static tables share one CU's 160 KB of LDS.
    Networks sized around that limit must either run (agreeing with the checker) or be rejected with an
    error by the host - never reach the launch with more LDS than the hardware has."""
    from types import SimpleNamespace
    from muzero_amd.engine import SearchEngine
    from oracle import oracle as orc
    B, A, S = 64, 2, 12
    ran = rejected = 0
    for width in (64, 72, 80, 88, 96, 104):
        ncfg = SimpleNamespace(network='mlp', observation_dim=[1, 1, 4], n_actions=A, n_stacked_observations=1,
                               encoding_size=64, fc_reward_layers=[width], fc_policy_layers=[width],
                               fc_value_layers=[width], fc_representation_layers=[], fc_hidden_state_layers=[width],
                               support_limit=10)
        net = MuZeroNetwork(ncfg, seed=width)
        packed = pack_network(net)
        rng = np.random.default_rng(width)
        cfg = make_config(False, A, S, 1, 0.997)
        obs, logits, hidden, value = _roots(net, B, (1, 1, 4), rng)
        legal, tp = np.ones((B, A)), np.zeros(B, np.int32)
        di = rng.dirichlet([0.25] * A, size=B)
        eng = SearchEngine(cfg, packed.H, B, seed=3, lanes_per_tree=lanes)
        try:
            eng.set_network(packed)
            prior = eng.root_prepare(logits, legal, di, 0.25)
            eng.reset(prior, value, hidden, tp, legal)
            eng.search_net()
        except ValueError:
            rejected += 1
            eng.close()
            continue
        ran += 1
        ref = orc.search_batch_net(cfg, packed, logits, hidden, value, legal, tp, dirichlet=di, frac=0.25, seed=3,
                                   tree_id0=0, n_threads=8)
        nv, _ = eng.stats()
        assert (nv.cpu().numpy() == ref.node_visits).mean() >= 0.95
        _, rv, _, _ = eng.to_act(0.0)
        assert (np.abs(rv.cpu().numpy() - ref.root_value) < 1e-5).mean() >= 0.95
        eng.close()
    assert ran >= 1 and rejected >= 1, (ran, rejected)


@pytest.mark.parametrize('B,temperature', [(70, 1.0), (70, 0.0), (2000, 0.5)])
def test_search_and_final_rule_in_one_call(B, temperature):
    """mz_search_run_act = mz_search_run + mz_to_act: identical outputs whether the search kernel runs the final
    rule itself (small batches of 2-action PUCT trees) or the rule is a second launch (large batch)."""
    from muzero_amd.engine import SearchEngine
    z, net = load_net('mlp_cartpole')
    packed = pack_network(net)
    A, S = 2, 30
    rng = np.random.default_rng(B)
    cfg = make_config(False, A, S, 1, 0.997)
    obs, logits, hidden, value = _roots(net, B, (1, 1, 4), rng)
    legal, tp = np.ones((B, A)), np.zeros(B, np.int32)
    legal[5::9, 1] = 0
    di = rng.dirichlet([0.25] * A, size=B)
    u = rng.random(B)
    outs = []
    for fused in (False, True):
        eng = SearchEngine(cfg, packed.H, B, seed=21)
        eng.set_network(packed)
        prior = eng.root_prepare(logits, legal, di, 0.25)
        eng.reset(prior, value, hidden, tp, legal)
        if fused:
            res = eng.search_net_act(temperature, u)
        else:
            eng.search_net()
            res = eng.to_act(temperature, u)
        outs.append([x.cpu().numpy().copy() for x in res])
        eng.close()
    for x, y in zip(*outs):
        np.testing.assert_array_equal(x, y)


@pytest.mark.parametrize('B,gumbel', [(70, False), (1, False), (2000, False), (70, True)])
def test_whole_move_in_one_call(B, gumbel):
    """mz_search_move = mz_search_begin_net + mz_search_run + mz_to_act: identical outputs and trees whether
    the move is a single launch (small batches of 2-action PUCT trees) or three."""
    from muzero_amd.engine import SearchEngine
    z, net = load_net('mlp_cartpole')
    packed = pack_network(net)
    A, S = 2, 30
    rng = np.random.default_rng(B + gumbel)
    cfg = make_config(gumbel, A, S, 1, 0.997)
    obs = rng.uniform(-0.05, 0.05, size=(B, 4)).astype(np.float32)
    legal, tp = np.ones((B, A)), np.zeros(B, np.int32)
    legal[5::9, 1] = 0
    di = None if gumbel else rng.dirichlet([0.25] * A, size=B)
    g = rng.gumbel(size=(B, A)) if gumbel else None
    u = rng.random(B)
    outs = []
    for fused in (False, True):
        eng = SearchEngine(cfg, packed.H, B, seed=21)
        eng.set_network(packed)
        if fused:
            res = eng.search_move(obs, legal, tp, di, 0.25, g, 7, 1.0, u)
        else:
            eng.begin_net(obs, legal, tp, di, 0.25, g, tree_id0=7)
            eng.search_net()
            res = eng.to_act(1.0, u)
        nv, calls = eng.stats()
        trees = [eng.export_tree(b) for b in range(0, B, max(1, B // 5))]
        outs.append(([x.cpu().numpy().copy() for x in res] + [nv.cpu().numpy().copy(), calls.cpu().numpy().copy()], trees))
        eng.close()
    for x, y in zip(outs[0][0], outs[1][0]):
        np.testing.assert_array_equal(x, y)
    for t0, t1 in zip(outs[0][1], outs[1][1]):
        for k in t0:
            np.testing.assert_array_equal(t0[k], t1[k], err_msg=k)


@pytest.mark.parametrize('S,B', [(1, 1), (2, 3), (63, 5), (64, 5)])
def test_register_tree_edge_shapes(S, B, monkeypatch):
    """The smallest search, a tree that fills all 64 lanes (63 simulations) and one that no longer fits (64):
    the register path - or whatever takes over from it - equals the lane-group path bit for bit, also with a
    root whose actions are all illegal."""
    from muzero_amd.engine import SearchEngine
    z, net = load_net('mlp_cartpole')
    packed = pack_network(net)
    A = 2
    rng = np.random.default_rng(S * 10 + B)
    cfg = make_config(False, A, S, 2, 0.997)
    obs, logits, hidden, value = _roots(net, B, (1, 1, 4), rng)
    legal, tp = np.ones((B, A)), rng.integers(0, 2, size=B).astype(np.int32)
    legal[-1] = 0   # no legal action at the last root
    di = rng.dirichlet([0.25] * A, size=B)
    out = []
    for reg in ('0', '1'):
        monkeypatch.setenv('MZ_TREE_REG', reg)
        monkeypatch.setenv('MZ_TREE_LDS', '0')
        eng = SearchEngine(cfg, packed.H, B, seed=2)
        eng.set_network(packed)
        prior = eng.root_prepare(logits, legal, di, 0.25)
        eng.reset(prior, value, hidden, tp, legal)
        res = eng.search_net_act(1.0, rng.random(B) * 0 + 0.37)
        nv, calls = eng.stats()
        out.append(([x.cpu().numpy().copy() for x in res], [eng.export_tree(b) for b in range(B)], nv.cpu().numpy().copy(),
                    calls.cpu().numpy().copy()))
        eng.close()
    for x, y in zip(out[0][0], out[1][0]):
        np.testing.assert_array_equal(x, y)
    for t0, t1 in zip(out[0][1], out[1][1]):
        for k in t0:
            np.testing.assert_array_equal(t0[k], t1[k], err_msg=k)
    np.testing.assert_array_equal(out[0][2], out[1][2])
    np.testing.assert_array_equal(out[0][3], out[1][3])


def test_network_without_a_tile_schedule_is_evaluated_level_by_level():
    """A board ResNet wide enough that neither tile size of the tile-resident kernels serves it (a 32-tree tile of its
    activations exceeds LDS, the 16-tree tile would need more than 64 jobs per wave; layers of 25 input blocks, more
    than a level kernel requests up front): both the one-call search and mz_expand_backup_net go level by level on
    their own, build the same trees, and agree with torch-evaluated leaves."""
    from muzero_amd.engine import SearchEngine
    ncfg = SimpleNamespace(network='resnet', observation_dim=[3, 4, 4], n_actions=9, n_stacked_observations=1, blocks=3,
                           channels=24, reduced_channels_reward=8, reduced_channels_policy=8, reduced_channels_value=8,
                           resnet_fc_reward_layers=[16], resnet_fc_policy_layers=[16], resnet_fc_value_layers=[16],
                           downsample=False, support_limit=10)
    net = MuZeroNetwork(ncfg, seed=3)
    gnet = MuZeroNetwork(ncfg, net.state_dict()).to('cuda')
    B, S, A = 70, 12, 9
    rng = np.random.default_rng(8)
    cfg = make_config(True, A, S, 2, 1)
    obs = rng.uniform(-1, 1, size=(B, 3, 4, 4)).astype(np.float32)
    logits, hidden, value = net.initial_inference(torch.as_tensor(obs))
    logits, hidden, value = logits.numpy(), hidden.numpy().reshape(B, -1), value.numpy()
    legal, tp, g = np.ones((B, A)), np.zeros(B, np.int32), rng.gumbel(size=(B, A))
    results = []
    for mode in ('net', 'torch', 'search'):
        eng = SearchEngine(cfg, net.H, B, seed=1)
        eng.set_network_mfma(*pack_network_mfma(net))
        eng.reset(eng.root_prepare(logits, legal), value, hidden, tp, legal, g)
        if mode == 'search':
            eng.search_mfma()
        for _ in range(S if mode != 'search' else 0):
            parent, action, phid = eng.select()
            if mode == 'net':
                eng.expand_backup_net()
            else:
                lg, nh, val, rew = gnet.recurrent_flat(phid, gnet.encode_actions(action))
                eng.expand_backup(lg, val, rew, nh)
        a, v, _, _ = eng.to_act(0.0)
        eng.check()
        nv, _ = eng.stats()
        results.append((a.cpu().numpy().copy(), v.cpu().numpy().copy(), nv.cpu().numpy().copy()))
        eng.close()
    assert (results[0][0] == results[1][0]).mean() >= 0.95
    assert (np.abs(results[0][1] - results[1][1]) < 1e-5).mean() >= 0.95
    for i in range(3):   # one call or one call per simulation: the very same search
        np.testing.assert_array_equal(results[0][i], results[2][i])
