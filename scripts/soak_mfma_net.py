"""Soak test (not part of the suite): leaves evaluated on the matrix cores, on random MLP and ResNet
networks and batch sizes.  (1) One network step against the same network run by torch (1e-5; hidden
1e-4); (0) the initial inference on the device against torch (same tolerances); (2) the stepped, backup+selection-fused, whole-search and layer-by-layer launch modes must give the
same trees bit for bit (long supports on the layer-by-layer path excepted: another float32 sum order);
(3) whole searches agree with torch-evaluated leaves on (nearly) every tree.  Boards start at 3x3: on
2x2 boards the reference's per-plane min-max scaling (4 cells per plane) amplifies a one-ulp difference in a
hidden state by 3-10x per recurrent step, so two float32 evaluators drift apart by more than 1e-5 along deep
paths whatever their quality.

    python scripts/soak_mfma_net.py [seed=1] [trials=40]
"""
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from helpers import make_config, assert_tree_equal, replay_on_checker
from muzero_amd.engine import SearchEngine
from muzero_amd.network import MuZeroNetwork
from muzero_amd.packing import pack_network, pack_network_mfma

SEED = int(sys.argv[1]) if len(sys.argv) > 1 else 1
TRIALS = int(sys.argv[2]) if len(sys.argv) > 2 else 40
ORDER = sys.argv[3].split(',') if len(sys.argv) > 3 else ['torch', 'mfma', 'mfma_level', 'mfma_fused', 'mfma_search']
rng = np.random.default_rng(SEED)
bad = 0
for trial in range(TRIALS):
    gumbel = bool(rng.integers(2)); S = int(rng.integers(1, 40)); players = int(rng.integers(1, 3))
    gamma = float(rng.choice([1.0, 0.997])); B = int(rng.integers(1, 400))
    L = int(rng.choice([1, 5, 10, 30, 100]))
    if rng.integers(3) == 0:
        side = int(rng.integers(3, 5)); A = int(rng.integers(2, min(12, side * side + 1))); ch = int(rng.choice([8, 16, 24]))
        red = int(rng.choice([2, 4, 8])); fc = int(rng.choice([8, 16, 32]))
        ncfg = SimpleNamespace(network='resnet', observation_dim=[3, side, side], n_actions=A, n_stacked_observations=1,
                               blocks=int(rng.integers(1, 3)), channels=ch, reduced_channels_reward=red,
                               reduced_channels_policy=red, reduced_channels_value=red, resnet_fc_reward_layers=[fc],
                               resnet_fc_policy_layers=[fc], resnet_fc_value_layers=[fc], downsample=False, support_limit=L)
        obs_shape = (3, side, side)
    else:
        A = int(rng.integers(2, 24)); obs_n = int(rng.integers(2, 40)); enc = int(rng.choice([4, 8, 16, 24, 48, 64, 96]))
        lay = lambda: [int(rng.choice([8, 16, 32, 64, 100]))] * int(rng.integers(0, 3))  # noqa: E731
        ncfg = SimpleNamespace(network='mlp', observation_dim=[1, 1, obs_n], n_actions=A, n_stacked_observations=1,
                               encoding_size=enc, fc_reward_layers=lay() or [16], fc_policy_layers=lay() or [16],
                               fc_value_layers=lay() or [16], fc_representation_layers=lay(),
                               fc_hidden_state_layers=lay() or [16], support_limit=L)
        obs_shape = (1, 1, obs_n)
    desc = dict(net=ncfg.network, gumbel=gumbel, A=A, S=S, B=B, L=L, obs=obs_shape)
    net = MuZeroNetwork(ncfg, seed=trial)
    gnet = MuZeroNetwork(ncfg, net.state_dict()).to('cuda')
    pm, io = pack_network_mfma(net)
    cfg = make_config(gumbel, A, S, players, gamma)
    obs = rng.uniform(-1, 1, size=(B,) + obs_shape).astype(np.float32)
    logits, hidden, value = net.initial_inference(torch.as_tensor(obs))
    logits, hidden, value = logits.numpy(), hidden.numpy().reshape(B, -1), value.numpy()
    legal = np.ones((B, A)); tp = rng.integers(0, players, size=B).astype(np.int32)
    g = rng.gumbel(size=(B, A)) if gumbel else None
    di = None if gumbel else rng.dirichlet([0.25] * A, size=B)
    ok, results = True, {}
    # (0) the initial inference on the device (representation program, hidden-state scaling, prediction program,
    # support_to_scalar) against the same network run by torch
    try:
        from muzero_amd.packing import pack_root_network_mfma
        eng = SearchEngine(cfg, net.H, B, seed=trial)
        eng.set_root_network_mfma(*pack_root_network_mfma(net))
        rl, rh, rv = eng.initial_inference_mfma(obs.reshape(B, -1))
        errs = (np.abs(rl.cpu().numpy() - logits).max(), np.abs(rh.cpu().numpy() - hidden).max() / 10,
                np.abs(rv.cpu().numpy() - value).max() / max(1.0, L / 10))
        eng.close()
        if not max(errs) <= 1e-5:
            ok = False
            print('ROOT INFERENCE DIFFERS trial', trial, errs, desc, flush=True)
    except ValueError as e:
        print('  root inference not served:', e, flush=True)
    for mode in ORDER:
        print('  trial', trial, mode, desc, flush=True)
        os.environ['MZ_NET_PATH'] = 'level' if mode == 'mfma_level' else 'tile'
        eng = SearchEngine(cfg, net.H, B, seed=trial)
        flag = None
        try:
            eng.set_network_mfma(pm, io)
            prior = eng.root_prepare(logits, legal, di, 0.25)
            eng.reset(prior, value, hidden, tp, legal, g)
            if mode == 'mfma_fused':
                eng.select(want_hidden=False)
                for sim in range(S):
                    eng.expand_backup_net(select_next=True)
            elif mode == 'mfma_search':
                eng.search_mfma()
            else:
                for sim in range(S):
                    parent, action, phid = eng.select()
                    # checked without a host synchronisation (flag read after the search); the clamp keeps a
                    # wrong index from faulting torch's one-hot scatter
                    oob = ((action < 0) | (action >= A) | (parent < 0) | (parent > sim)).any()
                    flag = oob if flag is None else flag | oob
                    action = action.clamp(0, A - 1)
                    if mode == 'torch' or sim == 0:
                        lg, nh, val, rew = gnet.recurrent_flat(phid, gnet.encode_actions(action))
                    if mode == 'torch':
                        eng.expand_backup(lg, val, rew, nh)
                        continue
                    eng.expand_backup_net()
                    if sim == 0:
                        for b in {0, B // 2, B - 1}:
                            t = eng.export_tree(b)
                            errs = (np.abs(t['children_prior_logits'][1] - lg[b].cpu().numpy()).max(),
                                    abs(t['node_values'][1] - float(val[b])),
                                    abs(t['children_rewards'][0][int(action[b])] - float(rew[b])),
                                    np.abs(t['hidden_states'][1] - nh[b].cpu().numpy()).max() / 10)
                            if not max(errs) <= 1e-5:
                                ok = False
                                print('STEP DIFFERS trial', trial, mode, b, errs, desc, flush=True)
        except ValueError as e:        # shapes a path does not serve (over 16 actions, tile over LDS)
            eng.close()
            continue
        if flag is not None and bool(flag):
            ok = False
            print('SELECTION OUT OF RANGE trial', trial, mode, desc, flush=True)
        a, v, p, _ = eng.to_act(0.0)
        nv, _ = eng.stats()
        results[mode] = (a.cpu().numpy().copy(), v.cpu().numpy().copy(), nv.cpu().numpy().copy())
        if mode in ('mfma', 'mfma_level') and 'trees' not in results:
            results['trees'] = [eng.export_tree(b) for b in range(B)]
        eng.close()
    base = results.get('mfma') or results.get('mfma_level')
    if base is None:
        print('no matrix-core path ran: trial', trial, desc, flush=True)
        bad += 1
        continue
    for k in ('mfma_fused', 'mfma_search', 'mfma_level'):
        if k not in results or results[k] is base: continue
        # over 16 actions, or when a 32-tree tile of activations does not fit in LDS, the whole search runs level by
        # level: the very search of the layer-by-layer mode
        search_by_level = A > 16 or pm.stride * 32 * 4 + 32 * 16 * 8 > 160 * 1024
        if k == 'mfma_search' and search_by_level and 'mfma_level' in results:
            if not all(np.array_equal(results['mfma_level'][i], results[k][i]) for i in range(3)):
                ok = False
                print('LEVEL MODES DIFFER trial', trial, desc, flush=True)
        if k in ('mfma_level', 'mfma_search') and 2 * L + 1 > 64 and (k == 'mfma_level' or search_by_level):
            same = (base[0] == results[k][0]).mean() >= 0.97 and (np.abs(base[1] - results[k][1]) < 1e-5).mean() >= 0.97
        else:
            same = all(np.array_equal(base[i], results[k][i]) for i in range(3))
        if not same:
            ok = False
            print('MODES DIFFER trial', trial, k, desc, flush=True)
    fa = (base[0] == results['torch'][0]).mean(); fv = (np.abs(base[1] - results['torch'][1]) < 1e-5).mean()
    if fa < 0.95 or fv < 0.95:
        # Allowed only with an explanation: given its own leaf outputs the matrix-core search is exactly the checker's,
        # and every one of those outputs is within tolerance of the checker's dense executor on identical inputs -
        # then the disagreement is float32 drift along deep chains of recurrent steps, not a defect.
        from oracle import oracle as orc
        packed = pack_network(net)
        dvv = np.abs(base[1] - results['torch'][1])
        explained = True
        for b in np.argsort(-dvv)[:8]:
            t = results['trees'][int(b)]
            try:
                o = replay_on_checker(cfg, t, net.H, legal[b], int(tp[b]), seed=trial, tree_id=int(b),
                                      gumbel_noise=None if g is None else g[b])
                assert_tree_equal(o, t, hidden=False)
            except AssertionError as e:
                explained = False
                print('   replay differs, tree', int(b), str(e)[:200], flush=True)
                continue
            par, act = t['parents'][1:], t['action_from_parent'][1:]
            lg2, nh2, val2, rew2 = orc.net_recurrent_batch(packed, t['hidden_states'][par].reshape(S, -1), act)
            tol = 1e-5 if L <= 32 else 5e-5
            errs = (np.abs(lg2 - t['children_prior_logits'][1:]).max(), np.abs(val2 - t['node_raw_values'][1:]).max(),
                    np.abs(rew2 - t['children_rewards'][par, act]).max())
            if not (errs[0] <= 1e-5 and errs[1] <= tol and errs[2] <= tol):
                explained = False
                print('   leaf outputs off, tree', int(b), errs, flush=True)
        ok = ok and explained
        dv = np.abs(base[1] - results['torch'][1])
        print('   explained by float32 drift:' if explained else '   NOT explained:', end=' ')
        print('TORCH DISAGREES trial', trial, 'actions %.3f values %.3f' % (fa, fv), 'largest value difference %.3g, median %.3g,' % (dv.max(), np.median(dv)),
              'mean leaf depth %.1f' % (base[2].mean() / S - 1), desc, flush=True)
    bad += not ok
    if trial % 5 == 0: print('trial', trial, desc, 'modes', sorted(k for k in results if k != 'trees'), flush=True)
print('done, failing trials:', bad)
