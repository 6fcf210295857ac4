"""Soak test (not part of the suite): the whole-search kernel with in-kernel leaf evaluation on random MLP
networks.  (1) Trees kept in LDS (with all-nodes selection for 2 actions, register-cached rows, the wave
epilogue) or in registers must equal the global-memory path bit for bit; (2) every group width must agree with the CPU
checker's search through the same lowered network on (nearly) every tree.

    python scripts/soak_fused_net.py [seed=1] [trials=60]
"""
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from helpers import make_config
from muzero_amd.engine import SearchEngine, NOISE_BUFFER, NOISE_COUNTER
from muzero_amd.network import MuZeroNetwork
from muzero_amd.packing import pack_network
from oracle import oracle as orc

SEED = int(sys.argv[1]) if len(sys.argv) > 1 else 1
TRIALS = int(sys.argv[2]) if len(sys.argv) > 2 else 60
rng = np.random.default_rng(SEED)
bad = 0
for trial in range(TRIALS):
    gumbel = bool(rng.integers(2)); A = int(rng.choice([2, 2, 2, 3, 4, 6])); S = int(rng.integers(1, 70)); players = int(rng.integers(1, 3))
    gamma = float(rng.choice([1.0, 0.997])); B = int(rng.integers(1, 300))
    obs_n = int(rng.integers(2, 9)); enc = int(rng.choice([2, 4, 6, 8, 8, 12, 16, 24])); L = int(rng.choice([1, 5, 10, 20]))
    lay = lambda: [int(rng.choice([8, 16, 32]))] * int(rng.integers(0, 2))  # noqa: E731
    ncfg = SimpleNamespace(network='mlp', observation_dim=[1, 1, obs_n], n_actions=A, n_stacked_observations=1, encoding_size=enc,
                           fc_reward_layers=lay() or [16], fc_policy_layers=lay() or [16], fc_value_layers=lay() or [16],
                           fc_representation_layers=lay(), fc_hidden_state_layers=lay() or [16], support_limit=L)
    net = MuZeroNetwork(ncfg, seed=trial)
    packed = pack_network(net)
    cfg = make_config(gumbel, A, S, players, gamma)
    obs = rng.uniform(-1, 1, size=(B, 1, 1, obs_n)).astype(np.float32)
    logits, hidden, value = net.initial_inference(torch.as_tensor(obs))
    logits, hidden, value = logits.numpy(), hidden.numpy().reshape(B, -1), value.numpy()
    legal = np.ones((B, A))
    if rng.integers(2): legal[np.arange(B), rng.integers(0, A, size=B)] = rng.integers(0, 2, size=B)   # some roots with one illegal action
    tp = rng.integers(0, players, size=B).astype(np.int32)
    g = rng.gumbel(size=(B, A)) if gumbel else None
    di = None if gumbel else rng.dirichlet([0.25] * A, size=B)
    # tie-break noise from the counter generator (the checker's) or from explicit rows (paths compared with each other)
    use_buffer = (not gumbel) and rng.integers(4) == 0
    rows = rng.random((B, S * (S + 1) // 2 + S, A)) if use_buffer else None
    stepped = int(rng.integers(0, min(S, 6))) if rng.integers(3) == 0 else 0   # simulations done by the step API first
    gnet = MuZeroNetwork(ncfg, net.state_dict()).to('cuda') if stepped else None
    ref = orc.search_batch_net(cfg, packed, logits, hidden, value, legal, tp, dirichlet=di, frac=0.25, gumbel=g, seed=trial,
                               tree_id0=0, n_threads=8)
    rl = 2
    while rl < A: rl *= 2
    ok, runs = True, {}
    for lanes, lds, reg in [(64, '0', '0'), (64, '1', '0'), (64, '1', '1'), (64, '1', 'lds-hidden')] + [(w, '1', '0') for w in sorted({rl, 16}) if w >= rl]:
        os.environ['MZ_TREE_LDS'] = lds
        os.environ['MZ_TREE_REG'] = '0' if reg == '0' else '1'     # trees in registers (2-action PUCT trees of at most 64 nodes)
        os.environ['MZ_NET_TINY'] = '0' if reg == 'lds-hidden' else '1'   # ... with the hidden states in registers too
        eng = SearchEngine(cfg, packed.H, B, seed=trial, lanes_per_tree=lanes,
                           noise_mode=NOISE_BUFFER if use_buffer else NOISE_COUNTER)
        try:
            eng.set_network(packed)
            prior = eng.root_prepare(logits, legal, di, 0.25)
            eng.reset(prior, value, hidden, tp, legal, g)
            for _ in range(stepped):
                parent, action, phid = eng.select(noise=rows)
                lg, nh, val, rew = gnet.recurrent_flat(phid, gnet.encode_actions(action))
                eng.expand_backup(lg, val, rew, nh)
            eng.search_net(noise=rows)
        except ValueError as e:       # program too large for LDS at this width: not this script's subject
            eng.close()
            continue
        action, rv, probs, _ = eng.to_act(0.0)
        nv, calls = eng.stats()
        trees = [eng.export_tree(b) for b in range(0, B, max(1, B // 6))]
        runs[(lanes, lds, reg)] = (action.cpu().numpy().copy(), rv.cpu().numpy().copy(), nv.cpu().numpy().copy(), trees)
        frac_same = float((nv.cpu().numpy() == ref.node_visits).mean())
        close = float((np.abs(rv.cpu().numpy() - ref.root_value) < 1e-5).mean())
        if not use_buffer and not stepped and (frac_same < 0.97 or close < 0.97):
            ok = False
            print('CHECKER DISAGREES trial', trial, (lanes, lds, reg), 'same visits %.3f close values %.3f' % (frac_same, close), flush=True)
        eng.close()
    for other in ((64, '1', '0'), (64, '1', '1'), (64, '1', 'lds-hidden')):
        if (64, '0', '0') not in runs or other not in runs: continue
        a, b = runs[(64, '0', '0')], runs[other]
        same = all(np.array_equal(a[i], b[i]) for i in range(3))
        for ta, tb in zip(a[3], b[3]):
            same = same and all(np.array_equal(ta[k], tb[k]) for k in ta)
        if not same:
            ok = False
            print('LDS / REGISTER PATH DIFFERS trial', trial, other, dict(gumbel=gumbel, A=A, S=S, B=B, enc=enc, L=L), flush=True)
    bad += not ok
    if trial % 10 == 0: print('trial', trial, dict(gumbel=gumbel, A=A, S=S, B=B, enc=enc, L=L, widths=sorted(runs)), flush=True)
print('done, failing trials:', bad)
