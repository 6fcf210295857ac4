"""Soak test (not part of the suite): the level-by-level PUCT search over cached decisions (tree_ops.h puct_decide) against
the same search with the table switched off (MZ_PUCT_CACHE=0), which scores every node of every descent with that call's
tie-break draws.  Random action counts (2..40; MZ_SEARCH_LEVELS=1 sends the small ones down the level-by-level path
too), MLP shapes, simulations (1..300: trees within and beyond the LDS copy of the table), players, discount, legality,
batch sizes, noise source (counter / caller's buffer), policy heads scaled from flat (exact ties everywhere) to sharply
peaked (chains), searches resumed after stepped simulations.  Trees, selection-call counts and root statistics must
agree bit for bit.

    python scripts/soak_puct_cache.py [seed=1] [trials=40]
"""
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from helpers import make_config
from muzero_amd import engine as E
from muzero_amd.network import MuZeroNetwork
from muzero_amd.packing import pack_network_mfma

SEED = int(sys.argv[1]) if len(sys.argv) > 1 else 1
TRIALS = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rng = np.random.default_rng(SEED)
os.environ['MZ_SEARCH_LEVELS'] = '1'
os.environ['MZ_SEARCH_CHAIN'] = '0'


def run(cached, cfg, net, gnet, pm_io, roots, legal, tp, di, B, S, A, buffer, stepped, seed):
    os.environ['MZ_PUCT_CACHE'] = '1' if cached else '0'
    eng = E.SearchEngine(cfg, net.H, B, seed=seed, noise_mode=E.NOISE_BUFFER if buffer else E.NOISE_COUNTER)
    eng.set_network_mfma(*pm_io)
    calls_max = (S + 1) * (S + 2) // 2
    noise = np.random.default_rng(seed).random((B, calls_max, A)) if buffer else None
    logits, hidden, value = roots
    eng.reset(eng.root_prepare(logits, legal, di, 0.25), value, hidden, tp, legal, None, tree_id0=0)
    if stepped:
        eng.select(noise, want_hidden=False)
        for i in range(stepped):
            eng.expand_backup_net(select_next=i + 1 < stepped, noise=noise)
    eng.search_mfma(noise)
    eng.check()
    nv, calls = eng.stats()
    act = eng.to_act(0.0)
    out = dict(nv=nv.cpu().numpy(), calls=calls.cpu().numpy(), act=[x.cpu().numpy() for x in act],
               trees=[eng.export_tree(b) for b in sorted({int(x) for x in np.linspace(0, B - 1, 5)})])
    eng.close()
    return out


bad = 0
for trial in range(TRIALS):
    A = int(rng.integers(2, 41)); S = int(rng.choice([1, 7, 30, 60, 120, 300])); B = int(rng.integers(1, 200))
    if S == 300:
        B = min(B, 24)
    players = int(rng.integers(1, 3)); gamma = float(rng.choice([1.0, 0.997, 0.9])); L = int(rng.choice([1, 10, 40]))
    fc = int(rng.choice([16, 32, 48])); buffer = bool(rng.random() < 0.3) and S <= 60; stepped = int(rng.choice([0, 0, 3])) if S >= 7 else 0   # (the engine holds S simulations per tree: stepped ones count)
    shape = SimpleNamespace(network='mlp', observation_dim=[1, 1, 20], n_actions=A, n_stacked_observations=1,
                            encoding_size=int(rng.choice([16, 24, 40])), fc_reward_layers=[fc], fc_policy_layers=[fc],
                            fc_value_layers=[fc], fc_representation_layers=[fc], fc_hidden_state_layers=[fc], support_limit=L)
    net0 = MuZeroNetwork(shape, seed=100 * SEED + trial)
    sd = {k: v.clone() for k, v in net0.state_dict().items()}
    scale = float(rng.choice([0.0, 1e-6, 1.0, 1.0, 8.0, 40.0]))
    for k in sorted(k for k in sd if 'policy_network' in k)[-2:]:
        sd[k] *= scale
    if rng.random() < 0.4:     # one preferred action: chains
        sd[sorted(k for k in sd if 'policy_network' in k and k.endswith('bias'))[-1]][int(rng.integers(A))] += 6.0
    net = MuZeroNetwork(shape, sd)
    gnet = MuZeroNetwork(shape, sd).to('cuda')
    pm_io = pack_network_mfma(gnet)
    cfg = make_config(False, A, S, players, gamma)
    obs = torch.as_tensor(rng.standard_normal((B, 20)).astype(np.float32), device='cuda')
    with torch.no_grad():
        roots = gnet.initial_flat(obs)
    legal = np.ones((B, A))
    for b in range(0, B, 3):
        legal[b, rng.choice(A, size=rng.integers(1, A), replace=False)] = 0
    tp = rng.integers(0, players, size=B).astype(np.int32)
    di = rng.dirichlet([0.25] * A, size=B)
    want = run(False, cfg, net, gnet, pm_io, roots, legal, tp, di, B, S, A, buffer, stepped, trial)
    got = run(True, cfg, net, gnet, pm_io, roots, legal, tp, di, B, S, A, buffer, stepped, trial)
    ok = np.array_equal(want['nv'], got['nv']) and np.array_equal(want['calls'], got['calls'])
    ok = ok and all(np.array_equal(x, y, equal_nan=True) for x, y in zip(want['act'], got['act']))
    for tw, tg in zip(want['trees'], got['trees']):
        for k in tw:
            if not np.array_equal(tw[k], tg[k], equal_nan=True):
                ok = False; print(f'trial {trial}: tree field {k} differs')
    bad += not ok
    print(f'trial {trial}: A={A} S={S} B={B} players={players} gamma={gamma} {"buffer" if buffer else "counter"} noise, '
          f'{stepped} stepped, policy scale {scale:g}, mean depth {want["nv"].mean() / max(S, 1) - 1:.1f}  '
          f'{"ok" if ok else "MISMATCH"}', flush=True)
print(f'{TRIALS - bad} of {TRIALS} configurations bit-identical')
sys.exit(1 if bad else 0)
