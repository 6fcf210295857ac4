"""Soak test (not part of the suite): the cached-decision Gumbel kernel (certified cheap argmax, work list, staged record
fetch) against the kernel that scores every node with the reference's operations on the way down.  Random action counts
(2..16: 3x3-board ResNets with 16 channels for 9 actions half of the time, small MLPs otherwise), records with 16- or
32-bit visit / index rows, policy heads scaled from nearly flat to sharply peaked so that leads of every size occur,
random simulations / players / discount / legality / batch and tile sizes.  Every root statistic of every tree and a
sample of whole trees must agree bit for bit.

    python scripts/soak_cached.py [seed=1] [trials=30]
"""
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from helpers import make_config
from muzero_amd.engine import SearchEngine
from muzero_amd.network import MuZeroNetwork
from muzero_amd.packing import pack_network_mfma

SEED = int(sys.argv[1]) if len(sys.argv) > 1 else 1
TRIALS = int(sys.argv[2]) if len(sys.argv) > 2 else 30
rng = np.random.default_rng(SEED)


def run(cached, tm, cfg, net, pm_io, logits, hidden, value, legal, tp, g, B):
    os.environ['MZ_SEARCH_CACHED'] = '2' if cached else '0'   # 2: an error if the kernel does not serve the case
    os.environ['MZ_SEARCH_TM'] = str(tm)
    eng = SearchEngine(cfg, net.H, B, seed=2)
    eng.set_network_mfma(*pm_io)
    prior = eng.root_prepare(logits, legal, None, 0.0)
    eng.reset(prior, value, hidden, tp, legal, g)
    eng.search_mfma()
    action, rv, probs, q = eng.to_act(0.0)
    nv, _ = eng.stats()
    out = dict(action=action.cpu().numpy().copy(), value=rv.cpu().numpy().copy(), probs=probs.cpu().numpy().copy(),
               q=q.cpu().numpy().copy(), nv=nv.cpu().numpy().copy(),
               trees=[eng.export_tree(b) for b in sorted({int(x) for x in np.linspace(0, B - 1, 6)})])
    eng.close()
    return out


bad = 0
for trial in range(TRIALS):
    S = int(rng.integers(1, 101)); players = int(rng.integers(1, 3)); gamma = float(rng.choice([1.0, 0.997]))
    B = int(rng.integers(1, 300)); L = int(rng.choice([1, 5, 10, 30]))
    red = int(rng.choice([2, 4, 8])); fc = int(rng.choice([8, 16, 32]))
    A = int(rng.integers(2, 17))
    if rng.random() < 0.3:
        A = 9
    board = A == 9 and rng.random() < 0.5
    if rng.random() < 0.4:
        os.environ['MZ_IDX16'] = '0'          # 32-bit rows even where 16 bits would shorten the record
    else:
        os.environ.pop('MZ_IDX16', None)
    if board:
        ncfg = SimpleNamespace(network='resnet', observation_dim=[3, 3, 3], n_actions=9, n_stacked_observations=1,
                               blocks=int(rng.integers(1, 3)), channels=16, reduced_channels_reward=red,
                               reduced_channels_policy=red, reduced_channels_value=red, resnet_fc_reward_layers=[fc],
                               resnet_fc_policy_layers=[fc], resnet_fc_value_layers=[fc], downsample=False, support_limit=L)
    else:
        ncfg = SimpleNamespace(network='mlp', observation_dim=[1, 1, 27], n_actions=A, n_stacked_observations=1,
                               encoding_size=int(rng.choice([8, 24, 40])), fc_reward_layers=[fc], fc_policy_layers=[fc],
                               fc_value_layers=[fc], fc_representation_layers=[fc], fc_hidden_state_layers=[2 * fc],
                               support_limit=L, blocks=0)
    net0 = MuZeroNetwork(ncfg, seed=1000 * SEED + trial)
    sd = {k: v.clone() for k, v in net0.state_dict().items()}
    scale = float(rng.choice([0.0, 1e-6, 1e-3, 1.0, 1.0, 10.0, 100.0]))     # flat ... peaked policy heads
    for k in [k for k in sd if 'policy' in k and (k.endswith('weight') or k.endswith('bias'))][-2:]:
        sd[k] *= scale
    net = MuZeroNetwork(ncfg, sd).to('cuda')
    pm_io = pack_network_mfma(net)
    cfg = make_config(True, A, S, players, gamma)
    obs = rng.uniform(-1, 1, size=(B, 27)).astype(np.float32)
    with torch.no_grad():
        logits, hidden, value = net.initial_flat(torch.as_tensor(obs, device='cuda'))
    legal = np.ones((B, A))
    for b in range(0, B, 3):
        legal[b, rng.choice(A, size=rng.integers(1, A), replace=False)] = 0
    tp = rng.integers(0, players, size=B).astype(np.int32)
    g = rng.gumbel(size=(B, A))
    want = run(False, 32, cfg, net, pm_io, logits, hidden, value, legal, tp, g, B)
    ok = True
    for tm in (32, 16):
        try:
            got = run(True, tm, cfg, net, pm_io, logits, hidden, value, legal, tp, g, B)
        except (ValueError, RuntimeError) as e:
            print(f'trial {trial} tm={tm}: not served ({e})')
            continue
        for k in ('action', 'value', 'probs', 'q', 'nv'):
            if not np.array_equal(want[k], got[k], equal_nan=True):
                ok = False; print(f'trial {trial} tm={tm}: {k} differs on {int((want[k] != got[k]).sum())} entries')
        for tw, tg in zip(want['trees'], got['trees']):
            for k in tw:
                if not np.array_equal(tw[k], tg[k], equal_nan=True):
                    ok = False; print(f'trial {trial} tm={tm}: tree field {k} differs')
    bad += not ok
    print(f'trial {trial}: A={A} rows={"32" if os.environ.get("MZ_IDX16") else "auto"} {"board" if board else "mlp"} S={S} B={B} players={players} L={L} blocks={ncfg.blocks} policy scale {scale:g} '
          f'mean visits per simulation {want["nv"].mean() / max(S, 1):.1f}  {"ok" if ok else "MISMATCH"}', flush=True)
print(f'{TRIALS - bad} of {TRIALS} configurations bit-identical')
sys.exit(1 if bad else 0)
