"""Experiment: the cached-decision Gumbel search against the search that scores every node on the way down, for
action counts other than 9 (small MLP networks).  Prints ms per search for both.

    python scripts/exp_cached_speed.py [trees=4096] [sims=50] [policy head scale=1: larger = deeper trees]
"""
import os
import sys
import time
from types import SimpleNamespace

import numpy as np
import torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from helpers import make_config
from muzero_amd.engine import SearchEngine
from muzero_amd.network import MuZeroNetwork
from muzero_amd.packing import pack_network_mfma

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
S = int(sys.argv[2]) if len(sys.argv) > 2 else 50
SCALE = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
rng = np.random.default_rng(0)
for A, rows in ((2, 'auto'), (4, 'auto'), (6, 'auto'), (9, 'auto'), (12, 'auto'), (16, 'auto'), (16, '32')):
    if rows == '32':
        os.environ['MZ_IDX16'] = '0'
    else:
        os.environ.pop('MZ_IDX16', None)
    shape = SimpleNamespace(network='mlp', observation_dim=[1, 1, 32], n_actions=A, n_stacked_observations=1,
                            encoding_size=64, fc_reward_layers=[64], fc_policy_layers=[64], fc_value_layers=[64],
                            fc_representation_layers=[64], fc_hidden_state_layers=[64], support_limit=10)
    net = MuZeroNetwork(shape, seed=A)
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    for k in [k for k in sd if 'policy' in k and (k.endswith('weight') or k.endswith('bias'))][-2:]:
        sd[k] *= SCALE
    net = MuZeroNetwork(shape, sd).to('cuda')
    pm_io = pack_network_mfma(net)
    cfg = make_config(True, A, S, 2, 0.997)
    obs = torch.as_tensor(rng.standard_normal((B, 32)).astype(np.float32), device='cuda')
    with torch.no_grad():
        logits, hidden, value = net.initial_flat(obs)
    legal, tp, g = np.ones((B, A)), np.zeros(B, np.int32), rng.gumbel(size=(B, A))
    res = {}
    for cached in (1, 0):
        os.environ['MZ_SEARCH_CACHED'] = str(cached)
        eng = SearchEngine(cfg, net.H, B, seed=2)
        eng.set_network_mfma(*pm_io)
        prior = eng.root_prepare(logits, legal, None, 0.0)
        ts = []
        for rep in range(6):
            eng.reset(prior, value, hidden, tp, legal, g)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            eng.search_mfma()
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        nv, _ = eng.stats()
        res[cached] = (min(ts[1:]) * 1e3, float(nv.sum().item()))
        eng.close()
    assert res[0][1] == res[1][1]
    print(f'A={A:2d} rows={rows:4s} {B} trees x {S} sims: cached {res[1][0]:.3f} ms, scored on the way down {res[0][0]:.3f} ms '
          f'(mean depth {res[1][1] / (B * S) - 1:.1f})', flush=True)
