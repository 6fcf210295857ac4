"""Soak test (not part of the suite): training targets built on the device (value / reward / policy
targets, stacked observations, action sequences' value targets, categorical supports) against the CPU
checker on random games: lengths from 1, bootstrap horizons longer and shorter than the games, unroll
windows that run off the end, one and two players, discounts, stacking with and without action planes.
Every sample must be equal bit for bit.

    python scripts/soak_targets.py [seed=1] [trials=60]
"""
import sys

import numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from test_targets_oracle import history_of
from muzero_amd.targets import TrajectoryStore, scalar_to_support
from oracle import oracle as orc

SEED = int(sys.argv[1]) if len(sys.argv) > 1 else 1
TRIALS = int(sys.argv[2]) if len(sys.argv) > 2 else 60
rng = np.random.default_rng(SEED)
bad = 0
for trial in range(TRIALS):
    A = int(rng.integers(1, 20)); G = int(rng.integers(1, 30)); players = int(rng.integers(1, 3))
    obs_shape = tuple(int(x) for x in rng.integers(1, 5, size=3))
    td = int(rng.integers(1, 60)); unroll = int(rng.integers(0, 12)); gamma = float(rng.choice([1.0, 0.997, 0.9, 0.5]))
    max_T = int(rng.choice([1, 3, 20, 300]))
    scale = float(rng.choice([1.0, 1e-3, 1e3]))
    games = []
    for g in range(G):
        T = int(rng.integers(1, max_T + 1))
        tp = (np.arange(T) % 2) if players == 2 and rng.integers(2) else rng.integers(0, players, size=T)
        games.append(dict(observations=rng.normal(size=(T,) + obs_shape), actions=rng.integers(0, A, size=T),
                          rewards=rng.normal(size=T) * scale, to_plays=tp, probs=rng.dirichlet([1] * A, size=T),
                          root_values=rng.normal(size=T) * scale))
    store = TrajectoryStore([history_of(d, A) for d in games], A)
    n = int(rng.integers(1, 700))
    sel = rng.integers(0, G, size=n)
    ts = np.array([rng.integers(0, len(games[g]['actions'])) for g in sel])
    ab = np.array([[rng.integers(0, len(games[g]['actions'])) for _ in range(unroll + 1)] for g in sel])
    n_stack = int(rng.integers(1, 5)); stack_action = bool(rng.integers(2))
    desc = dict(A=A, G=G, players=players, obs=obs_shape, td=td, unroll=unroll, gamma=gamma, max_T=max_T, n=n,
                n_stack=n_stack, stack_action=stack_action, scale=scale)
    obs, act, val, rew, pol = store.make_targets(sel, ts, ab, td, gamma, unroll, n_stack, stack_action)
    L = int(rng.choice([1, 5, 10, 300]))
    sup_v = scalar_to_support(val, L).cpu().numpy()
    val, rew, pol, obs = val.cpu().numpy(), rew.cpu().numpy(), pol.cpu().numpy(), obs.cpu().numpy()
    ok = True
    for i in range(n):
        d = games[sel[i]]
        v, r, p = orc.make_target(d['rewards'], d['to_plays'], d['root_values'], d['probs'], ts[i], td, gamma, unroll)
        o = orc.stack_observations(d['observations'], d['actions'], ts[i], n_stack, A, stack_action)
        same = (np.array_equal(val[i], v.astype(np.float32)) and np.array_equal(rew[i], r.astype(np.float32))
                and np.array_equal(pol[i], p.astype(np.float32))
                and np.array_equal(obs[i].reshape(-1), o.astype(np.float32).reshape(-1)))
        if not same:
            ok = False
            print('MISMATCH trial', trial, 'sample', i, 'game', int(sel[i]), 't', int(ts[i]), desc, flush=True)
            break
    if not np.array_equal(sup_v, orc.scalar_to_support(val, L)):
        ok = False
        print('SUPPORT MISMATCH trial', trial, 'L', L, desc, flush=True)
    bad += not ok
    if trial % 10 == 0: print('trial', trial, desc, flush=True)
print('done, failing trials:', bad)
