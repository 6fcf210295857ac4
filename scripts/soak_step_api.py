"""Soak test (not part of the suite): the step API (select / expand + backup per launch) with random
leaf outputs - quantised so that exact ties are common - against the CPU checker's single-tree search:
per-simulation (parent, action) choices and every field of the final trees, bit for bit.

    python scripts/soak_step_api.py [seed=5] [trials=150]
"""
import sys
import numpy as np
import torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from helpers import make_config
from muzero_amd.engine import SearchEngine, NOISE_BUFFER, NOISE_COUNTER
from oracle import oracle as orc

SEED = int(sys.argv[1]) if len(sys.argv) > 1 else 5
TRIALS = int(sys.argv[2]) if len(sys.argv) > 2 else 150
rng = np.random.default_rng(SEED)
FIELDS = ['node_visits', 'node_values', 'node_raw_values', 'parents', 'action_from_parent', 'to_play', 'children_index',
          'children_prior_logits', 'children_visits', 'children_rewards', 'children_values']
bad = 0
for trial in range(TRIALS):
    gumbel = bool(rng.integers(2)); A = int(rng.integers(1, 33)); S = int(rng.integers(1, 45)); players = int(rng.integers(1, 3))
    gamma = float(rng.choice([1.0, 0.997, 0.5])); B = int(rng.integers(1, 24))
    cfg = make_config(gumbel, A, S, players, gamma)
    if gumbel: cfg.max_considered_actions = int(rng.choice([2, 4, 16]))
    q = float(rng.choice([0.0, 0.25, 1.0]))             # quantisation step (0 = continuous)
    def draw(shape, scale):
        x = rng.normal(size=shape) * scale
        return np.round(x / q) * q if q else x
    logits = draw((B, S, A), float(rng.choice([0.5, 3.0, 20.0]))).astype(np.float32)
    value = draw((B, S), 1.0); reward = draw((B, S), 1.0) * float(rng.integers(2))
    root_logits = draw((B, A), float(rng.choice([2.0, 60.0]))).astype(np.float32); root_value = draw((B,), 1.0)
    legal = np.ones((B, A))
    if A > 2 and rng.integers(2):
        for b in range(B):
            legal[b, rng.choice(A, size=int(rng.integers(1, A - 1)), replace=False)] = 0
    to_play = rng.integers(0, players, size=B).astype(np.int32)
    g = rng.gumbel(size=(B, A)) if gumbel else None
    di = None if gumbel else rng.dirichlet([0.25] * A, size=B)
    use_buffer = (not gumbel) and bool(rng.integers(2))
    max_calls = S * (S + 1) // 2
    noise = rng.random((B, max_calls, A)) if use_buffer else None
    if use_buffer and rng.integers(2):
        noise = np.round(noise * 4) / 4                 # ties in the tie-break noise as well
    rl = 2
    while rl < A: rl *= 2
    lanes = int(rng.choice(sorted({rl, max(rl, 16), 64})))
    eng = SearchEngine(cfg, 0, B, seed=trial, lanes_per_tree=lanes, noise_mode=NOISE_BUFFER if use_buffer else NOISE_COUNTER)
    prior = eng.root_prepare(root_logits, legal, di, 0.25)
    eng.reset(prior, root_value, None, to_play, legal, g, tree_id0=100 * trial)
    nz = None if noise is None else torch.as_tensor(noise, device=eng.device)
    seq = np.zeros((B, S, 2), np.int64)
    for s in range(S):
        parent, action, _ = eng.select(nz)
        seq[:, s, 0], seq[:, s, 1] = parent.cpu().numpy(), action.cpu().numpy()
        eng.expand_backup(logits[:, s], value[:, s], reward[:, s], None)
    prior_h = prior.cpu().numpy()
    ok = True
    T = float(rng.choice([0.0, 0.5, 1.0]))
    uu = rng.random(B)
    act_e, val_e, probs_e, q_e = (x.cpu().numpy() for x in eng.to_act(T, uu))
    for b in range(B):   # root preparation itself (float64 log may differ from libm in the last bit; NaN where the reference has NaN)
        want = orc.gumbel_root_prior(root_logits[b], legal[b]) if gumbel else orc.puct_root_prior(root_logits[b], di[b], 0.25, legal[b])
        fin = np.isfinite(want) & (legal[b] != 0)
        tol = 0 if gumbel else 4e-16 * max(1.0, float(np.abs(want[fin]).max()) if fin.any() else 1.0)
        if not (np.array_equal(np.isnan(want), np.isnan(prior_h[b])) and np.allclose(prior_h[b][fin], want[fin], rtol=0, atol=tol)
                and np.array_equal(prior_h[b][legal[b] == 0], want[legal[b] == 0])):
            ok = False
            print('ROOT PRIOR MISMATCH trial', trial, 'tree', b, flush=True)
            break
    for b in range(B):
        o = orc.Oracle(cfg, 0)
        o.init(prior_h[b], root_value[b], None, to_play[b], legal[b], None if g is None else g[b])
        nzo = o.buffer_noise(noise[b]) if use_buffer else o.counter_noise(trial, 100 * trial + b)
        r = o.search(o.table_net(logits[b], value[b], reward[b]), nzo)
        t = eng.export_tree(b)
        same = np.array_equal(r.parent, seq[b, :, 0]) and np.array_equal(r.action, seq[b, :, 1])
        for f in FIELDS:
            same = same and np.array_equal(np.asarray(t[f]).reshape(o.arrays[f].shape), o.arrays[f], equal_nan=True)
        a_o, p_o, q_o = o.to_act(T, uu[b])
        same = same and int(act_e[b]) == a_o and float(val_e[b]) == float(o.arrays['node_values'][0])
        same = same and np.allclose(probs_e[b], p_o, rtol=1e-12, atol=1e-300, equal_nan=True)
        same = same and np.allclose(q_e[b], q_o, rtol=1e-12, atol=1e-300, equal_nan=True)
        if not same:
            ok = False
            first = int(np.nonzero((r.parent != seq[b, :, 0]) | (r.action != seq[b, :, 1]))[0][0]) if not (np.array_equal(r.parent, seq[b, :, 0]) and np.array_equal(r.action, seq[b, :, 1])) else -1
            print('MISMATCH trial', trial, 'tree', b, dict(gumbel=gumbel, A=A, S=S, players=players, gamma=gamma, q=q, lanes=lanes, buffer=use_buffer), 'first differing sim', first, flush=True)
            break
    bad += not ok
    eng.close()
    if trial % 25 == 0: print('trial', trial, flush=True)
print('done, mismatching trials:', bad)
