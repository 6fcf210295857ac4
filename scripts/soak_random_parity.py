"""Soak test (not part of the suite): random search configurations against the CPU checker, bit for bit.

    python scripts/soak_random_parity.py [seed=777] [trials=400]
"""
import sys, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from helpers import make_config
from muzero_amd import hashnet
from muzero_amd.engine import SearchEngine
from oracle import oracle as orc
SEED = int(sys.argv[1]) if len(sys.argv) > 1 else 777
TRIALS = int(sys.argv[2]) if len(sys.argv) > 2 else 400
rng = np.random.default_rng(SEED)
bad = 0
for trial in range(TRIALS):
    gumbel = bool(rng.integers(2)); A = int(rng.integers(1, 65)); S = int(rng.integers(1, 120)); players = int(rng.integers(1, 3))
    gamma = float(rng.choice([1.0, 0.997, 0.9, 0.5])); H = int(rng.integers(2, 12)); B = int(rng.integers(1, 200))
    cfg = make_config(gumbel, A, S, players, gamma)
    if gumbel: cfg.max_considered_actions = int(rng.choice([1, 2, 4, 16, 64]))
    hp = hashnet.HashNetParams(A, H, int(rng.integers(1, 8)), int(rng.integers(2)), int(rng.integers(2)))
    legal = np.ones((B, A))
    if A > 2 and rng.integers(2):
        for b in range(0, B, 2):
            k = rng.integers(1, A - 1); legal[b, rng.choice(A, size=k, replace=False)] = 0
    codes = rng.integers(0, 2**32, size=B, dtype=np.uint64).astype(np.uint32)
    to_play = rng.integers(0, players, size=B).astype(np.int32)
    g = rng.gumbel(size=(B, A)) if gumbel else None
    di = None if gumbel else rng.dirichlet([0.25] * A, size=B)
    ref = orc.search_batch_hash(cfg, hp, H, codes, legal, to_play, dirichlet=di, frac=0.25, gumbel=g, seed=trial, tree_id0=trial, n_threads=8)
    rl = 2
    while rl < A: rl *= 2
    lanes = int(rng.choice(sorted({rl, max(rl, 16), 64})))
    eng = SearchEngine(cfg, H, B, seed=trial, lanes_per_tree=lanes)
    logits = np.stack([hashnet.logits_of(int(c), hp) for c in codes])
    prior = eng.root_prepare(logits, legal, di, 0.25)
    eng.reset(prior, [hashnet.value_of(int(c)) for c in codes], np.stack([hashnet.hidden_of(int(c), hp) for c in codes]), to_play, legal, g, tree_id0=trial)
    eng.search_hash(hp.logit_shift, hp.reward_on, hp.parent_predicts)
    action, value, probs, _ = eng.to_act(0.0)
    nv, _ = eng.stats()
    ok = np.array_equal(nv.cpu().numpy(), ref.node_visits) and np.array_equal(value.cpu().numpy(), ref.root_value) and np.array_equal(action.cpu().numpy(), ref.action)
    if not ok:
        bad += 1
        print('MISMATCH', trial, gumbel, A, S, players, gamma, H, B, lanes, int((nv.cpu().numpy() != ref.node_visits).sum()), flush=True)
    eng.close()
    if trial % 100 == 0: print('trial', trial, flush=True)
print('done, mismatches:', bad)
