"""Soak test for the CPU checker itself (build container only: it imports the reference from
/root/reference): random search configurations with random, quantised network outputs - exact ties are
common - run through the reference's MCTS.search and through the checker; every Tree field, the action,
the root value and the policy must agree (policy at rtol 1e-12, everything else bit for bit).

    python scripts/soak_checker_vs_reference.py [seed=3] [trials=300]
"""
import sys, types, warnings
import numpy as np
import torch
warnings.simplefilter('ignore')
sys.dont_write_bytecode = True
sys.modules.setdefault('pygraphviz', types.ModuleType('pygraphviz'))
sys.path.insert(0, '/root/reference')
from mcts.mcts import MCTS  # noqa: E402  (reference)
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from helpers import make_config  # noqa: E402
from muzero_amd import noise as mznoise  # noqa: E402
from oracle import oracle as orc  # noqa: E402

SEED = int(sys.argv[1]) if len(sys.argv) > 1 else 3
TRIALS = int(sys.argv[2]) if len(sys.argv) > 2 else 300
rng = np.random.default_rng(SEED)
FIELDS = ['node_visits', 'node_values', 'node_raw_values', 'parents', 'action_from_parent', 'to_play', 'children_index',
          'children_prior_logits', 'children_visits', 'children_rewards', 'children_values']


class TableNet:
    """Replays prescribed outputs: root (logits, value) then one (logits, value, reward) per simulation."""
    def __init__(self, root_logits, root_value, logits, value, reward, H=2):
        self.r, self.rv, self.l, self.v, self.w, self.H, self.k = root_logits, root_value, logits, value, reward, H, 0

    def initial_inference(self, obs):
        return (torch.as_tensor(self.r[None]), torch.zeros(1, self.H), torch.as_tensor(np.array([self.rv])))

    def recurrent_inference(self, h, a):
        k = self.k
        self.k += 1
        return (torch.as_tensor(self.l[k][None]), torch.zeros(1, self.H), torch.as_tensor(np.array([self.v[k]])),
                torch.as_tensor(np.array([self.w[k]])))


class Capturing(MCTS):
    def initialize_tree(self, *a, **k):
        self.captured = super().initialize_tree(*a, **k)
        return self.captured


bad = 0
real_uniform, real_gumbel, real_dirichlet = np.random.uniform, np.random.gumbel, np.random.dirichlet
for trial in range(TRIALS):
    gumbel = bool(rng.integers(2)); A = int(rng.integers(2, 25)); S = int(rng.integers(1, 40)); players = int(rng.integers(1, 3))
    gamma = float(rng.choice([1.0, 0.997, 0.5]))
    cfg = make_config(gumbel, A, S, players, gamma)
    if gumbel: cfg.max_considered_actions = int(rng.choice([1, 2, 4, 16]))
    q = float(rng.choice([0.0, 0.25, 1.0]))
    def draw(shape, scale):
        x = rng.normal(size=shape) * scale
        return np.round(x / q) * q if q else x
    scale = float(rng.choice([0.5, 3.0, 20.0, 60.0]))
    logits = draw((S, A), scale).astype(np.float32)
    value, reward = draw((S,), 1.0), draw((S,), 1.0) * float(rng.integers(2))
    root_logits, root_value = draw((A,), scale).astype(np.float32), float(draw((1,), 1.0)[0])
    legal = np.ones(A)
    if A > 2 and rng.integers(2):
        legal[rng.choice(A, size=int(rng.integers(1, A - 1)), replace=False)] = 0
    to_play = int(rng.integers(0, players))
    temperature = float(rng.choice([0.0, 0.5, 1.0]))
    g = rng.gumbel(size=A); di = rng.dirichlet([0.25] * A); u = float(rng.random())
    calls = [0]
    def counter_uniform(low=0.0, high=1.0, size=None):
        row = mznoise.noise_row(trial, 7 * trial, calls[0], int(size))
        calls[0] += 1
        return row
    net = TableNet(root_logits, root_value, logits, value, reward)
    m = Capturing(cfg)
    np.random.uniform, np.random.gumbel, np.random.dirichlet = counter_uniform, (lambda size=None: g.copy()), (lambda alpha: di.copy())
    np.random.seed(trial)
    state = np.random.get_state()
    try:
        action, rv, probs = m.search(net, np.zeros((1, 1, 2)), legal, lambda a: np.eye(A)[a], to_play, temperature)
    finally:
        np.random.uniform, np.random.gumbel, np.random.dirichlet = real_uniform, real_gumbel, real_dirichlet
    np.random.set_state(state)
    u = np.random.random_sample()          # the uniform np.random.choice consumed (PUCT)
    t = m.captured
    o = orc.Oracle(cfg, 0)
    ref_prior = np.asarray(t.children_prior_logits[0])
    prior = orc.gumbel_root_prior(root_logits, legal) if gumbel else orc.puct_root_prior(root_logits, di, cfg.exploration_frac, legal)
    if gumbel:
        same = np.array_equal(prior, ref_prior)
    else:   # float64 log: libm and NumPy may differ in the last bit, so the search below starts from the reference's row
        lm = legal != 0
        scale_ = np.abs(ref_prior[lm])
        scale_ = float(np.nanmax(scale_)) if np.isfinite(scale_).any() else 1.0
        same = np.array_equal(prior[~lm], ref_prior[~lm]) and np.allclose(prior[lm], ref_prior[lm], rtol=0, atol=4e-16 * max(1.0, scale_), equal_nan=True)
    o.init(ref_prior, root_value, None, to_play, legal, g if gumbel else None)
    o.search(o.table_net(logits, value, reward), o.counter_noise(trial, 7 * trial))
    a2, p2, _ = o.to_act(temperature, u)
    v2 = o.arrays['node_values'][0]
    why = [] if same else ['root prior']
    for f in FIELDS:
        ref = np.asarray(getattr(t, f))
        if not np.array_equal(o.arrays[f].reshape(ref.shape), ref, equal_nan=True):
            why.append(f)
    if not np.allclose(np.asarray(probs, float), np.asarray(p2, float), rtol=1e-12, atol=1e-300, equal_nan=True):
        why.append('policy')
    same = not why
    same = same and int(action) == int(a2) and (float(rv) == float(v2))
    same = same and np.allclose(np.asarray(probs, float), np.asarray(p2, float), rtol=1e-12, atol=1e-300, equal_nan=True)
    if not same:
        bad += 1
        print('MISMATCH trial', trial, dict(gumbel=gumbel, A=A, S=S, players=players, gamma=gamma, q=q, scale=scale, T=temperature),
              'action', action, a2, 'value', rv, v2, why, flush=True)
    if trial % 50 == 0: print('trial', trial, flush=True)
print('done, mismatching trials:', bad)
