"""Shared helpers for the tests: fixture access and config construction."""
import os
from types import SimpleNamespace

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')

TREE_FIELDS = ['node_visits', 'node_values', 'node_raw_values', 'parents', 'action_from_parent',
               'children_index', 'children_prior_logits', 'children_visits', 'children_rewards',
               'children_values', 'to_play']


def load(name):
    return np.load(os.path.join(GOLDEN, name))


def make_config(gumbel, A, S, players, gamma, alpha=0.25, c_visit=50, c_scale=1.0, m=16):
    return SimpleNamespace(simulations=int(S), n_actions=int(A), gamma=float(gamma) if gamma != 1 else 1,
                           players=int(players), gumbel=bool(gumbel), max_considered_actions=m,
                           gumbel_scale=1.0, c_visit=c_visit, c_scale=c_scale, c_base=19625, c_init=1.25,
                           dirichlet_alpha=alpha, exploration_frac=0.25)


class HashCase:
    """One case of search_hash_{puct,gumbel}.npz / search_stream_puct.npz."""

    def __init__(self, z, prefix):
        self.z, self.p = z, prefix
        m = z[prefix + 'meta']
        (self.gumbel, self.A, self.S, self.players, self.H, self.to_play, self.parent_predicts,
         self.logit_shift, self.reward_on, self.seed, self.idx, self.root_code, self.calls) = [int(x) for x in m]
        self.gamma = float(z[prefix + 'gamma'])
        alpha = float(z[prefix + 'alpha']) if (prefix + 'alpha') in z else 0.25
        self.cfg = make_config(self.gumbel, self.A, self.S, self.players, self.gamma, alpha=alpha)

    def __getitem__(self, k):
        return self.z[self.p + k]

    def has(self, k):
        return (self.p + k) in self.z


def hash_cases(name, prefix='c'):
    z = load(name)
    return [HashCase(z, f'{prefix}{i}_') for i in range(int(z['n_cases']))]


def assert_tree_equal(got, case, fields=TREE_FIELDS, hidden=True):
    """Bit-exact comparison of every reference Tree field."""
    for f in fields:
        ref = case[f]
        val = np.asarray(getattr(got, f))
        assert val.shape == ref.shape or val.reshape(ref.shape).shape == ref.shape, (f, val.shape, ref.shape)
        val = val.reshape(ref.shape)
        assert val.dtype == ref.dtype, (f, val.dtype, ref.dtype)
        if not np.array_equal(val, ref):
            bad = np.argwhere(val != ref)
            raise AssertionError(f'{f}: {len(bad)} mismatches, first at {bad[0]}: '
                                 f'{val[tuple(bad[0])]!r} vs {ref[tuple(bad[0])]!r}')
    if hidden:
        ref = case['hidden_states']
        val = np.asarray(got.hidden_states).reshape(ref.shape)
        assert np.array_equal(val, ref), 'hidden_states'


def replay_on_checker(cfg, tree, H, legal, to_play, seed=0, tree_id=0, gumbel_noise=None):
    """The checker's single-tree search fed with the network outputs stored in a finished tree (an export of the GPU
    engine): if the engine's tree logic is right, the checker rebuilds that very tree, whatever evaluated the leaves."""
    from oracle import oracle as orc
    S, A = cfg.simulations, cfg.n_actions
    par, act = tree['parents'], tree['action_from_parent']
    rec_logits = tree['children_prior_logits'][1:].astype(np.float32)
    assert np.array_equal(rec_logits.astype(tree['children_prior_logits'].dtype), tree['children_prior_logits'][1:])  # float32 values
    rec_value = tree['node_raw_values'][1:]
    rec_reward = np.array([tree['children_rewards'][par[k], act[k]] for k in range(1, S + 1)])
    rec_hidden = tree['hidden_states'][1:].reshape(S, H)
    o = orc.Oracle(cfg, H)
    o.init(tree['children_prior_logits'][0], tree['node_raw_values'][0], tree['hidden_states'][0], to_play, legal, gumbel_noise)
    o.search(o.table_net(rec_logits, rec_value, rec_reward, rec_hidden), o.counter_noise(seed, tree_id))
    return o


def hidden_conditioned_error(net, parent_hidden, actions, stored_next_hidden):
    """Largest error of stored next hidden states, measured where it is well conditioned: the reference scales every
    group (plane) of the dynamics' raw output as (x - min) / (max - min + 1e-5) (utils/network_utils.py:132-158), which
    divides last-bit differences of the raw output by the plane's range - unbounded for a nearly flat plane.  Multiplied
    back by that range, the difference of two float32 evaluators is again a few ulps of the raw values.
    net: lowered network on the CPU (muzero_amd.network); parent_hidden f32[n, H]; actions int[n]; stored f32[n, H].
    Returns max over elements of |stored - reference| * (range of the element's group + 1e-5)."""
    import torch
    h = torch.as_tensor(np.ascontiguousarray(parent_hidden, np.float32))
    a = torch.as_tensor(np.asarray(actions, np.int64))
    with torch.no_grad():
        raw = net.dyn.run({'xa': torch.cat([h, net.encode_actions(a)], dim=1)})['next_raw'].numpy()
    n, H = raw.shape
    groups = int(net.norm_groups)
    r = raw.reshape(n, groups, H // groups)
    mn, mx = r.min(axis=2, keepdims=True), r.max(axis=2, keepdims=True)
    ref = (r - mn) / (mx - mn + np.float32(1e-5))
    got = np.asarray(stored_next_hidden, np.float32).reshape(n, groups, H // groups)
    return float((np.abs(got - ref) * (mx - mn + 1e-5)).max())
