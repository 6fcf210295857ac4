"""Generates the golden vectors under tests/golden/ by running the REFERENCE itself.

Runs only where /root/reference exists (the build container); the .npz files it
writes are data (inputs + the reference's outputs) and are committed, this
script is committed as their provenance.  Nothing from the reference is copied:
its `mcts`, `network`, `games` and `utils` modules are imported and driven.

    python tests/golden/make_fixtures.py
"""
from __future__ import annotations

import os
import sys
import types
import warnings
from types import SimpleNamespace

import numpy as np

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.modules.setdefault('pygraphviz', types.ModuleType('pygraphviz'))  # only visualize_tree uses it
sys.path.insert(0, '/root/reference')
warnings.filterwarnings('ignore', category=DeprecationWarning)

import torch  # noqa: E402
from mcts.mcts import MCTS  # noqa: E402  (reference)
from mcts import action_selection as ref_as, qtransforms as ref_qt  # noqa: E402
from mcts.tree import Tree  # noqa: E402
import network as ref_network  # noqa: E402
from utils import network_utils as ref_nu  # noqa: E402
from utils.utils import set_seed  # noqa: E402
from games.tictactoe import TicTacToe  # noqa: E402
from games.game import GameHistory  # noqa: E402

sys.path.insert(0, REPO)
from muzero_amd import hashnet, noise as mznoise  # noqa: E402

TREE_FIELDS = ['node_visits', 'node_values', 'node_raw_values', 'parents', 'action_from_parent',
               'children_index', 'children_prior_logits', 'children_visits', 'children_rewards',
               'children_values', 'hidden_states', 'to_play']


class CapturingMCTS(MCTS):
    """The arrays of the tree built by initialize_tree are mutated in place by the
    rest of the search, so keeping the object gives the final tree."""

    def initialize_tree(self, *a, **k):
        self.captured = super().initialize_tree(*a, **k)
        return self.captured


def make_config(gumbel, A, S, players, gamma, alpha=0.25, c_visit=50, c_scale=1.0, m=16):
    return SimpleNamespace(simulations=S, n_actions=A, gamma=gamma, players=players, gumbel=gumbel,
                           max_considered_actions=m, gumbel_scale=1.0, c_visit=c_visit, c_scale=c_scale,
                           c_base=19625, c_init=1.25, dirichlet_alpha=alpha, exploration_frac=0.25)


def c_table(cfg):
    n = np.arange(cfg.simulations + 2, dtype=np.int32)
    return np.array([cfg.c_init + np.log((k + cfg.c_base + 1) / cfg.c_base) for k in n])


def tree_dict(tree, prefix):
    out = {}
    for f in TREE_FIELDS:
        out[prefix + f] = np.array(getattr(tree, f))
    return out


class CounterUniform:
    """Stands in for np.random.uniform inside the reference's PUCT selection."""

    def __init__(self, seed, tree_id):
        self.seed, self.tree_id, self.calls = seed, tree_id, 0

    def __call__(self, low=0.0, high=1.0, size=None):
        row = mznoise.noise_row(self.seed, self.tree_id, self.calls, int(size))
        self.calls += 1
        return row


def run_hash_case(idx, gumbel, A, S, players, gamma, H, to_play, legal, parent_predicts, logit_shift,
                  reward_on, temperature, seed, per_node=False):
    cfg = make_config(gumbel, A, S, players, gamma)
    hp = hashnet.HashNetParams(A, H, logit_shift, reward_on, parent_predicts)
    net = hashnet.HashNetwork(hp)
    root_code = mznoise.splitmix64(seed * 7919 + idx) & 0xFFFFFFFF
    obs = hashnet.observation_of(root_code)
    mcts = CapturingMCTS(cfg)
    np.random.seed(seed)
    rs = np.random.RandomState(seed)
    out = {}
    real_uniform = np.random.uniform
    cu = CounterUniform(seed, idx)
    if gumbel:
        g = rs.gumbel(size=A)
        dirichlet, u = np.zeros(A), 0.0
    else:
        dirichlet = rs.dirichlet([cfg.dirichlet_alpha] * A)
        u = rs.random_sample()
        g = np.zeros(A)
    np.random.uniform = cu
    try:
        enc = lambda a: np.eye(A)[a]  # noqa: E731
        action, root_value, probs = mcts.search(net, obs, legal, enc, to_play, temperature)
    finally:
        np.random.uniform = real_uniform
    tree = mcts.captured
    if gumbel:
        assert np.array_equal(tree.gumbel_noise, g)
    p = f'c{idx}_'
    out.update(tree_dict(tree, p))
    out[p + 'meta'] = np.array([gumbel, A, S, players, H, to_play, parent_predicts, logit_shift,
                                reward_on, seed, idx, root_code, cu.calls], dtype=np.int64)
    out[p + 'gamma'] = np.float64(gamma)
    out[p + 'temperature'] = np.float64(temperature)
    out[p + 'legal'] = np.asarray(legal, np.float64)
    out[p + 'gumbel_noise'] = g
    out[p + 'dirichlet'] = dirichlet
    out[p + 'choice_u'] = np.float64(u)
    out[p + 'root_logits'] = hashnet.logits_of(root_code, hp)
    out[p + 'action'] = np.int64(action)
    out[p + 'root_value'] = np.float64(root_value)
    out[p + 'probs'] = np.asarray(probs, np.float64)
    out[p + 'c_table'] = c_table(cfg)
    if per_node:
        N = S + 1
        qt = np.zeros((N, A))
        sel = np.zeros(N, np.int64)
        for n in range(N):
            if gumbel:
                qt[n] = ref_qt.qtransform_completed_by_mix_value(tree, n, cfg.c_visit, cfg.c_scale)
                sel[n] = ref_as.gumbel_muzero_interior_action_selection(
                    tree, n, 1, cfg.gamma,
                    lambda t, i: ref_qt.qtransform_completed_by_mix_value(t, i, cfg.c_visit, cfg.c_scale))
            else:
                qt[n] = ref_qt.qtransform_by_parent_and_siblings(tree, n)
                np.random.uniform = lambda size=None: np.zeros(size)
                try:
                    sel[n] = ref_as.muzero_action_selection(
                        tree, n, 1, cfg.gamma, cfg.c_base, cfg.c_init,
                        ref_qt.qtransform_by_parent_and_siblings)
                finally:
                    np.random.uniform = real_uniform
        out[p + 'node_qtransform'] = qt
        out[p + 'node_select'] = sel
    return out


def gen_hash_fixtures():
    shapes = [  # (A, players, gamma, H)
        (9, 2, 1, 16),
        (2, 1, 0.997, 8),
        (18, 1, 0.997, 24),
    ]
    for gumbel in (False, True):
        out, idx = {}, 0
        rng = np.random.RandomState(1234 + int(gumbel))
        for (A, players, gamma, H) in shapes:
            for S in (10, 25, 50, 100):
                for to_play in ((0, 1) if players == 2 else (0,)):
                    for partial in (False, True):
                        if partial and A == 2 and S != 25:
                            continue
                        legal = np.ones(A)
                        if partial:
                            k = 1 if A == 2 else int(rng.randint(1, A - 2))
                            legal[rng.choice(A, size=k, replace=False)] = 0
                        pp = 1 if (idx % 3) else 0
                        shift = (3, 5, 2)[idx % 3]
                        temperature = (0.0, 1.0, 0.5, 0.25)[idx % 4]
                        out.update(run_hash_case(idx, gumbel, A, S, players, gamma, H, to_play, legal,
                                                 pp, shift, 1 if idx % 2 else 0, temperature,
                                                 seed=100 + idx, per_node=(S in (10, 50))))
                        idx += 1
        out['n_cases'] = np.int64(idx)
        name = 'search_hash_gumbel.npz' if gumbel else 'search_hash_puct.npz'
        np.savez_compressed(os.path.join(HERE, name), **out)
        print(name, idx, 'cases')


def gen_stream_fixtures():
    """PUCT searches that consume the real global np.random stream (dirichlet, the
    per-selection uniforms, the final choice), as a caller of the reference sees it."""
    out, idx = {}, 0
    real_uniform = np.random.uniform
    for (A, players, gamma, H, S, temperature) in [(9, 2, 1, 16, 25, 1.0), (9, 2, 1, 16, 50, 0.0),
                                                    (2, 1, 0.997, 8, 50, 1.0), (18, 1, 0.997, 24, 30, 0.5)]:
        for seed in (0, 1, 2):
            cfg = make_config(False, A, S, players, gamma, alpha=0.2 if A == 9 else 0.25)
            hp = hashnet.HashNetParams(A, H, 4, 1, 1)
            net = hashnet.HashNetwork(hp)
            root_code = (0xABCDE + 977 * idx) & 0xFFFFFFFF
            obs = hashnet.observation_of(root_code)
            legal = np.ones(A)
            if seed == 2 and A > 2:
                legal[[0, A - 1]] = 0
            rows = []

            def recording_uniform(low=0.0, high=1.0, size=None):
                r = real_uniform(low, high, size)
                rows.append(np.array(r))
                return r
            mcts = CapturingMCTS(cfg)
            set_seed(seed)
            np.random.uniform = recording_uniform
            try:
                action, root_value, probs = mcts.search(net, obs, legal, lambda a: np.eye(A)[a],
                                                        seed % players, temperature)
            finally:
                np.random.uniform = real_uniform
            after = np.random.random_sample()  # the next draw a caller would see
            p = f's{idx}_'
            out.update(tree_dict(mcts.captured, p))
            out[p + 'meta'] = np.array([0, A, S, players, H, seed % players, 1, 4, 1, seed, idx,
                                        root_code, len(rows)], dtype=np.int64)
            out[p + 'gamma'] = np.float64(gamma)
            out[p + 'alpha'] = np.float64(cfg.dirichlet_alpha)
            out[p + 'temperature'] = np.float64(temperature)
            out[p + 'legal'] = legal
            out[p + 'uniform_rows'] = np.array(rows)
            out[p + 'action'] = np.int64(action)
            out[p + 'root_value'] = np.float64(root_value)
            out[p + 'probs'] = np.asarray(probs, np.float64)
            out[p + 'next_random_sample'] = np.float64(after)
            idx += 1
    out['n_cases'] = np.int64(idx)
    np.savez_compressed(os.path.join(HERE, 'search_stream_puct.npz'), **out)
    print('search_stream_puct.npz', idx, 'cases')


# ---------------------------------------------------------------------- real networks

def tictactoe_net_config():
    return SimpleNamespace(network='resnet', observation_dim=[3, 3, 3], n_actions=9, n_stacked_observations=1,
                           blocks=1, channels=16, reduced_channels_reward=8, reduced_channels_policy=8,
                           reduced_channels_value=8, resnet_fc_reward_layers=[16], resnet_fc_policy_layers=[16],
                           resnet_fc_value_layers=[16], downsample=False, support_limit=1)


def cartpole_net_config():
    return SimpleNamespace(network='mlp', observation_dim=[1, 1, 4], n_actions=2, n_stacked_observations=1,
                           encoding_size=8, fc_reward_layers=[16], fc_policy_layers=[16], fc_value_layers=[16],
                           fc_representation_layers=[], fc_hidden_state_layers=[16], support_limit=10)


def synthetic_net_config():
    return SimpleNamespace(network='mlp', observation_dim=[1, 1, 32], n_actions=18, n_stacked_observations=1,
                           encoding_size=64, fc_reward_layers=[64], fc_policy_layers=[64], fc_value_layers=[64],
                           fc_representation_layers=[], fc_hidden_state_layers=[64], support_limit=30)


class RecordingNet:
    def __init__(self, net):
        self.net, self.rec = net, []

    def initial_inference(self, obs):
        return self.net.initial_inference(obs)

    def recurrent_inference(self, h, a):
        out = self.net.recurrent_inference(h, a)
        self.rec.append([o.detach().cpu().numpy().copy() for o in out])
        return out


def randomize_bn(net, seed):
    """Random-init BatchNorm has mean 0 / var 1 / weight 1 / bias 0, which would leave
    the BN-folding of the engine untested; give the running stats real values."""
    g = torch.Generator().manual_seed(seed)
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=g) * 0.1)
            m.running_var.copy_(torch.rand(m.running_var.shape, generator=g) + 0.5)
            m.weight.data.copy_(torch.rand(m.weight.shape, generator=g) + 0.5)
            m.bias.data.copy_(torch.randn(m.bias.shape, generator=g) * 0.1)


def gen_network_fixtures():
    for name, ncfg, obs_shape, enc_kind in [('mlp_cartpole', cartpole_net_config(), (1, 1, 4), 'onehot'),
                                            ('resnet_tictactoe', tictactoe_net_config(), (3, 3, 3), 'plane'),
                                            ('mlp_synthetic', synthetic_net_config(), (1, 1, 32), 'onehot')]:
        set_seed(0)
        net = ref_network.MuZeroNetwork(ncfg)
        if ncfg.network == 'resnet':
            randomize_bn(net, 1)
        net.eval()
        A = ncfg.n_actions
        out = {'sd_' + k: v.detach().cpu().numpy() for k, v in net.state_dict().items()}
        rng = np.random.RandomState(5)
        B = 64
        if enc_kind == 'plane':
            boards = rng.choice([-1, 0, 1], size=(B, 3, 3))
            tp = rng.choice([-1, 1], size=B)
            obs = np.stack([np.stack([(b == -1) * 1.0, (b == 1) * 1.0, np.full((3, 3), t * 1.0)])
                            for b, t in zip(boards, tp)])
        else:
            obs = rng.uniform(-0.05, 0.05, size=(B,) + obs_shape)
        actions = rng.randint(0, A, size=B)
        with torch.no_grad():
            lg, hid, val = net.initial_inference(torch.as_tensor(obs, dtype=torch.float32))
            if enc_kind == 'plane':
                enc = torch.zeros(B, 3, 3)
                enc.view(B, 9)[torch.arange(B), torch.as_tensor(actions)] = 1
            else:
                enc = torch.zeros(B, A)
                enc[torch.arange(B), torch.as_tensor(actions)] = 1
            lg2, hid2, val2, rew2 = net.recurrent_inference(hid, enc)
            # second step from the produced states, so normalised inputs are covered as well
            lg3, hid3, val3, rew3 = net.recurrent_inference(hid2, enc)
        out.update(obs=obs, actions=actions, init_logits=lg.numpy(), init_hidden=hid.numpy(),
                   init_value=val.numpy(), rec_logits=lg2.numpy(), rec_hidden=hid2.numpy(),
                   rec_value=val2.numpy(), rec_reward=rew2.numpy(), rec2_logits=lg3.numpy(),
                   rec2_hidden=hid3.numpy(), rec2_value=val3.numpy(), rec2_reward=rew3.numpy())
        # recorded searches: per-simulation network outputs + the reference's final tree
        if name == 'mlp_synthetic':
            cases = [(False, 40, 1, 0.997), (True, 40, 1, 0.997)]
        elif name == 'mlp_cartpole':
            cases = [(False, 50, 1, 0.997), (True, 50, 1, 0.997)]
        else:
            cases = [(False, 50, 2, 1), (True, 100, 2, 1), (True, 50, 2, 1)]
        real_uniform = np.random.uniform
        for ci, (gumbel, S, players, gamma) in enumerate(cases):
            cfg = make_config(gumbel, A, S, players, gamma, alpha=0.2 if A == 9 else 0.25)
            rnet = RecordingNet(net)
            mcts = CapturingMCTS(cfg)
            seed = 40 + ci
            np.random.seed(seed)
            rs = np.random.RandomState(seed)
            legal = np.ones(A)
            o = obs[ci]
            if enc_kind == 'plane':
                o = np.stack([np.zeros((3, 3)), np.zeros((3, 3)), np.full((3, 3), -1.0)])
                if ci == 2:
                    o[0, 1, 1] = 1; o[1, 0, 0] = 1; legal[[0, 4]] = 0
                encoder = lambda a: np.eye(9)[a].reshape(3, 3)  # noqa: E731
            else:
                encoder = lambda a: np.eye(A)[a]  # noqa: E731
            if gumbel:
                g, dirichlet, u = rs.gumbel(size=A), np.zeros(A), 0.0
            else:
                dirichlet = rs.dirichlet([cfg.dirichlet_alpha] * A)
                u, g = rs.random_sample(), np.zeros(A)
            cu = CounterUniform(seed, ci)
            np.random.uniform = cu
            try:
                with torch.no_grad():
                    action, root_value, probs = mcts.search(rnet, o, legal, encoder, ci % players, 1.0)
            finally:
                np.random.uniform = real_uniform
            with torch.no_grad():
                rl, rh, rv = net.initial_inference(torch.as_tensor(o, dtype=torch.float32).unsqueeze(0))
            p = f'r{ci}_'
            out.update(tree_dict(mcts.captured, p))
            out[p + 'meta'] = np.array([gumbel, A, S, players, int(np.prod(rh.shape[1:])), ci % players,
                                        seed, ci, cu.calls], dtype=np.int64)
            out[p + 'gamma'] = np.float64(gamma)
            out[p + 'alpha'] = np.float64(cfg.dirichlet_alpha)
            out[p + 'obs'] = o
            out[p + 'legal'] = legal
            out[p + 'gumbel_noise'] = g
            out[p + 'dirichlet'] = dirichlet
            out[p + 'choice_u'] = np.float64(u)
            out[p + 'root_logits'] = rl.numpy()[0]
            out[p + 'root_hidden'] = rh.numpy()[0]
            out[p + 'root_value'] = rv.numpy()[0]
            out[p + 'rec_logits'] = np.stack([r[0][0] for r in rnet.rec])
            out[p + 'rec_hidden'] = np.stack([r[1][0] for r in rnet.rec])
            out[p + 'rec_value'] = np.array([r[2][0] for r in rnet.rec])
            out[p + 'rec_reward'] = np.array([r[3][0] for r in rnet.rec])
            out[p + 'action'] = np.int64(action)
            out[p + 'search_value'] = np.float64(root_value)
            out[p + 'probs'] = np.asarray(probs, np.float64)
            out[p + 'c_table'] = c_table(cfg)
        out['n_search_cases'] = np.int64(len(cases))
        np.savez_compressed(os.path.join(HERE, f'net_{name}.npz'), **out)
        print(f'net_{name}.npz')


def downsample_net_config():
    return SimpleNamespace(network='resnet', observation_dim=[3, 48, 48], n_actions=9, n_stacked_observations=1,
                           blocks=1, channels=16, reduced_channels_reward=4, reduced_channels_policy=4,
                           reduced_channels_value=4, resnet_fc_reward_layers=[16], resnet_fc_policy_layers=[16],
                           resnet_fc_value_layers=[16], downsample=True, support_limit=5)


def gen_downsample_fixture():
    """The reference's residual network with `downsample` on (network.py:228-325): 48 x 48 frames -> a 3 x 3 state."""
    ncfg = downsample_net_config()
    set_seed(0)
    net = ref_network.MuZeroNetwork(ncfg)
    randomize_bn(net, 2)
    net.eval()
    out = {'sd_' + k: v.detach().cpu().numpy() for k, v in net.state_dict().items()}
    rng = np.random.RandomState(11)
    B = 4
    obs = rng.uniform(0.0, 1.0, size=(B, 3, 48, 48)).astype(np.float32)
    actions = rng.randint(0, 9, size=B)
    with torch.no_grad():
        lg, hid, val = net.initial_inference(torch.as_tensor(obs))
        enc = torch.zeros(B, 3, 3)
        enc.view(B, 9)[torch.arange(B), torch.as_tensor(actions)] = 1
        lg2, hid2, val2, rew2 = net.recurrent_inference(hid, enc)
        lg3, hid3, val3, rew3 = net.recurrent_inference(hid2, enc)
    out.update(obs=obs, actions=actions, init_logits=lg.numpy(), init_hidden=hid.numpy(), init_value=val.numpy(),
               rec_logits=lg2.numpy(), rec_hidden=hid2.numpy(), rec_value=val2.numpy(), rec_reward=rew2.numpy(),
               rec2_logits=lg3.numpy(), rec2_hidden=hid3.numpy(), rec2_value=val3.numpy(), rec2_reward=rew3.numpy())
    np.savez_compressed(os.path.join(HERE, 'net_resnet_downsample.npz'), **out)
    print('net_resnet_downsample.npz')


# ---------------------------------------------------------------------- misc known answers

def _find_filling_win():
    """A 9-move game whose last move completes a line (the reference scores it as a tie)."""
    import itertools
    lines = [(0, 1, 2), (3, 4, 5), (6, 7, 8), (0, 3, 6), (1, 4, 7), (2, 5, 8), (0, 4, 8), (2, 4, 6)]
    for perm in itertools.permutations(range(9)):
        b = [0] * 9
        ok = True
        for k, a in enumerate(perm):
            b[a] = -1 if k % 2 == 0 else 1
            won = any(abs(b[i] + b[j] + b[l]) == 3 for i, j, l in lines)
            if won and k < 8:
                ok = False
                break
        if ok and won:
            return list(perm)
    raise RuntimeError


FILLING_WIN = _find_filling_win()


def gen_misc():
    out = {}
    pairs = [(m, S) for m in (0, 1, 2, 3, 4, 5, 7, 8, 9, 16, 18) for S in (1, 5, 10, 25, 50, 100, 200)]
    out['cv_pairs'] = np.array(pairs)
    for m, S in pairs:
        out[f'cv_{m}_{S}'] = np.array(ref_as.get_considered_visits(m, S), dtype=np.int32)
    rng = np.random.RandomState(3)
    for L in (1, 10, 300):
        lg = (rng.randn(32, 2 * L + 1) * 2).astype(np.float32)
        out[f'sts_logits_{L}'] = lg
        out[f'sts_out_{L}'] = ref_nu.support_to_scalar(torch.as_tensor(lg), L).numpy()
    h = rng.randn(16, 8).astype(np.float32)
    out['norm_mlp_in'] = h
    out['norm_mlp_out'] = ref_nu.mlp_normalize_hidden_state(torch.as_tensor(h)).numpy()
    h4 = rng.randn(8, 16, 3, 3).astype(np.float32)
    out['norm_res_in'] = h4
    out['norm_res_out'] = ref_nu.normalize_hidden_state(torch.as_tensor(h4)).numpy()
    # float32 exp and pairwise sums as NumPy evaluates them
    x = np.concatenate([rng.uniform(-104, 89, 4000), rng.randn(4000) * 3, [0.0, -0.0, 88.7, -103.9, 1e-8]]
                       ).astype(np.float32)
    out['expf_in'] = x
    with np.errstate(over='ignore'):
        out['expf_out'] = np.exp(x)
    for n in (2, 7, 8, 9, 16, 18, 33, 64):
        a = rng.randn(200, n)
        out[f'sum64_in_{n}'] = a
        out[f'sum64_out_{n}'] = np.array([np.sum(r) for r in a])
        af = a.astype(np.float32)
        out[f'sum32_out_{n}'] = np.array([np.sum(r) for r in af], dtype=np.float32)
    # TicTacToe traces: random legal play, plus the board-filling win (winner overwritten by the tie rule)
    traces = []
    for g in range(40):
        game = TicTacToe()
        r = np.random.RandomState(g)
        moves = []
        if g == 0:
            script = [0, 1, 2, 4, 3, 5, 7, 6, 8]  # a full board without a line
        elif g == 1:
            script = [0, 1, 3, 4, 2, 6, 5, 8, 7]  # placeholder, replaced below by a board-filling WIN
            script = FILLING_WIN
        else:
            script = None
        obs_list, rew_list, term_list, legal_list, tp_list = [], [], [], [], []
        t = 0
        while True:
            legal = game.legal_actions()
            a = script[t] if script else int(r.choice(np.flatnonzero(legal)))
            tp_list.append(game.to_play())
            legal_list.append(legal)
            o, rew, term = game.step(a)
            moves.append(a); obs_list.append(o); rew_list.append(rew); term_list.append(term)
            t += 1
            if term:
                break
        traces.append(dict(moves=moves, obs=obs_list, rew=rew_list, term=term_list, legal=legal_list, tp=tp_list))
    out['ttt_n'] = np.int64(len(traces))
    for i, tr in enumerate(traces):
        out[f'ttt{i}_moves'] = np.array(tr['moves'])
        out[f'ttt{i}_obs'] = np.array(tr['obs'], dtype=np.float64)
        out[f'ttt{i}_rew'] = np.array(tr['rew'], dtype=np.float64)
        out[f'ttt{i}_term'] = np.array(tr['term'])
        out[f'ttt{i}_legal'] = np.array(tr['legal'])
        out[f'ttt{i}_tp'] = np.array(tr['tp'])
    np.savez_compressed(os.path.join(HERE, 'misc.npz'), **out)
    print('misc.npz')


def gen_selfplay():
    """Whole self-play games: the loop of the reference's SelfPlay.play (self_play.py:77-115, which
    cannot be imported here because it needs ray) spelled out with the reference's own MCTS, TicTacToe
    and GameHistory objects."""
    out, gi = {}, 0
    real_uniform = np.random.uniform
    for gumbel in (True, False):
        for seed in (0, 1, 2):
            cfg = make_config(gumbel, 9, 20, 2, 1, alpha=0.2)
            cfg.n_stacked_observations, cfg.stack_action, cfg.max_moves = 1, False, 9
            net = hashnet.HashNetwork(hashnet.HashNetParams(9, 8, 3, 1, 1), obs_hash=True)
            game = TicTacToe()
            mcts = MCTS(cfg)
            np.random.seed(seed)
            temperature = 1.0
            observation = game.reset()
            history = GameHistory(game.action_encoder, observation)
            cu = CounterUniform(seed, gi)
            np.random.uniform = cu
            search_obs = []
            try:
                while True:
                    to_play = game.to_play()
                    stacked = history.stack_n_observations(-1, cfg.n_stacked_observations, cfg.n_actions, cfg.stack_action)
                    search_obs.append(np.array(stacked, dtype=np.float64))
                    action, root_value, probs = mcts.search(net, stacked, game.legal_actions(), game.action_encoder,
                                                            to_play, temperature)
                    next_observation, reward, terminated = game.step(action)
                    history.save(observation, action, reward, to_play, probs, root_value)
                    observation = next_observation
                    if terminated or len(history) > cfg.max_moves:
                        break
            finally:
                np.random.uniform = real_uniform
            p = f'g{gi}_'
            out[p + 'meta'] = np.array([gumbel, seed, gi, len(history)], dtype=np.int64)
            out[p + 'observations'] = np.array(history.observations, dtype=np.float64)
            out[p + 'search_obs'] = np.array(search_obs)
            out[p + 'actions'] = np.array(history.actions, dtype=np.int64)
            out[p + 'rewards'] = np.array(history.rewards, dtype=np.float64)
            out[p + 'to_plays'] = np.array(history.to_plays, dtype=np.int64)
            out[p + 'probs'] = np.array(history.action_probabilities, dtype=np.float64)
            out[p + 'root_values'] = np.array(history.root_values, dtype=np.float64)
            out[p + 'final_obs'] = np.array(observation, dtype=np.float64)
            # targets the replay buffer would build from this game (games/game.py:185-253)
            vt, rt, pt = history.make_target(0, 5, cfg.gamma, 5, 9)
            out[p + 'target_value'], out[p + 'target_reward'] = np.array(vt, dtype=np.float64), np.array(rt, dtype=np.float64)
            out[p + 'target_policy'] = np.array(pt, dtype=np.float64)
            out[p + 'return_p0'] = np.float64(history.compute_return(cfg.gamma, 0, 2))
            out[p + 'return_p1'] = np.float64(history.compute_return(cfg.gamma, 1, 2))
            gi += 1
    out['n_games'] = np.int64(gi)
    np.savez_compressed(os.path.join(HERE, 'selfplay_tictactoe.npz'), **out)
    print('selfplay_tictactoe.npz', gi, 'games')


def _play_tictactoe(cfg, net, seed, tree_id):
    """One self-play game with the reference's objects (the loop of self_play.py:77-115)."""
    real_uniform = np.random.uniform
    game = TicTacToe()
    mcts = MCTS(cfg)
    np.random.seed(seed)
    observation = game.reset()
    history = GameHistory(game.action_encoder, observation)
    np.random.uniform = CounterUniform(seed, tree_id)
    try:
        while True:
            to_play = game.to_play()
            stacked = history.stack_n_observations(-1, cfg.n_stacked_observations, cfg.n_actions, cfg.stack_action)
            action, root_value, probs = mcts.search(net, stacked, game.legal_actions(), game.action_encoder, to_play, 1.0)
            next_observation, reward, terminated = game.step(action)
            history.save(observation, action, reward, to_play, probs, root_value)
            observation = next_observation
            if terminated or len(history) > cfg.max_moves:
                break
    finally:
        np.random.uniform = real_uniform
    return history


def gen_reanalyse():
    """The body of Reanalyser.reanalyse (reanalyse.py:44-78; the actor itself needs ray) spelled out with
    the reference's MCTS, TicTacToe and GameHistory: every position of a stored game is searched again
    with a target network, from a fresh game's legal mask, and the game's targets are rebuilt."""
    out, gi = {}, 0
    real_uniform = np.random.uniform
    for gumbel in (True, False):
        for seed in (3, 4):
            cfg = make_config(gumbel, 9, 20, 2, 1, alpha=0.2)
            cfg.n_stacked_observations, cfg.stack_action, cfg.max_moves = 1, False, 9
            play_net = hashnet.HashNetwork(hashnet.HashNetParams(9, 8, 3, 1, 1), obs_hash=True)
            history = _play_tictactoe(cfg, play_net, seed, gi)
            target_net = hashnet.HashNetwork(hashnet.HashNetParams(9, 8, 4, 1, 1), obs_hash=True)
            fresh_game = TicTacToe()   # the reanalyser's own game object is never stepped
            mcts = MCTS(cfg)
            observations = [history.stack_n_observations(t, cfg.n_stacked_observations, cfg.n_actions, cfg.stack_action)
                            for t in range(len(history))]
            values = [target_net.initial_inference(torch.as_tensor(o, dtype=torch.float32).unsqueeze(0))[2].item()
                      for o in observations]
            np.random.seed(100 + seed)
            probs, roots = [], []
            try:
                for t, o in enumerate(observations):
                    np.random.uniform = CounterUniform(100 + seed, t)   # one noise key per position
                    _, rv, ap = mcts.search(target_net, o, fresh_game.legal_actions(), fresh_game.action_encoder,
                                            history.to_plays[t])
                    probs.append(ap)
                    roots.append(rv)
            finally:
                np.random.uniform = real_uniform
            p = f'g{gi}_'
            out[p + 'meta'] = np.array([gumbel, seed, gi, len(history)], dtype=np.int64)
            out[p + 'observations'] = np.array(history.observations, dtype=np.float64)
            out[p + 'actions'] = np.array(history.actions, dtype=np.int64)
            out[p + 'rewards'] = np.array(history.rewards, dtype=np.float64)
            out[p + 'to_plays'] = np.array(history.to_plays, dtype=np.int64)
            out[p + 'probs'] = np.array(history.action_probabilities, dtype=np.float64)
            out[p + 'root_values'] = np.array(history.root_values, dtype=np.float64)
            out[p + 'stacked'] = np.array(observations, dtype=np.float64)
            out[p + 'target_net_values'] = np.array(values, dtype=np.float64)
            out[p + 're_probs'] = np.array(probs, dtype=np.float64)
            out[p + 're_root_values'] = np.array(roots, dtype=np.float64)
            out[p + 'next_random_sample'] = np.float64(np.random.random_sample())
            history.save_reanalysed_stats(probs, roots)
            vt, rt, pt = history.make_target(1, 5, cfg.gamma, 5, 9)
            out[p + 'target_value'], out[p + 'target_reward'] = np.array(vt, dtype=np.float64), np.array(rt, dtype=np.float64)
            out[p + 'target_policy'] = np.array(pt, dtype=np.float64)
            gi += 1
    out['n_games'] = np.int64(gi)
    np.savez_compressed(os.path.join(HERE, 'reanalyse_tictactoe.npz'), **out)
    print('reanalyse_tictactoe.npz', gi, 'games')


def gen_targets():
    """Training targets as ReplayBuffer.sample builds them (replay_buffer.py:59-105; the buffer itself is
    a ray actor): reference GameHistory objects filled with synthetic trajectories, sampled positions,
    `stack_n_observations`, absorbing-action padding and `make_target`, converted like `ftensor`."""
    rng = np.random.default_rng(7)
    out, ci = {}, 0
    for players, gamma, A, td, unroll, reanalysed in ((1, 0.997, 2, 10, 5, False), (1, 0.997, 2, 3, 5, True),
                                                       (2, 1, 9, 9, 5, False), (2, 0.9, 4, 4, 3, True),
                                                       (2, 0.9, 4, 50, 6, False)):
        one_hot = lambda a, A=A: np.eye(A)[a][None]  # noqa: E731
        games = []
        for g in range(6):
            T = int(rng.integers(1, 14))
            h = GameHistory(one_hot, np.zeros((1, 1, 4)))
            for t in range(T):
                pi = rng.dirichlet([0.5] * A)
                h.save(rng.normal(size=(1, 1, 4)), int(rng.integers(A)), float(rng.integers(0, 3)) * 0.5,
                       t % players, [float(x) for x in pi], float(rng.normal()))
            if reanalysed:
                h.save_reanalysed_stats([[float(x) for x in rng.dirichlet([0.5] * A)] for _ in range(T)],
                                        [float(rng.normal()) for _ in range(T)])
            games.append(h)
        Bs = 16
        np.random.seed(11 + ci)
        sel = np.random.choice(range(len(games)), size=Bs)
        batch, ts, absorbed = [[], [], [], [], []], [], []
        for gidx in sel:
            h = games[gidx]
            t = np.random.randint(len(h))
            ts.append(t)
            obs = h.stack_n_observations(t, 1, A, False)
            enc = h.encoded_actions[t:t + unroll + 1]
            pad = []
            if len(enc) < unroll + 1:
                pad = list(np.random.choice(range(len(h.encoded_actions)), size=unroll + 1 - len(enc)))
                enc = enc + [h.encoded_actions[i] for i in pad]
            absorbed.append(pad + [-1] * (unroll + 1 - len(pad)))
            vt, rt, pt = h.make_target(t, td, gamma, unroll, A)
            for k, x in enumerate((obs, enc, vt, rt, pt)):
                batch[k].append(x)
        p = f'c{ci}_'
        out[p + 'meta'] = np.array([players, A, td, unroll, int(reanalysed), len(games), Bs], dtype=np.int64)
        out[p + 'gamma'] = np.float64(gamma)
        out[p + 'gamma_pow'] = np.array([gamma ** k for k in range(td + 1)], dtype=np.float64)
        for g, h in enumerate(games):
            q = f'{p}g{g}_'
            out[q + 'observations'] = np.array(h.observations, dtype=np.float64)
            out[q + 'actions'] = np.array(h.actions, dtype=np.int64)
            out[q + 'rewards'] = np.array(h.rewards, dtype=np.float64)
            out[q + 'to_plays'] = np.array(h.to_plays, dtype=np.int64)
            out[q + 'probs'] = np.array(h.action_probabilities, dtype=np.float64)
            out[q + 'root_values'] = np.array(h.root_values, dtype=np.float64)
            if reanalysed:
                out[q + 're_probs'] = np.array(h.reanalysed_action_probabilities, dtype=np.float64)
                out[q + 're_root_values'] = np.array(h.reanalysed_root_values, dtype=np.float64)
        out[p + 'sel'] = np.asarray(sel, dtype=np.int64)
        out[p + 't'] = np.asarray(ts, dtype=np.int64)
        out[p + 'absorbed'] = np.asarray(absorbed, dtype=np.int64)
        names = ('obs', 'actions', 'value', 'reward', 'policy')
        for k, n in enumerate(names):   # ftensor(np.asarray(.)): float32 tensors
            out[p + 'batch_' + n] = torch.as_tensor(np.asarray(batch[k]), dtype=torch.float32).numpy()
        # what the trainer does first with the value / reward targets (trainer.py:70-71)
        L = 10 if ci % 2 == 0 else 300
        out[p + 'support_limit'] = np.int64(L)
        for n in ('value', 'reward'):
            out[p + n + '_support'] = ref_nu.scalar_to_support(torch.as_tensor(out[p + 'batch_' + n]), L).numpy()
        ci += 1
    out['n_cases'] = np.int64(ci)
    np.savez_compressed(os.path.join(HERE, 'targets.npz'), **out)
    print('targets.npz', ci, 'cases')


def gen_stacking():
    """GameHistory.stack_n_observations (games/game.py:120-160) for more than one stacked observation, with and
    without action planes: what self-play hands to the search before every move (self_play.py:86-91: t = -1 on the
    history so far, the empty history included) and what the replay buffer stacks for a sampled position."""
    rng = np.random.default_rng(23)
    out, ci = {}, 0
    for A, shape, T in ((9, (3, 3, 3), 9), (2, (1, 1, 4), 12), (4, (2, 2, 3), 5)):
        h = GameHistory(lambda a, A=A: np.eye(A)[a][None], rng.normal(size=shape))
        obs = [h.initial_observation] + [rng.normal(size=shape) for _ in range(T)]
        acts = [int(rng.integers(A)) for _ in range(T)]
        p = f'c{ci}_'
        out[p + 'observations'] = np.array(obs[:T], dtype=np.float64)   # observations[t] is saved with move t
        out[p + 'actions'] = np.array(acts, dtype=np.int64)
        out[p + 'meta'] = np.array([A, T], dtype=np.int64)
        for n, stack_action in ((1, True), (2, False), (3, True), (4, False), (6, True)):
            q = f'{p}n{n}a{int(stack_action)}_'
            g = GameHistory(h.action_encoder, h.initial_observation)
            before = [g.stack_n_observations(-1, n, A, stack_action)]     # before move 0: the empty history
            for t in range(T):
                g.save(obs[t], acts[t], 0.0, 0, [1.0 / A] * A, 0.0)
                before.append(g.stack_n_observations(-1, n, A, stack_action))
            out[q + 'before_move'] = np.array(before, dtype=np.float64)   # [T + 1]: entry t = the search input of move t
            out[q + 'at'] = np.array([g.stack_n_observations(t, n, A, stack_action) for t in range(T)], dtype=np.float64)
        ci += 1
    out['n_cases'] = np.int64(ci)
    np.savez_compressed(os.path.join(HERE, 'stacking.npz'), **out)
    print('stacking.npz', ci, 'cases')


def gen_nan_regime():
    """Searches in which the reference's own scores are NaN: with logits spread over +-64 and most actions
    illegal, the masked root priors (shifted by the maximum over ALL logits, mcts/mcts.py + utils/game_utils.py:71-78)
    are so negative that the float32 softmax without max-subtraction (utils/utils.py:18-20) is 0 / 0."""
    rng = np.random.default_rng(244)
    A, S, H, B = 19, 9, 4, 48
    cfg = make_config(True, A, S, 1, 0.5, m=2)
    hp = hashnet.HashNetParams(A, H, 1, 0, 0)
    net = hashnet.HashNetwork(hp)
    legal = np.zeros((B, A))
    for b in range(B):
        legal[b, rng.choice(A, size=int(rng.integers(1, 4)), replace=False)] = 1
    codes = rng.integers(0, 2**32, size=B, dtype=np.uint64).astype(np.uint32)
    g = rng.gumbel(size=(B, A))
    out = dict(legal=legal, codes=codes, gumbel=g, meta=np.array([A, S, H, B, 2], dtype=np.int64))
    acts, vals, visits, probs, nan_roots = [], [], [], [], 0
    real_gumbel = np.random.gumbel
    try:
        for b in range(B):
            np.random.gumbel = lambda size=None, b=b: g[b].copy()
            m = CapturingMCTS(cfg)
            a, v, p = m.search(net, hashnet.observation_of(int(codes[b])), legal[b], lambda x: np.eye(A)[x], 0, 0)
            acts.append(a); vals.append(v); probs.append(p); visits.append(np.array(m.captured.children_visits[0]))
            nan_roots += int(np.isnan(p).any())
    finally:
        np.random.gumbel = real_gumbel
    out.update(action=np.array(acts, dtype=np.int64), root_value=np.array(vals), root_visits=np.array(visits), probs=np.array(probs))
    np.savez_compressed(os.path.join(HERE, 'nan_regime.npz'), **out)
    print('nan_regime.npz:', nan_roots, 'of', B, 'searches return a NaN policy')


def gen_trainer():
    """Three optimisation steps as Trainer.update_weights / update_weights_continuously perform them
    (trainer.py:43-45, 62-123; the actor itself needs ray), spelled out with the reference's own
    scalar_to_support / scale_gradient / update_lr on a small stand-in network (tests/tiny_net.py)."""
    sys.path.insert(0, os.path.join(REPO, 'tests'))
    from tiny_net import TinyNet
    out = {}
    for ci, (optimizer, A, L, U) in enumerate((('Adam', 2, 10, 5), ('SGD', 3, 4, 3))):
        cfg = SimpleNamespace(lr=0.007, lr_decay_rate=0.9, lr_decay_steps=10, weight_decay=1e-4, momentum=0.9,
                              support_limit=L, value_loss_weight=0.25, optimizer=optimizer)
        rng = np.random.default_rng(40 + ci)
        Bs = 12
        net = TinyNet(4, A, 8, L, seed=ci)
        net.train()
        out[f'c{ci}_meta'] = np.array([A, L, U, Bs, optimizer == 'Adam'], dtype=np.int64)
        for k, v in net.state_dict().items():
            out[f'c{ci}_w0_{k}'] = v.numpy().copy()
        if optimizer == 'Adam':
            opt = torch.optim.Adam(net.parameters(), lr=cfg.lr, weight_decay=cfg.weight_decay)
        else:
            opt = torch.optim.SGD(net.parameters(), lr=cfg.lr, momentum=cfg.momentum, weight_decay=cfg.weight_decay)
        losses = []
        for step in range(3):
            obs = torch.as_tensor(rng.normal(size=(Bs, 1, 1, 4)), dtype=torch.float32)
            acts = torch.as_tensor(np.eye(A)[rng.integers(0, A, size=(Bs, U + 1))], dtype=torch.float32)
            vt = torch.as_tensor(rng.normal(size=(Bs, U + 1)) * 3, dtype=torch.float32)
            rt = torch.as_tensor(rng.integers(0, 3, size=(Bs, U + 1)) * 0.5, dtype=torch.float32)
            pt = torch.as_tensor(rng.dirichlet([0.5] * A, size=(Bs, U + 1)), dtype=torch.float32)
            for n, x in (('obs', obs), ('acts', acts), ('vt', vt), ('rt', rt), ('pt', pt)):
                out[f'c{ci}_s{step}_{n}'] = x.numpy().copy()
            ref_nu.update_lr(cfg.lr, cfg.lr_decay_rate, cfg.lr_decay_steps, step, opt)
            vtb = ref_nu.scalar_to_support(vt, L)
            rtb = ref_nu.scalar_to_support(rt, L)
            policy_logits, hidden, value_logits = net.initial_inference(obs, False)
            predictions = [(1.0, value_logits, torch.zeros_like(value_logits, requires_grad=True), policy_logits)]
            for k in range(1, acts.shape[1]):
                policy_logits, hidden, value_logits, reward_logits = net.recurrent_inference(hidden, acts[:, k], False)
                ref_nu.scale_gradient(hidden, 0.5)
                predictions.append((1.0 / acts.shape[1], value_logits, reward_logits, policy_logits))
            value_loss, reward_loss, policy_loss = 0, 0, 0
            f = torch.nn.CrossEntropyLoss()
            for k, (scale, vl, rl, pl) in enumerate(predictions):
                v_k, r_k, p_k = f(vl, vtb[:, k]), f(rl, rtb[:, k]), f(pl, pt[:, k])
                ref_nu.scale_gradient(v_k, scale)
                ref_nu.scale_gradient(r_k, scale)
                ref_nu.scale_gradient(p_k, scale)
                if k == 0:
                    r_k = torch.zeros_like(v_k)
                value_loss += v_k
                reward_loss += r_k
                policy_loss += p_k
            loss = (value_loss * cfg.value_loss_weight + reward_loss + policy_loss) / 3
            opt.zero_grad()
            loss.backward()
            opt.step()
            losses.append([loss.item(), value_loss.item(), reward_loss.item(), policy_loss.item(), opt.param_groups[0]['lr']])
        out[f'c{ci}_losses'] = np.array(losses)
        for k, v in net.state_dict().items():
            out[f'c{ci}_w3_{k}'] = v.numpy().copy()
    out['n_cases'] = np.int64(2)
    np.savez_compressed(os.path.join(HERE, 'trainer.npz'), **out)
    print('trainer.npz')


if __name__ == '__main__':
    torch.set_num_threads(1)
    if len(sys.argv) > 1 and sys.argv[1] == 'trainer':
        gen_trainer()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'downsample':
        gen_downsample_fixture()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'stacking':
        gen_stacking()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'nan':
        gen_nan_regime()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'selfplay':
        gen_selfplay()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'reanalyse':
        gen_reanalyse()
        gen_targets()
        sys.exit(0)
    gen_selfplay()
    gen_reanalyse()
    gen_targets()
    gen_trainer()
    gen_stacking()
    gen_nan_regime()
    gen_misc()
    gen_hash_fixtures()
    gen_stream_fixtures()
    gen_network_fixtures()
    gen_downsample_fixture()
