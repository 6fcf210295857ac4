"""`MCTS`: the reference's search entry point (mcts/mcts.py:17-252) on the MI355X engine.

`MCTS(config).search(network, observation, legal_actions, action_encoder, to_play, temperature)`
keeps its signature and return value `(action, root_value, action_probs)`; `search_batch` is the
batched form the engine is built for.  All tree work (selection, expansion, backup, the final
action rule, root-prior preparation) runs in the HIP kernels of libmuzero_amd.so.

Random numbers are drawn on the host from the global `np.random` stream in the reference's
order, so a caller that seeds NumPy sees the same draws consumed:
  Gumbel: one `np.random.gumbel(size=A)` per search (mcts/mcts.py:211);
  PUCT:   one `np.random.dirichlet` (mcts/mcts.py:27), one `np.random.uniform(size=A)` per
          selection call (mcts/action_selection.py:46), one uniform for `np.random.choice`
          (mcts/action_selection.py:69).
"""
from __future__ import annotations

import os
import sys
from typing import Callable, Optional, Tuple

import numpy as np
import torch

try:
    from muzero_amd import engine as _engine
except ImportError:  # imported as the top-level package `mcts`
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    from muzero_amd import engine as _engine

from .tree import Tree


def _to_numpy(x):
    return x.detach().cpu().numpy() if torch.is_tensor(x) else np.asarray(x)


def network_device(network) -> torch.device:
    """Where a network object wants its inputs: its `device` attribute, else the device of its first parameter, else
    the host (plain Python objects such as the synthetic hash network)."""
    d = getattr(network, 'device', None)
    if d is not None:
        return torch.device(d)
    if isinstance(network, torch.nn.Module):
        for p in network.parameters():
            return p.device
    return torch.device('cpu')


class MCTS:

    def __init__(self, config) -> None:
        self.config = config
        self._engines = {}
        self._lowered = {}
        self.last_engine: Optional[_engine.SearchEngine] = None
        self._last_meta = None

    # ------------------------------------------------------------------ engine cache
    def _engine_for(self, hidden_size: int, batch: int, noise_mode: int, seed: int = 0):
        key = (hidden_size, noise_mode)
        eng = self._engines.get(key)
        if eng is None or eng.max_batch < batch:
            if eng is not None:
                eng.close()
            eng = _engine.SearchEngine(self.config, hidden_size, max(batch, 1), noise_mode=noise_mode, seed=seed)
            self._engines[key] = eng
        if eng.seed != (int(seed) & 0xFFFFFFFFFFFFFFFF):
            eng.set_seed(seed)   # the engine is reused across searches, the noise key is per search
        return eng

    # ------------------------------------------------------------------ search
    def search(self, network, observation: np.ndarray, legal_actions: np.ndarray,
               action_encoder: Callable, to_play: int, temperature: float = 0
               ) -> Tuple[int, float, np.ndarray]:
        """One position; consumes the global NumPy stream exactly like the reference."""
        actions, values, probs = self._search(network, np.asarray(observation)[None],
                                              np.asarray(legal_actions)[None], [to_play], temperature,
                                              action_encoder, stream_noise=True)
        return actions[0], values[0], probs[0]

    def search_batch(self, network, observations, legal_actions, to_play, temperature: float = 0,
                     action_encoder: Optional[Callable] = None, seed: int = 0, tree_id0: int = 0):
        """B independent positions at once.  Root noise (Gumbel / Dirichlet / the final sampling
        uniform) is drawn from `np.random` tree by tree (PUCT: Dirichlet row then sampling uniform, like
        consecutive reference searches); the per-selection tie-break uniforms are counter-based (`muzero_amd/noise.py`) keyed by (seed, tree_id0 + b)."""
        return self._search(network, observations, legal_actions, to_play, temperature, action_encoder,
                            stream_noise=False, seed=seed, tree_id0=tree_id0)

    def _auto_lower(self, network):
        """A reference network object (an eval-mode `nn.Module` whose `state_dict()` has the layout of
        network.py:133-186 / :405-479, described by this MCTS's config) is lowered once per weight
        version so that its searches take the single-launch paths; anything else is returned as is and
        evaluated between the select and expand kernels.  MUZERO_AMD_AUTOLOWER=0 turns this off."""
        cfg = self.config
        if hasattr(network, 'dyn') or os.environ.get('MUZERO_AMD_AUTOLOWER', '1') == '0':
            return network
        if not isinstance(network, torch.nn.Module) or network.training:
            return network
        if not all(hasattr(cfg, k) for k in ('network', 'observation_dim', 'support_limit', 'n_stacked_observations')):
            return network
        tensors = list(network.parameters()) + list(network.buffers())
        stamp = tuple((id(x), x._version) for x in tensors)
        cached = self._lowered.get(id(network))
        if cached is None or cached[0] != stamp:
            try:
                from muzero_amd.network import MuZeroDenseNetwork
                low = MuZeroDenseNetwork(cfg, network.state_dict())
            except (KeyError, ValueError, AttributeError, RuntimeError, IndexError):
                low = None  # not a layout this build lowers
            cached = self._lowered[id(network)] = (stamp, low)
        return cached[1] if cached[1] is not None else network

    @staticmethod
    def _lowered_plan(network, A: int):
        """How a lowered network (muzero_amd.network.MuZeroDenseNetwork) is evaluated inside the search
        kernels: ('fused', PackedNetwork) when its weights fit in LDS, ('mfma', (PackedMfma, io), the two root
        programs) for larger ones with at most 16 actions, None otherwise (or for any other network object).
        Packings are cached on the network and rebuilt when its weights change."""
        if not (hasattr(network, 'dyn') and hasattr(network, 'pred') and hasattr(network, 'rep')):
            return None
        stamp = id(getattr(network, 'state_dict_cpu', None))
        cached = getattr(network, '_mz_plan', None)
        if cached is not None and cached[0] == stamp:
            return cached[1]
        from muzero_amd.packing import pack_network, pack_network_mfma, pack_root_network_mfma
        plan = None
        # (a representation that torch evaluates - downsampled observation stacks - has no program to pack: the roots
        # come from network.initial_flat and the simulations run on the matrix cores)
        torch_rep = getattr(network.rep, 'steps', None) is None
        packed = None if torch_rep else pack_network(network)
        if packed is not None and max(p.weight_bytes() for p in (packed.recurrent, packed.represent, packed.predict)) <= 48 * 1024:
            plan = ('fused', packed)
        elif A <= 16:
            pm, io = pack_network_mfma(network)
            if pm.stride * 32 * 4 + 4096 <= 160 * 1024:
                plan = ('mfma', (pm, io), None if torch_rep else pack_root_network_mfma(network))
        network._mz_plan = (stamp, plan)
        return plan

    def _search(self, network, observations, legal_actions, to_play, temperature, action_encoder,
                stream_noise, seed=0, tree_id0=0):
        cfg = self.config
        A, S = int(cfg.n_actions), int(cfg.simulations)
        # (observations may already be a tensor on the device: batched callers build them there)
        obs = (observations.to(torch.float32) if isinstance(observations, torch.Tensor)
               else torch.as_tensor(np.asarray(observations), dtype=torch.float32))
        B = obs.shape[0]
        legal = np.asarray(legal_actions, dtype=np.float64).reshape(B, A)
        gumbel = bool(cfg.gumbel)
        network = self._auto_lower(network)
        noise_mode = _engine.NOISE_BUFFER if (stream_noise and not gumbel) else _engine.NOISE_COUNTER
        plan = self._lowered_plan(network, A)
        if plan is not None:
            # the root is evaluated on the device too: inside the first launch (fused) or level by level on the
            # matrix cores (mz_initial_inference_mfma)
            H, logits, hidden, value = int(network.H), None, None, None
        else:
            with torch.no_grad():
                logits, hidden, value = network.initial_inference(obs.to(network_device(network)))
            hidden_shape = tuple(hidden.shape[1:])
            H = int(np.prod(hidden_shape))
        eng = self._engine_for(H, B, noise_mode, seed)
        dev = eng.device
        if plan is not None and getattr(eng, '_plan_of', None) != (id(network), id(plan)):
            if plan[0] == 'fused':
                eng.set_network(plan[1])
            else:
                eng.set_network_mfma(*plan[1])
                if plan[2] is not None:
                    eng.set_root_network_mfma(*plan[2])
            eng._plan_of = (id(network), id(plan))
        if plan is not None and plan[0] == 'mfma':
            if plan[2] is not None:
                logits, hidden, value = eng.initial_inference_mfma(obs.reshape(B, -1))
            else:
                if network.device != dev:
                    network.to(dev)
                logits, hidden, value = network.initial_flat(obs.reshape(B, -1).to(dev))
        # Root noise from the global NumPy stream in the order B consecutive reference searches consume it (their
        # tie-break uniforms aside), drawn in ONE call per batch: a Python-level draw per tree costs more than the
        # search once batches reach thousands of trees.
        g = dirichlet = u_pre = None
        if gumbel:
            g = _engine.legacy_gumbel((B, A))     # row b = the b-th search's np.random.gumbel(size=A) (mcts/mcts.py:211)
        elif stream_noise:
            dirichlet, _ = _engine.legacy_root_noise(float(cfg.dirichlet_alpha), A, B, with_uniform=False)
        else:
            # per tree: Dirichlet row, then the uniform of the final np.random.choice
            dirichlet, u_pre = _engine.legacy_root_noise(float(cfg.dirichlet_alpha), A, B, with_uniform=True)
        frac = 0.0 if gumbel else cfg.exploration_frac
        tp = np.asarray(to_play, dtype=np.int32).reshape(B)
        if plan is not None and plan[0] == 'fused':
            eng.begin_net(obs.reshape(B, -1), legal, tp, dirichlet, frac, g, tree_id0=tree_id0)
        else:
            prior = eng.root_prepare(logits.reshape(B, A), legal, dirichlet, frac)
            eng.reset(prior, value.reshape(B), hidden.reshape(B, H), tp, legal, g, tree_id0=tree_id0)
        noise = rng_state = None
        if noise_mode == _engine.NOISE_BUFFER:
            # Pre-draw the longest possible run of per-selection rows (a chain tree makes
            # S(S+1)/2 calls), then rewind the stream to exactly the rows that were consumed.
            rng_state = np.random.get_state()
            noise = torch.as_tensor(np.random.uniform(size=(B, S * (S + 1) // 2, A)), device=dev)
        if plan is not None:   # every simulation inside one launch
            if plan[0] == 'fused':
                eng.search_net(noise)
            else:
                eng.search_mfma(noise)
        else:
            net_dev = getattr(network, 'device', None)
            for _ in range(S):
                parent, action, phid = eng.select(noise)
                with torch.no_grad():
                    if hasattr(network, 'recurrent_flat'):
                        h_in = phid if net_dev == dev else phid.to(network.device)
                        lg, nh, val, rew = network.recurrent_flat(h_in, network.encode_actions(action.to(h_in.device)))
                    else:
                        acts = action.cpu().numpy()
                        enc = torch.as_tensor(np.stack([np.asarray(action_encoder(int(a))) for a in acts]),
                                              dtype=torch.float32)
                        lg, nh, val, rew = network.recurrent_inference(phid.cpu().reshape(B, *hidden_shape), enc)
                eng.expand_backup(lg.reshape(B, A), val.reshape(B), rew.reshape(B), nh.reshape(B, H))
        u = None
        if not gumbel:
            if rng_state is not None:
                eng.check()
                _, calls = eng.stats()
                np.random.set_state(rng_state)
                for n in calls.cpu().numpy():  # advance past the rows the search used
                    np.random.uniform(size=(int(n), A))
            u = u_pre if u_pre is not None else np.random.random_sample(B)
        action, root_value, probs, _ = eng.to_act(float(temperature), u)
        self.last_engine = eng
        self._last_meta = (legal, g)
        a = action.cpu().numpy().astype(np.int64)
        v = root_value.cpu().numpy()
        p = probs.cpu().numpy()
        return a, v, p

    def last_tree(self, b: int = 0) -> Tree:
        """The tree of batch slot `b` from the most recent search, as host arrays."""
        eng = self.last_engine
        arrs = eng.export_tree(b)
        legal, g = self._last_meta
        hid = arrs.pop('hidden_states')
        return Tree(hidden_states=hid, root_legal_actions=legal[b], discount=self.config.gamma,
                    gumbel_noise=None if g is None else g[b], **arrs)

    # ------------------------------------------------------------------ single-tree methods
    # The reference exposes the steps of one search as methods that take and return a `Tree` of
    # host arrays mutated in place.  They are kept for callers that drive a search step by step;
    # arithmetic steps run through a one-tree engine context, array bookkeeping is plain assignment.

    def add_exploration_noise(self, prior_logits: np.ndarray) -> np.ndarray:
        """log((1 - f) * softmax(logits) + f * Dirichlet(alpha)) (mcts/mcts.py:22-31)."""
        A = len(prior_logits)
        noises = np.random.dirichlet([self.config.dirichlet_alpha] * A)
        eng = self._engine_for(0, 1, _engine.NOISE_NONE)
        out = eng.root_prepare(np.asarray(prior_logits, np.float32)[None], None, noises[None],
                               self.config.exploration_frac, shift_and_mask=False)
        return out[0].cpu().numpy()

    def update(self, attr, v, index):
        attr[index] = v
        return attr

    def update_tree_node(self, tree: Tree, node_index: int, prior_logits, hidden_state, value,
                         to_play: int) -> Tree:
        """Writes one node's evaluation into the arrays (mcts/mcts.py:37-57)."""
        tree.children_prior_logits[node_index] = np.asarray(prior_logits).reshape(-1)
        tree.node_visits[node_index] += 1
        v = np.asarray(value).reshape(-1)[0]
        tree.node_values[node_index] = v
        tree.node_raw_values[node_index] = v
        tree.hidden_states[node_index] = hidden_state
        tree.to_play[node_index] = to_play
        return tree

    def initialize_tree(self, prior_logits, hidden_state, value, to_play: int, root_legal_actions,
                        gumbel_noise) -> Tree:
        """Fresh arrays for simulations+1 nodes with the root filled in (mcts/mcts.py:59-96)."""
        N, A = self.config.simulations + 1, self.config.n_actions
        value = np.asarray(value)
        prior_logits = np.asarray(prior_logits)
        hidden_state = np.asarray(hidden_state)
        tree = Tree(
            node_visits=np.zeros(N, np.int32), node_values=np.zeros(N, value.dtype),
            node_raw_values=np.zeros(N, value.dtype), parents=np.full(N, Tree.NO_PARENT, np.int32),
            action_from_parent=np.full(N, Tree.NO_PARENT, np.int32),
            children_index=np.full((N, A), Tree.UNVISITED, np.int32),
            children_prior_logits=np.zeros((N, A), prior_logits.dtype),
            children_visits=np.zeros((N, A), np.int32), children_rewards=np.zeros((N, A), value.dtype),
            children_values=np.zeros((N, A), value.dtype),
            hidden_states=np.zeros((N,) + hidden_state.shape, hidden_state.dtype),
            to_play=np.full(N, Tree.P1, np.int32), root_legal_actions=root_legal_actions,
            discount=self.config.gamma, gumbel_noise=gumbel_noise)
        return self.update_tree_node(tree, Tree.ROOT_INDEX, prior_logits, hidden_state, value, to_play)

    def simulate(self, tree: Tree, action_selection_func: Callable) -> Tuple[int, int, int]:
        """Descends with `action_selection_func(tree, node, depth)` until an unexpanded edge
        (mcts/mcts.py:98-114); `next_to_play` starts from player 0 whatever the root's to_play is."""
        nxt, to_play, depth = Tree.ROOT_INDEX, Tree.P1, 0
        while nxt != Tree.UNVISITED:
            node = nxt
            action = action_selection_func(tree, node, depth)
            nxt = tree.children_index[node, action]
            if self.config.players not in (1, 2):
                raise AssertionError('The project supports 1P and 2P games only.')
            to_play = to_play if self.config.players == 1 else 1 - to_play
            depth += 1
        return node, action, to_play

    def expand(self, tree: Tree, node_index: int, action: int, next_node_index: int, prior_logits,
               hidden_state, reward, value, next_to_play: int) -> Tree:
        """Creates the child reached by `action` (mcts/mcts.py:116-141)."""
        self.update_tree_node(tree, next_node_index, prior_logits, hidden_state, value, next_to_play)
        tree.parents[next_node_index] = node_index
        tree.action_from_parent[next_node_index] = action
        tree.children_index[node_index, action] = next_node_index
        tree.children_rewards[node_index, action] = np.asarray(reward).reshape(-1)[0]
        return tree

    def backward(self, tree: Tree, leaf_index: int, to_play: int) -> Tree:
        """Backs the leaf value up to the root on the device and writes the result into the same
        host arrays (mcts/mcts.py:143-164)."""
        from . import _node_engine
        gumbel = tree.children_prior_logits.dtype == np.float32
        eng = _node_engine.load(tree, gumbel)
        eng.node_backward(0, leaf_index, to_play)
        out = eng.export_tree(0)
        for k in ('node_visits', 'node_values', 'children_values', 'children_visits'):
            getattr(tree, k)[...] = out[k].reshape(getattr(tree, k).shape)
        return tree

    def visualize_tree(self, tree: Tree, output_file: str = None) -> None:
        """Draws the tree with graphviz (mcts/mcts.py:254-284); needs the optional `pygraphviz`."""
        import pygraphviz  # imported here: nothing else depends on it

        def node_label(i, reward=0.0):
            return (f'{i}\nReward: {reward:.2f}\nValue: {tree.node_values[i]:.2f}\n'
                    f'Visits: {tree.node_visits[i]}\nTo play: {tree.to_play[i]}')
        graph = pygraphviz.AGraph(directed=True)
        graph.add_node(0, label=node_label(0), color='green')
        for n in range(tree.n_simulations):
            e = np.exp(tree.children_prior_logits[n].astype(np.float64))
            probs = e / e.sum()
            q = tree.qvalues(n)
            for a in range(tree.n_actions):
                c = tree.children_index[n, a]
                if c >= 0:
                    graph.add_node(c, label=node_label(c, tree.children_rewards[n, a]), color='red')
                    graph.add_edge(n, c, label=f'{a}\nQ: {q[a]:.2f}\np: {probs[a]:.2f}')
        graph.draw(output_file if output_file is not None else '/tmp/search_tree.png', prog='dot')
