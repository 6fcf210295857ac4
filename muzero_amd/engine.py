"""ctypes binding of libmuzero_amd.so and the batched search engine built on it.

PyTorch is used for device memory and streams only; every search step runs in
the HIP kernels of muzero_amd/csrc.  There is no CPU fallback: if the shared
library is missing, importing this module's engine raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.environ.get('MUZERO_AMD_LIB') or os.path.join(_HERE, 'libmuzero_amd.so')
_lib = None

NOISE_COUNTER, NOISE_BUFFER, NOISE_NONE = 0, 1, 2
NET_HASH, NET_DENSE, NET_MFMA = 0, 1, 2


class EngineError(RuntimeError):
    pass


class MzConfig(C.Structure):
    _fields_ = [('n_actions', C.c_int32), ('n_simulations', C.c_int32), ('players', C.c_int32),
                ('gumbel', C.c_int32), ('hidden_size', C.c_int32), ('max_batch', C.c_int32),
                ('discount', C.c_double), ('c_base', C.c_double), ('c_init', C.c_double),
                ('max_considered_actions', C.c_int32), ('noise_mode', C.c_int32),
                ('c_visit', C.c_double), ('c_scale', C.c_double), ('seed', C.c_uint64),
                ('c_table_h', C.c_void_p), ('device', C.c_int32), ('lanes_per_tree', C.c_int32)]


class MzTreeHost(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in (
        'node_visits', 'node_values', 'node_raw_values', 'parents', 'action_from_parent', 'to_play',
        'children_index', 'children_prior_logits', 'children_visits', 'children_rewards',
        'children_values', 'hidden_states')]


class MzEnvState(C.Structure):
    _fields_ = [('kind', C.c_int32), ('max_moves', C.c_int32), ('stale_observation', C.c_int32),
                ('obs_size', C.c_int32), ('state', C.c_void_p), ('mover', C.c_void_p), ('to_play', C.c_void_p),
                ('legal', C.c_void_p), ('cur_obs', C.c_void_p), ('search_obs', C.c_void_p), ('moves', C.c_void_p),
                ('reset_state', C.c_void_p), ('n_stacked_observations', C.c_int32), ('stack_action', C.c_int32)]


class MzMoveLog(C.Structure):
    _fields_ = [('capacity', C.c_int32), ('n_envs', C.c_int32), ('obs_size', C.c_int32), ('n_actions', C.c_int32),
                ('observations', C.c_void_p), ('actions', C.c_void_p), ('rewards', C.c_void_p), ('to_plays', C.c_void_p),
                ('root_values', C.c_void_p), ('probs', C.c_void_p), ('done', C.c_void_p), ('open_at_start', C.c_void_p)]


ENV_TICTACTOE, ENV_CARTPOLE = 1, 2

EXPORTS = ['mz_create', 'mz_destroy', 'mz_last_error', 'mz_set_seed', 'mz_check', 'mz_root_prepare', 'mz_tree_reset', 'mz_select',
           'mz_select_outputs', 'mz_expand_backup', 'mz_to_act', 'mz_search_run', 'mz_search_run_act', 'mz_search_move', 'mz_set_network', 'mz_set_root_network', 'mz_search_begin_net', 'mz_set_network_mfma', 'mz_set_root_network_mfma', 'mz_initial_inference_mfma', 'mz_expand_backup_net', 'mz_expand_backup_net_static', 'mz_act_env_step', 'mz_act_env_step_log', 'mz_log_count_games', 'mz_log_list_games', 'mz_tree_export',
           'mz_stats', 'mz_node_qtransform', 'mz_node_select', 'mz_node_backward', 'mz_tree_import', 'mz_considered_visits', 'mz_legacy_root_noise', 'mz_legacy_gumbel', 'mz_make_targets', 'mz_scalar_to_support',
           'mz_version']


def lib():
    """The loaded shared library; raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise EngineError(f'{_LIB_PATH} is missing: build it with `python -m muzero_amd.build` '
                              '(hipcc, gfx950). There is no CPU fallback.')
        L = C.CDLL(_LIB_PATH)
        L.mz_create.restype = C.c_void_p
        L.mz_create.argtypes = [C.POINTER(MzConfig)]
        L.mz_destroy.argtypes = [C.c_void_p]
        L.mz_destroy.restype = None
        L.mz_last_error.restype = C.c_char_p
        vp, i32, f64, u64 = C.c_void_p, C.c_int32, C.c_double, C.c_uint64
        L.mz_set_seed.argtypes = [vp, u64]
        L.mz_check.argtypes = [vp, vp]
        L.mz_root_prepare.argtypes = [vp, i32, vp, vp, f64, vp, i32, vp, vp]
        L.mz_node_backward.argtypes = [vp, i32, i32, i32, vp]
        L.mz_tree_reset.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp, u64, vp]
        L.mz_select.argtypes = [vp, vp, i32, vp, vp, vp, vp]
        L.mz_select_outputs.argtypes = [vp, vp, vp, vp, vp]
        L.mz_expand_backup.argtypes = [vp, vp, vp, vp, vp, vp]
        L.mz_to_act.argtypes = [vp, f64, vp, vp, vp, vp, vp, vp]
        L.mz_search_run.argtypes = [vp, i32, vp, vp, i32, vp]
        L.mz_search_move.argtypes = [vp, i32, vp, vp, vp, vp, C.c_double, vp, C.c_uint64, vp, i32, C.c_double, vp, vp, vp, vp, vp, vp]
        L.mz_search_run_act.argtypes = [vp, i32, vp, vp, i32, C.c_double, vp, vp, vp, vp, vp, vp]
        L.mz_set_network.argtypes = [vp, vp, i32, vp, i32, vp]
        L.mz_set_root_network.argtypes = [vp, vp, i32, vp, i32, vp, i32, vp, i32, vp]
        L.mz_set_network_mfma.argtypes = [vp, vp, i32, i32, vp, i32, vp, C.c_int64, vp, i32, vp, i32, vp, i32, i32, i32, vp]
        L.mz_set_root_network_mfma.argtypes = [vp, i32, vp, i32, i32, vp, i32, vp, C.c_int64, vp, i32, vp, i32, vp, i32, i32, i32, vp]
        L.mz_initial_inference_mfma.argtypes = [vp, i32, vp, vp, vp, vp, vp]
        L.mz_expand_backup_net.argtypes = [vp, i32, vp, i32, vp]
        L.mz_expand_backup_net_static.argtypes = [vp, i32, vp, i32, vp]
        L.mz_act_env_step.argtypes = [vp, C.POINTER(MzEnvState), f64, vp, vp, vp, vp, vp, vp, vp, vp, vp]
        L.mz_act_env_step_log.argtypes = [vp, C.POINTER(MzEnvState), C.POINTER(MzMoveLog), i32, f64, vp, vp, vp, vp, vp, vp,
                                          vp, vp, vp]
        L.mz_log_count_games.argtypes = [C.POINTER(MzMoveLog), i32, vp, vp]
        L.mz_log_list_games.argtypes = [C.POINTER(MzMoveLog), i32, vp, vp, vp, vp]
        L.mz_search_begin_net.argtypes = [vp, i32, vp, vp, vp, vp, f64, vp, u64, vp, vp, vp]
        L.mz_tree_export.argtypes = [vp, i32, C.POINTER(MzTreeHost), vp]
        L.mz_tree_import.argtypes = [vp, i32, C.POINTER(MzTreeHost), vp, vp, vp]
        L.mz_stats.argtypes = [vp, vp, vp, vp]
        L.mz_legacy_root_noise.argtypes = [vp, vp, C.c_double, i32, i32, i32, vp, vp]
        L.mz_legacy_gumbel.argtypes = [vp, vp, C.c_int64, vp]
        L.mz_node_qtransform.argtypes = [vp, i32, i32, i32, vp, vp]
        L.mz_node_select.argtypes = [vp, i32, i32, i32, vp, vp, vp]
        L.mz_considered_visits.argtypes = [i32, i32, vp]
        _lib = L
    return _lib


def _check(rc, what=''):
    if rc != 0:
        msg = lib().mz_last_error().decode()
        if rc == -1:
            raise ValueError(msg)
        raise EngineError(msg or f'{what} failed with code {rc}')


def considered_visits(max_considered_actions: int, simulations: int) -> tuple:
    """Sequential-halving visit schedule (host entry point of the library)."""
    out = np.zeros(max(int(simulations), 0), np.int32)
    _check(lib().mz_considered_visits(int(max_considered_actions), int(simulations),
                                      out.ctypes.data_as(C.c_void_p)))
    return tuple(int(x) for x in out)


def legacy_root_noise(alpha: float, n_actions: int, B: int, with_uniform: bool):
    """Root noise of B consecutive PUCT searches from the GLOBAL NumPy stream, in one native call: per search one
    `np.random.dirichlet([alpha] * A)` row (mcts/mcts.py:27) and, with_uniform, the `random_sample()` behind the final
    `np.random.choice` (mcts/action_selection.py:69), in that order.  The stream is left where B such searches would
    leave it.  Falls back to drawing with NumPy row by row where the native generator does not apply (alpha >= 1, a
    global generator that is not MT19937, a pending cached Gaussian is kept either way).
    Returns (dirichlet f64[B, A], u f64[B] or None)."""
    A = int(n_actions)
    state = np.random.get_state()
    if not (0.0 < alpha < 1.0) or state[0] != 'MT19937':
        di, u = np.empty((B, A)), (np.empty(B) if with_uniform else None)
        for b in range(B):
            di[b] = np.random.dirichlet([alpha] * A)
            if with_uniform:
                u[b] = np.random.random_sample()
        return di, u
    key = np.ascontiguousarray(state[1], dtype=np.uint32).copy()
    pos = C.c_int32(int(state[2]))
    di = np.empty((B, A), np.float64)
    u = np.empty(B, np.float64) if with_uniform else None
    _check(lib().mz_legacy_root_noise(key.ctypes.data_as(C.c_void_p), C.byref(pos), float(alpha), A, int(B),
                                      1 if with_uniform else 0, di.ctypes.data_as(C.c_void_p),
                                      u.ctypes.data_as(C.c_void_p) if with_uniform else None), 'mz_legacy_root_noise')
    np.random.set_state((state[0], key, int(pos.value), state[3], state[4]))
    return di, u


def legacy_gumbel(shape):
    """np.random.gumbel(size=shape) from the GLOBAL NumPy stream in one native call (the stream is left where NumPy
    would leave it): the uniforms come off the Mersenne Twister in order, the logarithms are taken on a few host threads.
    For the root noise of thousands of Gumbel searches NumPy's single-threaded draw costs as much as a tenth of the
    search itself.  Falls back to NumPy when the global generator is not MT19937 or the draw is small."""
    n = int(np.prod(shape))
    state = np.random.get_state()
    if state[0] != 'MT19937' or n < 16384:
        return np.random.gumbel(size=shape)
    key = np.ascontiguousarray(state[1], dtype=np.uint32).copy()
    pos = C.c_int32(int(state[2]))
    out = np.empty(n, np.float64)
    _check(lib().mz_legacy_gumbel(key.ctypes.data_as(C.c_void_p), C.byref(pos), C.c_int64(n),
                                  out.ctypes.data_as(C.c_void_p)), 'mz_legacy_gumbel')
    np.random.set_state((state[0], key, int(pos.value), state[3], state[4]))
    return out.reshape(shape)


def c_table_numpy(config) -> np.ndarray:
    """c(n) for n = 0..S+1, evaluated with NumPy exactly as mcts/action_selection.py:43 does on an
    np.int32 visit count, so the exploration constant matches the reference to the last bit."""
    n = np.arange(int(config.simulations) + 2, dtype=np.int32)
    return np.array([config.c_init + np.log((k + config.c_base + 1) / config.c_base) for k in n],
                    dtype=np.float64)


def _ptr(t):
    if t is None:
        return None
    return C.c_void_p(t.data_ptr())


def _stream():
    """The current torch stream of the current device as a raw hipStream_t (the private accessor is an order of
    magnitude cheaper than torch.cuda.current_stream(), which dominated the host cost of a launch)."""
    raw = getattr(torch._C, '_cuda_getCurrentRawStream', None)
    if raw is not None:
        return C.c_void_p(raw(torch.cuda.current_device()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class SearchEngine:
    """B independent search trees resident on one GPU.

    config: an object with the attributes `MCTS` reads from its config (simulations, n_actions, gamma,
    players, gumbel, and c_base/c_init or max_considered_actions/c_visit/c_scale).
    """

    def __init__(self, config, hidden_size: int, max_batch: int, device=None, noise_mode=NOISE_COUNTER,
                 seed: int = 0, lanes_per_tree: int = 0, c_table=None):
        if not torch.cuda.is_available():
            raise EngineError('no GPU visible: the search engine runs on MI355X only')
        self.config = config
        self.device = torch.device('cuda', torch.cuda.current_device() if device is None else
                                   (device.index if isinstance(device, torch.device) else int(device)))
        self.A, self.S, self.H = int(config.n_actions), int(config.simulations), int(hidden_size)
        self.N = self.S + 1
        self.gumbel = bool(config.gumbel)
        self.max_batch = int(max_batch)
        self.noise_mode = noise_mode
        g = self.gumbel
        self._c_table = None
        if not g:
            self._c_table = np.ascontiguousarray(c_table_numpy(config) if c_table is None else c_table,
                                                 dtype=np.float64)
            assert self._c_table.shape == (self.S + 2,)
        cfg = MzConfig(self.A, self.S, int(config.players), int(g), self.H, self.max_batch,
                       float(config.gamma), 0.0 if g else float(config.c_base),
                       0.0 if g else float(config.c_init),
                       int(config.max_considered_actions) if g else 0, int(noise_mode),
                       float(config.c_visit) if g else 0.0, float(config.c_scale) if g else 0.0,
                       int(seed) & 0xFFFFFFFFFFFFFFFF,
                       None if g else self._c_table.ctypes.data_as(C.c_void_p),
                       self.device.index, int(lanes_per_tree))
        self._ctx = lib().mz_create(C.byref(cfg))
        if not self._ctx:
            msg = lib().mz_last_error().decode()
            raise ValueError(msg) if 'mz_create:' in msg and 'hip' not in msg.lower() else EngineError(msg)
        self.B = 0
        self.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        self.prior_dtype = torch.float32 if g else torch.float64
        self._parent = self._action = self._phid = None

    def close(self):
        if getattr(self, '_ctx', None):
            lib().mz_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_seed(self, seed: int):
        """Key of the counter-based tie-break noise for the searches that follow."""
        _check(lib().mz_set_seed(self._ctx, C.c_uint64(int(seed) & 0xFFFFFFFFFFFFFFFF)))
        self.seed = int(seed) & 0xFFFFFFFFFFFFFFFF

    # ------------------------------------------------------------------ helpers
    def _dev(self, x, dtype, shape=None):
        t = torch.as_tensor(x)
        if t.dtype != dtype or t.device != self.device:
            t = t.to(device=self.device, dtype=dtype)
        t = t.contiguous()
        if shape is not None and tuple(t.shape) != tuple(shape):
            raise ValueError(f'expected shape {tuple(shape)}, got {tuple(t.shape)}')
        return t

    # ------------------------------------------------------------------ steps
    def root_prepare(self, logits, legal, dirichlet=None, exploration_frac=0.0, shift_and_mask=True):
        lg = self._dev(logits, torch.float32)
        B = lg.shape[0]
        lg = lg.reshape(B, self.A)
        le = None if legal is None else self._dev(
            np.asarray(legal) != 0 if not torch.is_tensor(legal) else legal != 0, torch.uint8, (B, self.A))
        di = None if dirichlet is None else self._dev(dirichlet, torch.float64, (B, self.A))
        out = torch.empty((B, self.A), dtype=self.prior_dtype, device=self.device)
        _check(lib().mz_root_prepare(self._ctx, B, _ptr(lg), _ptr(di), float(exploration_frac), _ptr(le),
                                     int(bool(shift_and_mask)), _ptr(out), _stream()))
        return out

    def reset(self, root_prior, root_value, root_hidden, root_to_play, legal, gumbel=None, tree_id0=0):
        rp = self._dev(root_prior, self.prior_dtype)
        B = rp.shape[0]
        rp = rp.reshape(B, self.A)
        rv = self._dev(root_value, torch.float64).reshape(B)
        rh = None if root_hidden is None or self.H == 0 else self._dev(root_hidden, torch.float32).reshape(B, self.H)
        tp = self._dev(root_to_play, torch.int32).reshape(B)
        le = self._dev(np.asarray(legal) != 0 if not torch.is_tensor(legal) else legal != 0, torch.uint8, (B, self.A))
        gu = None if gumbel is None else self._dev(gumbel, torch.float64, (B, self.A))
        _check(lib().mz_tree_reset(self._ctx, B, _ptr(rp), _ptr(rv), _ptr(rh), _ptr(tp), _ptr(le), _ptr(gu),
                                   C.c_uint64(int(tree_id0)), _stream()))
        self.B = B
        self._parent = torch.empty(B, dtype=torch.int32, device=self.device)
        self._action = torch.empty(B, dtype=torch.int32, device=self.device)
        self._phid = torch.empty((B, self.H), dtype=torch.float32, device=self.device) if self.H else None

    def select(self, noise=None, want_hidden=True):
        """One descent per tree.  Returns (parent i32[B], action i32[B], parent_hidden f32[B,H])."""
        nz, max_calls = None, 0
        if noise is not None:
            nz = self._dev(noise, torch.float64)
            max_calls = nz.numel() // (self.B * self.A)
            self._noise_keep = nz
        ph = self._phid if want_hidden else None
        _check(lib().mz_select(self._ctx, _ptr(nz), max_calls, _ptr(ph), _ptr(self._parent),
                               _ptr(self._action), _stream()))
        return self._parent, self._action, ph

    def expand_backup(self, logits, value, reward, next_hidden=None):
        B = self.B
        lg = self._dev(logits, torch.float32).reshape(B, self.A)
        va = self._dev(value, torch.float64).reshape(B)
        re = self._dev(reward, torch.float64).reshape(B)
        nh = None if next_hidden is None or self.H == 0 else self._dev(next_hidden, torch.float32).reshape(B, self.H)
        _check(lib().mz_expand_backup(self._ctx, _ptr(lg), _ptr(va), _ptr(re), _ptr(nh), _stream()))

    def search_hash(self, logit_shift=5, reward_on=1, parent_predicts=1, noise=None):
        """All simulations in one launch with the built-in synthetic hash network."""
        params = (C.c_int32 * 3)(int(logit_shift), int(reward_on), int(parent_predicts))
        nz, max_calls = None, 0
        if noise is not None:
            nz = self._dev(noise, torch.float64)
            max_calls = nz.numel() // (self.B * self.A)
        _check(lib().mz_search_run(self._ctx, NET_HASH, params, _ptr(nz), max_calls, _stream()))

    def set_network(self, packed):
        """Uploads a PackedNetwork (muzero_amd.packing) as the leaf evaluator of `search_net`."""
        r, io = packed.recurrent, packed.io()
        self._net_keep = (r, io)
        _check(lib().mz_set_network(self._ctx, r.meta.ctypes.data_as(C.c_void_p), int(r.meta.size),
                                    r.data.ctypes.data_as(C.c_void_p), int(r.data.size),
                                    io.ctypes.data_as(C.c_void_p)))

        rp, pp = packed.represent, packed.predict
        rio = np.asarray([rp.slots['obs'], rp.slots['hidden_raw'], pp.slots['hidden'], pp.slots['policy_logits'],
                          pp.slots['value_logits'], packed.obs_size, 2 * packed.support_limit + 1,
                          packed.norm_groups], np.int32)
        self._root_keep = (rp, pp, rio)
        vp = lambda x: x.ctypes.data_as(C.c_void_p)  # noqa: E731
        _check(lib().mz_set_root_network(self._ctx, vp(rp.meta), int(rp.meta.size), vp(rp.data), int(rp.data.size),
                                         vp(pp.meta), int(pp.meta.size), vp(pp.data), int(pp.data.size), vp(rio)))
        self.obs_size = packed.obs_size

    def begin_net(self, obs, legal, to_play, dirichlet=None, exploration_frac=0.0, gumbel=None, tree_id0=0,
                  want_outputs=False):
        """Root of B searches in one launch: initial inference, root-prior preparation, root rows."""
        ob = self._dev(obs, torch.float32)
        B = ob.shape[0]
        ob = ob.reshape(B, self.obs_size)
        le = legal if (torch.is_tensor(legal) and legal.dtype == torch.uint8 and legal.device == self.device) else \
            self._dev(np.asarray(legal) != 0 if not torch.is_tensor(legal) else legal != 0, torch.uint8, (B, self.A))
        tp = self._dev(to_play, torch.int32).reshape(B)
        di = None if dirichlet is None else self._dev(dirichlet, torch.float64, (B, self.A))
        gu = None if gumbel is None else self._dev(gumbel, torch.float64, (B, self.A))
        ol = torch.empty((B, self.A), dtype=torch.float32, device=self.device) if want_outputs else None
        ov = torch.empty(B, dtype=torch.float64, device=self.device) if want_outputs else None
        _check(lib().mz_search_begin_net(self._ctx, B, _ptr(ob), _ptr(le), _ptr(tp), _ptr(di),
                                         float(exploration_frac), _ptr(gu), C.c_uint64(int(tree_id0)),
                                         _ptr(ol), _ptr(ov), _stream()))
        self.B = B
        if self._parent is None or self._parent.shape[0] != B:
            self._parent = torch.empty(B, dtype=torch.int32, device=self.device)
            self._action = torch.empty(B, dtype=torch.int32, device=self.device)
            self._phid = torch.empty((B, self.H), dtype=torch.float32, device=self.device) if self.H else None
        return ol, ov

    def set_network_mfma(self, pm, io):
        """Uploads a network in matrix-core layout (muzero_amd.packing.pack_network_mfma)."""
        self._mfma_keep = (pm, io)
        vp = lambda x: x.ctypes.data_as(C.c_void_p)  # noqa: E731
        _check(lib().mz_set_network_mfma(self._ctx, vp(pm.steps), pm.steps.size // 12, int(pm.n_levels),
                                         vp(pm.sched), int(pm.sched.size), vp(pm.wfrag),
                                         C.c_int64(pm.wfrag.size), vp(pm.bias), int(pm.bias.size),
                                         vp(pm.ctinfo), pm.ctinfo.size // 3, vp(pm.kidx), int(pm.kidx.size),
                                         int(pm.stride), int(pm.plane_major), vp(io)))

    def set_root_network_mfma(self, rep, pred):
        """Uploads the representation and prediction programs of the initial inference in matrix-core layout
        (muzero_amd.packing.pack_root_network_mfma: two (program, io) pairs)."""
        self._root_mfma_keep = (rep, pred)
        vp = lambda x: x.ctypes.data_as(C.c_void_p)  # noqa: E731
        for part, (pm, io) in enumerate((rep, pred)):
            _check(lib().mz_set_root_network_mfma(self._ctx, part, vp(pm.steps), pm.steps.size // 12, int(pm.n_levels),
                                                  vp(pm.sched), int(pm.sched.size), vp(pm.wfrag),
                                                  C.c_int64(pm.wfrag.size), vp(pm.bias), int(pm.bias.size),
                                                  vp(pm.ctinfo), pm.ctinfo.size // 3, vp(pm.kidx), int(pm.kidx.size),
                                                  int(pm.stride), int(pm.plane_major), vp(io)))

    def initial_inference_mfma(self, obs):
        """network.initial_inference (network.py:86-103) of B observations on the device, through the programs given to
        set_root_network_mfma: (policy logits f32[B, A], hidden states f32[B, H], values f64[B]), device tensors."""
        obs = self._dev(obs, torch.float32)
        B = obs.shape[0]
        obs = obs.reshape(B, -1).contiguous()
        logits = torch.empty((B, self.A), dtype=torch.float32, device=self.device)
        hidden = torch.empty((B, self.H), dtype=torch.float32, device=self.device)
        value = torch.empty(B, dtype=torch.float64, device=self.device)
        _check(lib().mz_initial_inference_mfma(self._ctx, B, _ptr(obs), _ptr(logits), _ptr(hidden), _ptr(value), _stream()))
        return logits, hidden, value

    def expand_backup_net(self, select_next: bool = False, noise=None, static_scratch: bool = False):
        """Leaf evaluation on the matrix cores + expansion + backup for the pending selection; with
        `select_next` the next simulation's selection runs in the same launch.  `static_scratch`: stage
        through the library's fixed device array (mz_expand_backup_net_static; no allocation, 16 MiB limit)."""
        nz, max_calls = None, 0
        if noise is not None:
            nz = self._dev(noise, torch.float64)
            max_calls = nz.numel() // (self.B * self.A)
        fn = lib().mz_expand_backup_net_static if static_scratch else lib().mz_expand_backup_net
        _check(fn(self._ctx, C.c_int32(int(bool(select_next))), _ptr(nz), max_calls, _stream()))

    def search_net(self, noise=None):
        """All simulations in one launch, leaves evaluated by the uploaded network."""
        nz, max_calls = None, 0
        if noise is not None:
            nz = self._dev(noise, torch.float64)
            max_calls = nz.numel() // (self.B * self.A)
        _check(lib().mz_search_run(self._ctx, NET_DENSE, None, _ptr(nz), max_calls, _stream()))

    def search_move(self, obs, legal, to_play, dirichlet=None, exploration_frac=0.0, gumbel=None, tree_id0=0,
                    temperature=0.0, u=None, noise=None):
        """begin_net + search_net + to_act as one call (two launches for small batches of 2-action PUCT trees,
        else three).  Returns what to_act returns."""
        ob = self._dev(obs, torch.float32)
        B = ob.shape[0]
        ob = ob.reshape(B, self.obs_size)
        le = legal if (torch.is_tensor(legal) and legal.dtype == torch.uint8 and legal.device == self.device) else \
            self._dev(np.asarray(legal) != 0 if not torch.is_tensor(legal) else legal != 0, torch.uint8, (B, self.A))
        tp = self._dev(to_play, torch.int32).reshape(B)
        di = None if dirichlet is None else self._dev(dirichlet, torch.float64, (B, self.A))
        gu = None if gumbel is None else self._dev(gumbel, torch.float64, (B, self.A))
        self.B = B
        nz, max_calls = None, 0
        if noise is not None:
            nz = self._dev(noise, torch.float64)
            max_calls = nz.numel() // (B * self.A)
        action, value, probs, q, uu = self._act_buffers(u)
        _check(lib().mz_search_move(self._ctx, B, _ptr(ob), _ptr(le), _ptr(tp), _ptr(di), float(exploration_frac), _ptr(gu),
                                    C.c_uint64(int(tree_id0)), _ptr(nz), max_calls, float(temperature), _ptr(uu),
                                    _ptr(action), _ptr(probs), _ptr(value), _ptr(q), _stream()))
        if self._parent is None or self._parent.shape[0] != B:
            self._parent = torch.empty(B, dtype=torch.int32, device=self.device)
            self._action = torch.empty(B, dtype=torch.int32, device=self.device)
            self._phid = torch.empty((B, self.H), dtype=torch.float32, device=self.device) if self.H else None
        return action, value, probs, q

    def search_net_act(self, temperature=0.0, u=None, noise=None):
        """search_net followed by to_act as one call (one launch where the search kernel can run the final rule
        itself).  Returns what to_act returns."""
        nz, max_calls = None, 0
        if noise is not None:
            nz = self._dev(noise, torch.float64)
            max_calls = nz.numel() // (self.B * self.A)
        action, value, probs, q, uu = self._act_buffers(u)
        _check(lib().mz_search_run_act(self._ctx, NET_DENSE, None, _ptr(nz), max_calls, float(temperature), _ptr(uu),
                                       _ptr(action), _ptr(probs), _ptr(value), _ptr(q), _stream()))
        return action, value, probs, q

    def search_mfma(self, noise=None):
        """All simulations in one launch, leaves evaluated on the matrix cores (set_network_mfma)."""
        nz, max_calls = None, 0
        if noise is not None:
            nz = self._dev(noise, torch.float64)
            max_calls = nz.numel() // (self.B * self.A)
        _check(lib().mz_search_run(self._ctx, NET_MFMA, None, _ptr(nz), max_calls, _stream()))

    def _act_buffers(self, u):
        B = self.B
        uu = None if u is None else self._dev(u, torch.float64).reshape(B)
        if getattr(self, '_act_out', None) is None or self._act_out[0].shape[0] != B:
            self._act_out = (torch.empty(B, dtype=torch.int32, device=self.device),
                             torch.empty(B, dtype=torch.float64, device=self.device),
                             torch.empty((B, self.A), dtype=torch.float64, device=self.device),
                             torch.empty((B, self.A), dtype=torch.float64, device=self.device))
        return (*self._act_out, uu)

    def to_act(self, temperature=0.0, u=None):
        action, value, probs, q, uu = self._act_buffers(u)
        _check(lib().mz_to_act(self._ctx, float(temperature), _ptr(uu), _ptr(action), _ptr(probs), _ptr(value),
                               _ptr(q), _stream()))
        return action, value, probs, q

    def act_env_step(self, env_struct, temperature, u, out, log=None, row=0):
        """Final action rule + environment step + trajectory record + restart of finished games.
        `out`: dict of preallocated device tensors action/probs/value/obs_before/reward/done/to_play (with a
        log it may be None or hold only some of them); log: MzMoveLog whose row `row` receives the records."""
        uu = None if u is None else self._dev(u, torch.float64).reshape(self.B)
        self._u_keep = uu
        out = out or {}
        outs = [_ptr(out.get(k)) for k in ('action', 'probs', 'value', 'obs_before', 'reward', 'done', 'to_play')]
        if log is None:
            _check(lib().mz_act_env_step(self._ctx, C.byref(env_struct), float(temperature), _ptr(uu), *outs, _stream()))
        else:
            _check(lib().mz_act_env_step_log(self._ctx, C.byref(env_struct), C.byref(log), int(row), float(temperature),
                                             _ptr(uu), *outs, _stream()))

    def check(self):
        """Waits for the stream and raises EngineError if a kernel flagged an inconsistent state or a noise
        buffer shorter than the selection calls a search made."""
        _check(lib().mz_check(self._ctx, _stream()))

    def stats(self):
        nv = torch.empty(self.B, dtype=torch.int64, device=self.device)
        nc = torch.empty(self.B, dtype=torch.int32, device=self.device)
        _check(lib().mz_stats(self._ctx, _ptr(nv), _ptr(nc), _stream()))
        return nv, nc

    # ------------------------------------------------------------------ host views
    def _host_arrays(self):
        N, A = self.N, self.A
        pd = np.float32 if self.gumbel else np.float64
        return dict(node_visits=np.zeros(N, np.int32), node_values=np.zeros(N), node_raw_values=np.zeros(N),
                    parents=np.zeros(N, np.int32), action_from_parent=np.zeros(N, np.int32),
                    to_play=np.zeros(N, np.int32), children_index=np.zeros((N, A), np.int32),
                    children_prior_logits=np.zeros((N, A), pd), children_visits=np.zeros((N, A), np.int32),
                    children_rewards=np.zeros((N, A)), children_values=np.zeros((N, A)),
                    hidden_states=np.zeros((N, max(self.H, 1)), np.float32))

    @staticmethod
    def _host_struct(arrs, with_hidden=True):
        th = MzTreeHost()
        for k, v in arrs.items():
            setattr(th, k, v.ctypes.data_as(C.c_void_p))
        if not with_hidden:
            th.hidden_states = None
        return th

    def export_tree(self, b: int) -> dict:
        """Tree `b` as host arrays named like the fields of the reference's Tree."""
        arrs = self._host_arrays()
        th = self._host_struct(arrs, self.H > 0)
        _check(lib().mz_tree_export(self._ctx, int(b), C.byref(th), _stream()))
        return arrs

    def import_tree(self, b: int, arrays: dict, root_legal, gumbel_noise=None):
        arrs = self._host_arrays()
        for k in arrs:
            if k == 'hidden_states':
                if self.H > 0 and arrays.get(k) is not None:
                    arrs[k][...] = np.asarray(arrays[k], np.float32).reshape(self.N, self.H)
                continue
            arrs[k][...] = np.asarray(arrays[k]).reshape(arrs[k].shape)
        th = self._host_struct(arrs, self.H > 0 and arrays.get('hidden_states') is not None)
        rl = np.ascontiguousarray(root_legal, np.float64)
        gn = None if gumbel_noise is None else np.ascontiguousarray(gumbel_noise, np.float64)
        _check(lib().mz_tree_import(self._ctx, int(b), C.byref(th), rl.ctypes.data_as(C.c_void_p),
                                    None if gn is None else gn.ctypes.data_as(C.c_void_p), _stream()))
        self.B = max(self.B, b + 1)

    def node_qtransform(self, b, node, which):
        out = np.zeros(self.A)
        _check(lib().mz_node_qtransform(self._ctx, int(b), int(node), int(which), out.ctypes.data_as(C.c_void_p),
                                        _stream()))
        return out

    def node_backward(self, b, leaf, to_play):
        _check(lib().mz_node_backward(self._ctx, int(b), int(leaf), int(to_play), _stream()))

    def node_select(self, b, node, depth, uniform=None):
        act = C.c_int32(-1)
        u = None if uniform is None else np.ascontiguousarray(uniform, np.float64)
        _check(lib().mz_node_select(self._ctx, int(b), int(node), int(depth),
                                    None if u is None else u.ctypes.data_as(C.c_void_p), C.byref(act), _stream()))
        return int(act.value)
