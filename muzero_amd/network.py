"""MuZero networks lowered to dense-layer programs for batched leaf evaluation.

The reference builds its MLP and ResNet networks as nn.Module trees and calls them one
position at a time (`network.py:86-126`).  Here a checkpoint's ``state_dict`` (same keys as
the reference's `MuZeroFullyConnectedNetwork` / `MuZeroResidualNetwork`, `network.py:133-225`)
is *compiled* once into a short program of dense layers:

* every maximal affine stretch (3x3 / 1x1 convolutions on the fixed small board, eval-mode
  BatchNorm, Flatten, Linear) is folded into one matrix + bias by probing it with basis
  vectors, so a board convolution becomes a plain GEMM over the flattened ``C*h*w`` state;
* what remains between them are the pointwise pieces: ReLU / ELU, residual adds, the
  min-max normalisation of hidden states (`utils/network_utils.py:132-158`) and the
  categorical-support -> scalar transform (`utils/network_utils.py:75-96`).

The same program is executed either by torch matmuls (any device; used on the host for the
single-position drop-in path and as the fp32 reference in tests) or by the HIP kernels of the
engine, which take the packed weights.  Hidden states are handled flattened (``B x H``) and
reshaped to the reference's shape only at the API boundary.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-5       # torch.nn.BatchNorm2d default, used by the reference's ConvBlock/ResidualBlock
NORM_EPS = 1e-5     # utils/network_utils.py:144,157
SCALAR_EPS = 0.001  # utils/network_utils.py:75


@dataclass
class DenseStep:
    """out = act(inp @ W.T + b (+ skip))"""
    out: str
    inp: str
    W: torch.Tensor            # [N, K] float32
    b: torch.Tensor            # [N] float32
    act: str = 'none'          # 'none' | 'relu' | 'elu'
    skip: Optional[str] = None


@dataclass
class DenseProgram:
    steps: List[DenseStep] = field(default_factory=list)

    def add(self, out, inp, W, b, act='none', skip=None):
        self.steps.append(DenseStep(out, inp, W.contiguous().float(), b.contiguous().float(), act, skip))

    def run(self, env: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        for s in self.steps:
            y = F.linear(env[s.inp], s.W, s.b)
            if s.skip is not None:
                y = y + env[s.skip]
            if s.act == 'relu':
                y = torch.relu(y)
            elif s.act == 'elu':
                y = F.elu(y)
            env[s.out] = y
        return env

    def to(self, device):
        for s in self.steps:
            s.W, s.b = s.W.to(device), s.b.to(device)
        return self

    def macs(self) -> int:
        return sum(int(s.W.shape[0] * s.W.shape[1]) for s in self.steps)


def affine_of(fn, in_shape: Sequence[int]):
    """Matrix and bias of an affine map `fn` over inputs of shape `in_shape` (probed in float64)."""
    K = int(np.prod(in_shape))
    with torch.no_grad():
        zero = torch.zeros((1, *in_shape), dtype=torch.float64)
        b = fn(zero).reshape(-1)
        basis = torch.eye(K, dtype=torch.float64).reshape(K, *in_shape)
        W = (fn(basis).reshape(K, -1) - b).T  # [N, K]
    return W.contiguous(), b.contiguous()


def support_to_scalar(logits: torch.Tensor, support_limit: int) -> torch.Tensor:
    """utils/network_utils.py:80-96: softmax (f32) . arange(-L..L) in float64, then the inverse
    of the value-scaling transform h(x) = sign(x)(sqrt(|x|+1)-1) + eps*x."""
    p = torch.softmax(logits.float(), dim=1).double()
    support = torch.arange(-support_limit, support_limit + 1, dtype=torch.float64, device=logits.device)
    x = p @ support
    e = SCALAR_EPS
    return torch.sign(x) * (((torch.sqrt(1 + 4 * e * (torch.abs(x) + 1 + e)) - 1) / (2 * e)) ** 2 - 1)


def minmax_normalize(h: torch.Tensor, groups: int) -> torch.Tensor:
    """Per-sample min-max scaling over each of `groups` equal slices of the flattened state
    (1 group for MLP states, one per channel for board states)."""
    B = h.shape[0]
    g = h.reshape(B, groups, -1)
    mx = g.max(dim=2, keepdim=True)[0]
    mn = g.min(dim=2, keepdim=True)[0]
    return ((g - mn) / (mx - mn + NORM_EPS)).reshape(B, -1)


def _bn(x, sd, p):
    return F.batch_norm(x, sd[p + '.running_mean'].double(), sd[p + '.running_var'].double(),
                        sd[p + '.weight'].double(), sd[p + '.bias'].double(), False, 0.0, BN_EPS)


def _conv3(x, sd, key, stride=1):
    return F.conv2d(x, sd[key].double(), None, stride=stride, padding=1)


class TorchRepresentation:
    """network.py:228-325 with `downsample` on: the DownSample stack (two stride-2 convolutions, eight residual blocks,
    two average poolings: h x w -> ceil(h/16) x ceil(w/16)) and the residual tower, evaluated by torch's own
    convolutions in float32 on the network's device.  Observations at full resolution are not lowered to dense
    programs - a dense matrix over a stack of frames has 10^8..10^9 entries - and the representation runs once per
    search, at the root; the dynamics and prediction programs, which every simulation evaluates, work on the
    downsampled state and are lowered like any board network.  Same `run` interface as DenseProgram; `steps` is None."""
    steps = None

    def __init__(self, sd, cin, h, w, channels, blocks):
        self.cin, self.h, self.w = cin, h, w
        p = 'representation_network.downsample_net'
        self.names = ([p + '.conv1'] + [f'{p}.res_block1.{i}' for i in range(2)] + [p + '.conv2'] +
                      [f'{p}.res_block2.{i}' for i in range(3)] + ['pool'] + [f'{p}.res_block3.{i}' for i in range(3)] +
                      ['pool'] + [f'representation_network.res_tower.{i}' for i in range(blocks)])
        keep = tuple(n + '.' for n in self.names if n != 'pool')
        self.t = {k: v.float() for k, v in sd.items() if k.startswith(keep) and v.is_floating_point()}

    def to(self, device):
        self.t = {k: v.to(device) for k, v in self.t.items()}
        return self

    def _bn(self, x, p):
        t = self.t
        return F.batch_norm(x, t[p + '.running_mean'], t[p + '.running_var'], t[p + '.weight'], t[p + '.bias'], False, 0.0, BN_EPS)

    def run(self, inputs):
        x = inputs['obs'].float().reshape(-1, self.cin, self.h, self.w)
        for n in self.names:
            if n == 'pool':
                x = F.avg_pool2d(x, kernel_size=3, stride=2, padding=1)
            elif n.endswith(('.conv1', '.conv2')) and '.res_' not in n:
                x = F.conv2d(x, self.t[n + '.weight'], None, stride=2, padding=1)
            else:   # utils/network_utils.py:47-68
                y = F.relu(self._bn(F.conv2d(x, self.t[n + '.conv1.weight'], None, padding=1), n + '.bn1'))
                x = F.relu(self._bn(F.conv2d(y, self.t[n + '.conv2.weight'], None, padding=1), n + '.bn2') + x)
        return {'hidden_raw': x.flatten(1)}

    def macs(self) -> int:
        return 0   # not part of a search's simulations


class MuZeroDenseNetwork:
    """A MuZero network compiled to dense programs; same call interface as the reference's
    networks (`initial_inference`, `recurrent_inference`, `set_weights`)."""

    def __init__(self, config, state_dict: Optional[Dict[str, torch.Tensor]] = None, seed: Optional[int] = None):
        self.config = config
        self.kind = config.network
        self.A = int(config.n_actions)
        self.support_limit = int(config.support_limit)
        self.device = torch.device('cpu')
        if state_dict is None:
            state_dict = random_state_dict(config, seed)
        self.set_weights(state_dict)

    # ------------------------------------------------------------------ compilation
    def set_weights(self, weights) -> None:
        if weights is None:
            return
        sd = {k: torch.as_tensor(v).detach().cpu() for k, v in weights.items()}
        self.state_dict_cpu = sd
        if self.kind == 'resnet':
            self._compile_resnet(sd)
        else:
            self._compile_mlp(sd)
        self.to(self.device)

    def load_state_dict(self, weights):
        self.set_weights(weights)

    def state_dict(self):
        return self.state_dict_cpu

    def eval(self):
        return self

    @staticmethod
    def _mlp_steps(prog, sd, prefix, inp, out, pad_to=None, final_act='none'):
        """utils/network_utils.py:8-22: Linear + ELU stacks, identity on the last layer."""
        idx = sorted({int(k[len(prefix) + 1:].split('.')[0]) for k in sd if k.startswith(prefix + '.') and k.endswith('.weight')})
        cur = inp
        for j, i in enumerate(idx):
            W, b = sd[f'{prefix}.{i}.weight'], sd[f'{prefix}.{i}.bias']
            if j == 0 and pad_to is not None and W.shape[1] < pad_to:
                W = torch.cat([W, torch.zeros(W.shape[0], pad_to - W.shape[1])], dim=1)
            last = j == len(idx) - 1
            name = out if last else f'{out}_h{j}'
            prog.add(name, cur, W, b, final_act if last else 'elu')
            cur = name

    def _compile_mlp(self, sd):
        """network.py:189-225, 405-479."""
        c = self.config
        self.obs_size = int(c.n_stacked_observations * math.prod(c.observation_dim))
        self.H = int(c.encoding_size)
        self.hidden_shape = (self.H,)
        self.norm_groups = 1
        self.action_size = self.A          # one-hot action, games/cartpole.py:28-31
        rep, rec = DenseProgram(), DenseProgram()
        self._mlp_steps(rep, sd, 'representation_network.network', 'obs', 'hidden_raw')
        self._mlp_steps(rec, sd, 'dynamics_network.hidden_state_network', 'xa', 'next_raw')
        self._mlp_steps(rec, sd, 'dynamics_network.reward_network', 'xa', 'reward_logits')
        pred = DenseProgram()
        self._mlp_steps(pred, sd, 'prediction_network.policy_network', 'hidden', 'policy_logits')
        self._mlp_steps(pred, sd, 'prediction_network.value_network', 'hidden', 'value_logits')
        self.rep, self.dyn, self.pred = rep, rec, pred

    def _res_block(self, prog, sd, p, inp, out, shape):
        """utils/network_utils.py:47-68: relu(bn2(conv2(relu(bn1(conv1(x))))) + x)."""
        W1, b1 = affine_of(lambda x: _bn(_conv3(x, sd, p + '.conv1.weight'), sd, p + '.bn1'), shape)
        W2, b2 = affine_of(lambda x: _bn(_conv3(x, sd, p + '.conv2.weight'), sd, p + '.bn2'), shape)
        prog.add(out + '_mid', inp, W1, b1, 'relu')
        prog.add(out, out + '_mid', W2, b2, 'relu', skip=inp)

    def _compile_resnet(self, sd):
        """network.py:133-186, 228-402: the board (or the downsampled observation) is small, so each conv stretch of
        the dynamics and prediction networks is one dense matrix; so is the representation of a board game
        (downsample off, the only form the reference's games use, main.py:47-48)."""
        c = self.config
        C, (h, w) = int(c.channels), c.observation_dim[1:]
        cin = int(c.n_stacked_observations * c.observation_dim[0])
        self.obs_size = cin * h * w
        down = bool(getattr(c, 'downsample', False))
        if down:   # network.py:152-156, 298-304: dynamics and prediction work on the downsampled state
            rep_torch = TorchRepresentation(sd, cin, h, w, C, int(c.blocks))
            h, w = math.ceil(h / 16), math.ceil(w / 16)
        self.H = C * h * w
        if self.H + h * w > 4096:
            raise NotImplementedError('board too large for dense lowering')
        self.hidden_shape = (C, h, w)
        self.norm_groups = C
        self.action_size = h * w           # board plane, games/tictactoe.py:66-69
        if self.A > self.action_size:
            raise ValueError(f'{self.A} actions do not fit a {h}x{w} action plane (one cell per action)')
        blocks = int(c.blocks)
        rep, dyn, pred = DenseProgram(), DenseProgram(), DenseProgram()
        # representation: ConvBlock + residual tower (downsampled observations: evaluated by torch, see TorchRepresentation)
        if down:
            rep = rep_torch
        else:
            p = 'representation_network.conv_block.network'
            W, b = affine_of(lambda x: _bn(_conv3(x, sd, p + '.0.weight'), sd, p + '.1'), (cin, h, w))
            rep.add('r0' if blocks else 'hidden_raw', 'obs', W, b, 'relu')
            for i in range(blocks):
                self._res_block(rep, sd, f'representation_network.res_tower.{i}', f'r{i}',
                                'hidden_raw' if i == blocks - 1 else f'r{i + 1}', (C, h, w))
        # dynamics: ConvBlock over [hidden, action plane] + tower, reward head conv3x3 -> flatten -> mlp
        p = 'dynamics_network.conv_block.network'
        W, b = affine_of(lambda x: _bn(_conv3(x, sd, p + '.0.weight'), sd, p + '.1'), (C + 1, h, w))
        dyn.add('d0' if blocks else 'next_raw', 'xa', W, b, 'relu')
        for i in range(blocks):
            self._res_block(dyn, sd, f'dynamics_network.res_tower.{i}', f'd{i}',
                            'next_raw' if i == blocks - 1 else f'd{i + 1}', (C, h, w))
        self._head(dyn, sd, 'dynamics_network.reward_head', 'next_raw', 'reward_logits', (C, h, w), conv3=True)
        # prediction: tower + two 1x1-conv heads
        cur = 'hidden'
        for i in range(blocks):
            self._res_block(pred, sd, f'prediction_network.res_tower.{i}', cur, f'p{i + 1}', (C, h, w))
            cur = f'p{i + 1}'
        self._head(pred, sd, 'prediction_network.policy_head', cur, 'policy_logits', (C, h, w))
        self._head(pred, sd, 'prediction_network.value_head', cur, 'value_logits', (C, h, w))
        self.rep, self.dyn, self.pred = rep, dyn, pred

    @staticmethod
    def _head(prog, sd, p, inp, out, shape, conv3=False):
        """conv (3x3 no bias for the reward head, 1x1 with bias otherwise) -> Flatten -> mlp; the conv,
        the flatten and the mlp's first Linear have no nonlinearity between them and fold into one matrix."""
        idx = sorted({int(k.split('.')[-2]) for k in sd if k.startswith(p + '.2.') and k.endswith('.weight')})
        first = idx[0]

        def stretch(x):
            if conv3:
                y = F.conv2d(x, sd[p + '.0.weight'].double(), None, padding=1)
            else:
                y = F.conv2d(x, sd[p + '.0.weight'].double(), sd[p + '.0.bias'].double())
            return F.linear(y.flatten(1), sd[f'{p}.2.{first}.weight'].double(), sd[f'{p}.2.{first}.bias'].double())
        W, b = affine_of(stretch, shape)
        last = len(idx) == 1
        prog.add(out if last else out + '_h0', inp, W, b, 'none' if last else 'elu')
        cur = out + '_h0'
        for j, i in enumerate(idx[1:], start=1):
            last = j == len(idx) - 1
            name = out if last else f'{out}_h{j}'
            prog.add(name, cur, sd[f'{p}.2.{i}.weight'], sd[f'{p}.2.{i}.bias'], 'none' if last else 'elu')
            cur = name

    def to(self, device):
        self.device = torch.device(device)
        for prog in (self.rep, self.dyn, self.pred):
            prog.to(self.device)
        return self

    # ------------------------------------------------------------------ execution (torch)
    def encode_actions(self, actions: torch.Tensor) -> torch.Tensor:
        """Action indices -> the flattened encoding the dynamics input uses (one-hot vector or one-hot board plane)."""
        return F.one_hot(actions.long(), self.action_size).float()

    def _predict(self, hidden):
        env = self.pred.run({'hidden': hidden})
        return env['policy_logits'], support_to_scalar(env['value_logits'], self.support_limit)

    def initial_flat(self, obs_flat: torch.Tensor):
        with torch.no_grad():
            raw = self.rep.run({'obs': obs_flat.float()})['hidden_raw']
            hidden = minmax_normalize(raw, self.norm_groups)
            logits, value = self._predict(hidden)
        return logits, hidden, value

    def recurrent_flat(self, hidden: torch.Tensor, enc_action_flat: torch.Tensor):
        """hidden f32[B,H] (the PARENT's state), encoded action f32[B,action_size] ->
        (policy logits of the parent state, next hidden, value of the parent state, reward),
        exactly the pairing of network.py:120-125."""
        with torch.no_grad():
            xa = torch.cat([hidden, enc_action_flat], dim=1)
            env = self.dyn.run({'xa': xa})
            next_hidden = minmax_normalize(env['next_raw'], self.norm_groups)
            reward = support_to_scalar(env['reward_logits'], self.support_limit)
            logits, value = self._predict(hidden)
        return logits, next_hidden, value, reward

    # -- the reference's call interface (network.py:86-126) ---------------
    def initial_inference(self, observation: torch.Tensor):
        B = observation.shape[0]
        logits, hidden, value = self.initial_flat(observation.reshape(B, -1).to(self.device))
        return logits, hidden.reshape(B, *self.hidden_shape), value

    def recurrent_inference(self, hidden_state: torch.Tensor, encoded_action: torch.Tensor):
        B = hidden_state.shape[0]
        logits, nh, value, reward = self.recurrent_flat(hidden_state.reshape(B, -1).to(self.device).float(),
                                                        encoded_action.reshape(B, -1).to(self.device).float())
        return logits, nh.reshape(B, *self.hidden_shape), value, reward

    def recurrent_macs(self) -> int:
        return self.dyn.macs() + self.pred.macs()


def MuZeroNetwork(config, state_dict=None, seed=None) -> MuZeroDenseNetwork:
    """Factory with the reference's name (`network.py:11-42`)."""
    return MuZeroDenseNetwork(config, state_dict, seed)


# ---------------------------------------------------------------------- random initialisation

def _linear_init(g, out_f, in_f):
    bound = 1.0 / math.sqrt(in_f)
    W = (torch.rand(out_f, in_f, generator=g) * 2 - 1) * bound
    b = (torch.rand(out_f, generator=g) * 2 - 1) * bound
    return W, b


def _conv_init(g, co, ci, k, bias):
    fan_in = ci * k * k
    bound = 1.0 / math.sqrt(fan_in)
    W = (torch.rand(co, ci, k, k, generator=g) * 2 - 1) * bound
    b = (torch.rand(co, generator=g) * 2 - 1) * bound if bias else None
    return W, b


def random_state_dict(config, seed=None) -> Dict[str, torch.Tensor]:
    """A randomly initialised checkpoint with the reference's parameter names and shapes
    (uniform +-1/sqrt(fan_in) like torch's default Linear/Conv2d init; BatchNorm at identity)."""
    g = torch.Generator().manual_seed(0 if seed is None else int(seed))
    sd: Dict[str, torch.Tensor] = {}
    c = config
    A, sup = int(c.n_actions), 2 * int(c.support_limit) + 1

    def mlp(prefix, sizes):
        for i in range(len(sizes) - 1):
            W, b = _linear_init(g, sizes[i + 1], sizes[i])
            sd[f'{prefix}.{2 * i}.weight'], sd[f'{prefix}.{2 * i}.bias'] = W, b

    def bn(prefix, ch):
        sd[prefix + '.weight'], sd[prefix + '.bias'] = torch.ones(ch), torch.zeros(ch)
        sd[prefix + '.running_mean'], sd[prefix + '.running_var'] = torch.zeros(ch), torch.ones(ch)
        sd[prefix + '.num_batches_tracked'] = torch.tensor(0)

    def res(prefix, ch):
        for j in (1, 2):
            sd[f'{prefix}.conv{j}.weight'] = _conv_init(g, ch, ch, 3, False)[0]
            bn(f'{prefix}.bn{j}', ch)

    if c.network == 'resnet':
        C, (h, w) = int(c.channels), c.observation_dim[1:]
        cin = int(c.n_stacked_observations * c.observation_dim[0])
        if getattr(c, 'downsample', False):   # network.py:228-260
            h, w = math.ceil(h / 16), math.ceil(w / 16)
            p = 'representation_network.downsample_net'
            sd[p + '.conv1.weight'] = _conv_init(g, C // 2, cin, 3, False)[0]
            for i in range(2):
                res(f'{p}.res_block1.{i}', C // 2)
            sd[p + '.conv2.weight'] = _conv_init(g, C, C // 2, 3, False)[0]
            for i in range(3):
                res(f'{p}.res_block2.{i}', C)
            for i in range(3):
                res(f'{p}.res_block3.{i}', C)
        else:
            sd['representation_network.conv_block.network.0.weight'] = _conv_init(g, C, cin, 3, False)[0]
            bn('representation_network.conv_block.network.1', C)
        res_sz = h * w
        for i in range(int(c.blocks)):
            res(f'representation_network.res_tower.{i}', C)
        sd['dynamics_network.conv_block.network.0.weight'] = _conv_init(g, C, C + 1, 3, False)[0]
        bn('dynamics_network.conv_block.network.1', C)
        for i in range(int(c.blocks)):
            res(f'dynamics_network.res_tower.{i}', C)
        sd['dynamics_network.reward_head.0.weight'] = _conv_init(g, int(c.reduced_channels_reward), C, 3, False)[0]
        mlp('dynamics_network.reward_head.2', [int(c.reduced_channels_reward) * res_sz, *c.resnet_fc_reward_layers, sup])
        for i in range(int(c.blocks)):
            res(f'prediction_network.res_tower.{i}', C)
        for head, rc, layers, out in (('policy_head', c.reduced_channels_policy, c.resnet_fc_policy_layers, A),
                                      ('value_head', c.reduced_channels_value, c.resnet_fc_value_layers, sup)):
            W, b = _conv_init(g, int(rc), C, 1, True)
            sd[f'prediction_network.{head}.0.weight'], sd[f'prediction_network.{head}.0.bias'] = W, b
            mlp(f'prediction_network.{head}.2', [int(rc) * res_sz, *layers, out])
    else:
        E = int(c.encoding_size)
        obs = int(c.n_stacked_observations * math.prod(c.observation_dim))
        mlp('representation_network.network', [obs, *c.fc_representation_layers, E])
        mlp('dynamics_network.hidden_state_network', [E + A, *c.fc_hidden_state_layers, E])
        mlp('dynamics_network.reward_network', [E + A, *c.fc_reward_layers, sup])
        mlp('prediction_network.policy_network', [E, *c.fc_policy_layers, A])
        mlp('prediction_network.value_network', [E, *c.fc_value_layers, sup])
    return sd
