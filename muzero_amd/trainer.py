"""Training step on the device: the unrolled MuZero loss and optimiser update.

Produces the numbers of the reference's `Trainer` (trainer.py:19-33 construction, :43-45 learning-rate
schedule, :62-123 `update_weights` / `loss`) for any differentiable network with the reference's call
interface (`initial_inference(obs, False)`, `recurrent_inference(hidden, action, False)`), without the
actor plumbing.  Stock PyTorch-ROCm carries the forward / backward passes; the categorical value and
reward targets come from the engine's `mz_scalar_to_support` kernel, and the batches from
`muzero_amd.targets.TrajectoryStore.sample`.

Structure: the network is unrolled once into three stacked logit tensors `[K+1, B, classes]` (value,
reward, policy); each head's cross-entropy against its soft targets is then ONE expression over the
whole stack, giving a `[3, K+1]` table of per-step losses.  The two places where the reference rescales
gradients without touching the forward values are one autograd function (`ScaleGradient`): a per-step
weight vector on the loss table, and 1/2 on every hidden state handed to the next recurrent step.

Behaviour kept from the reference: recurrent step k consumes `action_batch[:, k]` (slot 0 is never
used, trainer.py:77-79); the step-0 reward term is excluded (trainer.py:98-99); the reported sums are
unweighted; loss = (value_loss_weight * value + reward + policy) / 3 (trainer.py:104).
"""
from __future__ import annotations

from typing import Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from .targets import scalar_to_support


class ScaleGradient(torch.autograd.Function):
    """Identity in the forward pass; multiplies the incoming gradient by `factor` (a float or a tensor
    that broadcasts against it) in the backward pass."""

    @staticmethod
    def forward(ctx, x, factor):
        ctx.factor = factor
        return x.view_as(x)

    @staticmethod
    def backward(ctx, grad):
        return grad * ctx.factor, None


def soft_cross_entropy(logits: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """Cross-entropy against probability targets, averaged over the batch axis: `[..., B, C]` -> `[...]`
    (what `nn.CrossEntropyLoss()` computes for one `[B, C]` slice, trainer.py:119-122)."""
    return -(target * F.log_softmax(logits, dim=-1)).sum(dim=-1).mean(dim=-1)


def build_optimizer(network: nn.Module, config) -> torch.optim.Optimizer:
    """trainer.py:22-33: Adam, or SGD with momentum, both with weight decay."""
    common = dict(lr=config.lr, weight_decay=config.weight_decay)
    if config.optimizer == 'Adam':
        return torch.optim.Adam(network.parameters(), **common)
    return torch.optim.SGD(network.parameters(), momentum=config.momentum, **common)


class Trainer:
    HIDDEN_GRADIENT_SCALE = 0.5

    def __init__(self, network: nn.Module, config, training_step: int = 0, optimizer_state=None) -> None:
        self.config = config
        self.network = network.train()
        self.optimizer = build_optimizer(network, config)
        if optimizer_state is not None:
            self.optimizer.load_state_dict(optimizer_state)
        self.training_step = int(training_step)

    # ------------------------------------------------------------------ schedule
    def update_lr(self) -> float:
        """Exponential decay lr * rate ** (step / decay_steps) (utils/network_utils.py:166-174)."""
        cfg = self.config
        lr = cfg.lr * cfg.lr_decay_rate ** (self.training_step / cfg.lr_decay_steps)
        for group in self.optimizer.param_groups:
            group['lr'] = lr
        return lr

    # ------------------------------------------------------------------ forward
    def unroll(self, obs: torch.Tensor, actions: torch.Tensor):
        """Runs the network along the stored actions.  Returns (value, reward, policy) logits stacked over
        the K+1 positions: `[K+1, B, 2L+1]`, `[K+1, B, 2L+1]`, `[K+1, B, A]`.  The reward row of position 0
        (there is no transition into it) is zeros."""
        policy, hidden, value = self.network.initial_inference(obs, False)
        rows = {'value': [value], 'reward': [torch.zeros_like(value)], 'policy': [policy]}
        for k in range(1, actions.shape[1]):
            policy, hidden, value, reward = self.network.recurrent_inference(hidden, actions[:, k], False)
            hidden = ScaleGradient.apply(hidden, self.HIDDEN_GRADIENT_SCALE)
            for name, x in (('value', value), ('reward', reward), ('policy', policy)):
                rows[name].append(x)
        return tuple(torch.stack(rows[name]) for name in ('value', 'reward', 'policy'))

    def loss_table(self, batch) -> torch.Tensor:
        """`[3, K+1]` per-position losses (rows: value, reward, policy), gradient-weighted per position."""
        device = next(self.network.parameters()).device
        obs, actions, value_target, reward_target, policy_target = (x.to(device) for x in batch)
        L = self.config.support_limit
        targets = (scalar_to_support(value_target, L), scalar_to_support(reward_target, L), policy_target)
        logits = self.unroll(obs, actions)
        table = torch.stack([soft_cross_entropy(x, t.transpose(0, 1)) for x, t in zip(logits, targets)])
        n_pos = table.shape[1]
        keep = torch.ones_like(table)
        keep[1, 0] = 0.0                                   # no reward term at the root position
        weight = torch.full((n_pos,), 1.0 / n_pos, dtype=table.dtype, device=table.device)
        weight[0] = 1.0                                    # gradient weights: root 1, unrolled positions 1/(K+1)
        return ScaleGradient.apply(table * keep, weight)

    # ------------------------------------------------------------------ update
    def update_weights(self, batch) -> Tuple[float, float, float, float]:
        value_loss, reward_loss, policy_loss = self.loss_table(batch).sum(dim=1)
        loss = (self.config.value_loss_weight * value_loss + reward_loss + policy_loss) / 3
        self.optimizer.zero_grad()
        loss.backward()
        self.optimizer.step()
        self.training_step += 1
        return loss.item(), value_loss.item(), reward_loss.item(), policy_loss.item()

    def step(self, batch) -> Tuple[float, float, float, float]:
        """One iteration of the training loop: schedule the learning rate, then update (trainer.py:41-45)."""
        self.update_lr()
        return self.update_weights(batch)
