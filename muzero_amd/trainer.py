"""Training step on the device: the unrolled MuZero loss and optimiser update.

Mirrors the reference's `Trainer` (trainer.py:19-33 construction, :43-45 learning-rate schedule,
:62-123 `update_weights` / `loss`) for any differentiable network with the reference's call
interface (`initial_inference(obs, False)`, `recurrent_inference(hidden, action, False)`), without
the actor plumbing.  Stock PyTorch-ROCm carries the forward / backward passes; the categorical
value and reward targets come from the engine's `mz_scalar_to_support` kernel, and the batches from
`muzero_amd.targets.TrajectoryStore.sample`.

Behaviour kept from the reference: the recurrent step k is fed `action_batch[:, k]` (the first
unrolled action slot is never used, trainer.py:77-79); the gradient entering each new hidden state is
halved; every unrolled step's losses are scaled by 1 / (unroll_steps + 1) in the backward pass only
(the reported sums are unscaled); the reward loss of step 0 is dropped; loss =
(value_loss_weight * value + reward + policy) / 3.
"""
from __future__ import annotations

from typing import Tuple

import torch
import torch.nn as nn

from .targets import scalar_to_support


def _scale_gradient(x: torch.Tensor, factor: float) -> None:
    x.register_hook(lambda grad: grad * factor)


class Trainer:
    def __init__(self, network: nn.Module, config, training_step: int = 0, optimizer_state=None) -> None:
        self.config = config
        self.network = network
        self.network.train()
        if config.optimizer == 'Adam':
            self.optimizer = torch.optim.Adam(network.parameters(), lr=config.lr, weight_decay=config.weight_decay)
        else:
            self.optimizer = torch.optim.SGD(network.parameters(), lr=config.lr, momentum=config.momentum,
                                             weight_decay=config.weight_decay)
        if optimizer_state is not None:
            self.optimizer.load_state_dict(optimizer_state)
        self.training_step = int(training_step)

    def update_lr(self) -> float:
        """Exponential schedule of utils/network_utils.py:166-174, evaluated at the current step."""
        cfg = self.config
        lr = cfg.lr * cfg.lr_decay_rate ** (self.training_step / cfg.lr_decay_steps)
        for group in self.optimizer.param_groups:
            group['lr'] = lr
        return lr

    def step(self, batch) -> Tuple[float, float, float, float]:
        """One iteration of update_weights_continuously: schedule the learning rate, then update."""
        self.update_lr()
        return self.update_weights(batch)

    def update_weights(self, batch) -> Tuple[float, float, float, float]:
        cfg = self.config
        device = next(self.network.parameters()).device
        obs, actions, value_target, reward_target, policy_target = (x.to(device) for x in batch)
        value_target = scalar_to_support(value_target, cfg.support_limit)
        reward_target = scalar_to_support(reward_target, cfg.support_limit)

        policy_logits, hidden, value_logits = self.network.initial_inference(obs, False)
        predictions = [(1.0, value_logits, torch.zeros_like(value_logits, requires_grad=True), policy_logits)]
        steps = actions.shape[1]
        for k in range(1, steps):
            policy_logits, hidden, value_logits, reward_logits = self.network.recurrent_inference(
                hidden, actions[:, k], False)
            _scale_gradient(hidden, 0.5)
            predictions.append((1.0 / steps, value_logits, reward_logits, policy_logits))

        ce = nn.CrossEntropyLoss()
        value_loss = reward_loss = policy_loss = 0
        for k, (scale, v_logits, r_logits, p_logits) in enumerate(predictions):
            v_k = ce(v_logits, value_target[:, k])
            r_k = ce(r_logits, reward_target[:, k])
            p_k = ce(p_logits, policy_target[:, k])
            for t in (v_k, r_k, p_k):
                _scale_gradient(t, scale)
            if k == 0:
                r_k = torch.zeros_like(v_k)
            value_loss = value_loss + v_k
            reward_loss = reward_loss + r_k
            policy_loss = policy_loss + p_k

        loss = (value_loss * cfg.value_loss_weight + reward_loss + policy_loss) / 3
        self.optimizer.zero_grad()
        loss.backward()
        self.optimizer.step()
        self.training_step += 1
        return loss.item(), value_loss.item(), reward_loss.item(), policy_loss.item()
