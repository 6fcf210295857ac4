"""A small differentiable network with the call interface the reference's trainer uses
(`initial_inference(obs, False)`, `recurrent_inference(hidden, action, False)` returning logits).
Test scaffolding only: it lets the trainer logic be pinned without the reference's network classes,
which cannot travel to the GPU box.  Used by tests/golden/make_fixtures.py and the trainer test."""
import torch
import torch.nn as nn


class TinyNet(nn.Module):
    def __init__(self, obs_size=4, n_actions=2, hidden=8, support_limit=10, seed=0):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        bins = 2 * support_limit + 1
        self.support_limit = support_limit
        self.rep = nn.Linear(obs_size, hidden)
        self.dyn = nn.Linear(hidden + n_actions, hidden)
        self.rew = nn.Linear(hidden + n_actions, bins)
        self.pol = nn.Linear(hidden, n_actions)
        self.val = nn.Linear(hidden, bins)
        with torch.no_grad():
            for p in self.parameters():
                p.copy_(torch.randn(p.shape, generator=g) * 0.3)

    def initial_inference(self, observation, inv_transform=True):
        h = torch.tanh(self.rep(observation.reshape(observation.shape[0], -1)))
        return self.pol(h), h, self.val(h)

    def recurrent_inference(self, hidden_state, encoded_action, scalar_transform=True):
        x = torch.cat((hidden_state, encoded_action.reshape(encoded_action.shape[0], -1)), dim=1)
        return self.pol(hidden_state), torch.tanh(self.dyn(x)), self.val(hidden_state), self.rew(x)
