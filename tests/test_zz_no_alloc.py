"""GPU: mz_expand_backup_net_static (staging in the library's fixed device array, no allocation) against
mz_expand_backup_net, through the C ABI: two searches in a row on the same GPU, the second with a smaller batch,
every tree and every output bit for bit; a batch beyond the array's size is rejected."""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _net_and_cfg(kind):
    from muzero_amd.network import MuZeroNetwork
    if kind == 'resnet':
        ncfg = SimpleNamespace(network='resnet', observation_dim=[3, 3, 3], n_actions=9, n_stacked_observations=1,
                               blocks=1, channels=16, reduced_channels_reward=8, reduced_channels_policy=8,
                               reduced_channels_value=8, resnet_fc_reward_layers=[16], resnet_fc_policy_layers=[16],
                               resnet_fc_value_layers=[16], downsample=False, support_limit=1)
        cfg = SimpleNamespace(simulations=12, n_actions=9, gamma=1, players=2, gumbel=True, max_considered_actions=16,
                              gumbel_scale=1.0, c_visit=50, c_scale=1.0)
        obs_size = 27
    else:
        ncfg = SimpleNamespace(network='mlp', observation_dim=[1, 1, 16], n_actions=5, n_stacked_observations=1,
                               encoding_size=64, fc_reward_layers=[64], fc_policy_layers=[64], fc_value_layers=[64],
                               fc_representation_layers=[32], fc_hidden_state_layers=[64], support_limit=40)
        cfg = SimpleNamespace(simulations=10, n_actions=5, gamma=0.997, players=1, gumbel=False, c_base=19625,
                              c_init=1.25, dirichlet_alpha=0.3, exploration_frac=0.25)
        obs_size = 16
    return MuZeroNetwork(ncfg, seed=3).to('cuda'), cfg, obs_size


def _search(eng, net, cfg, obs, rng_seed, static):
    rng = np.random.default_rng(rng_seed)
    B, A = obs.shape[0], cfg.n_actions
    legal = np.ones((B, A))
    tp = rng.integers(0, cfg.players, size=B).astype(np.int32)
    g = rng.gumbel(size=(B, A)) if cfg.gumbel else None
    di = None if cfg.gumbel else rng.dirichlet([0.3] * A, size=B)
    lg, hid, val = net.initial_flat(obs)
    prior = eng.root_prepare(lg, legal, di, 0.0 if cfg.gumbel else 0.25)
    eng.reset(prior, val, hid, tp, legal, g, tree_id0=11)
    eng.select(want_hidden=False)
    for s in range(cfg.simulations):
        # alternate the fused backup+select form and the plain form followed by a separate selection
        nxt = s % 2 == 0
        eng.expand_backup_net(select_next=nxt, static_scratch=static)
        if not nxt and s + 1 < cfg.simulations:
            eng.select(want_hidden=False)
    action, value, probs, q = eng.to_act(1.0, rng.random(B))
    trees = [eng.export_tree(b) for b in (0, B // 2, B - 1)]
    return action.cpu().numpy(), value.cpu().numpy(), probs.cpu().numpy(), trees


@pytest.mark.parametrize('kind', ['resnet', 'mlp'])
def test_static_scratch_step_equals_allocating_step(kind):
    from muzero_amd.engine import SearchEngine
    from muzero_amd.packing import pack_network_mfma
    net, cfg, obs_size = _net_and_cfg(kind)
    pm_io = pack_network_mfma(net)
    rng = np.random.default_rng(0)
    engines = []
    for _ in range(2):
        e = SearchEngine(cfg, net.H, 300, seed=5)
        e.set_network_mfma(*pm_io)
        engines.append(e)
    ref_eng, st_eng = engines
    for B in (300, 77):          # the second call is smaller: rows 96..299 of the array keep the first call's data
        obs = torch.as_tensor(rng.standard_normal((B, obs_size)).astype(np.float32), device='cuda')
        want = _search(ref_eng, net, cfg, obs, 100 + B, static=False)
        got = _search(st_eng, net, cfg, obs, 100 + B, static=True)
        for w, g_ in zip(want[:3], got[:3]):
            assert np.array_equal(w, g_)
        for tw, tg in zip(want[3], got[3]):
            for k in tw:
                assert np.array_equal(tw[k], tg[k]), (B, k)
    for e in engines:
        e.close()


def test_static_scratch_rejects_what_does_not_fit():
    from muzero_amd.engine import SearchEngine
    from muzero_amd.packing import pack_network_mfma
    net, cfg, obs_size = _net_and_cfg('resnet')
    pm, io = pack_network_mfma(net)
    B = (16 << 20) // (pm.stride * 4) + 64       # activations alone exceed the 16 MiB array
    eng = SearchEngine(cfg, net.H, B, seed=0)
    eng.set_network_mfma(pm, io)
    obs = torch.zeros((B, obs_size), device='cuda')
    lg, hid, val = net.initial_flat(obs)
    eng.reset(eng.root_prepare(lg, np.ones((B, 9))), val, hid, np.zeros(B, np.int32), np.ones((B, 9)),
              np.zeros((B, 9)))
    eng.select(want_hidden=False)
    with pytest.raises(ValueError, match='fixed scratch'):
        eng.expand_backup_net(static_scratch=True)
    eng.expand_backup_net()      # the pending selection is still there for the allocating entry point
    eng.close()
