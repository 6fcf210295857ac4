"""The dense-program lowering of the reference's networks against outputs of the reference's own
nn.Modules (golden files net_*.npz hold the checkpoints and the reference's outputs)."""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from helpers import load
from muzero_amd.network import MuZeroNetwork, support_to_scalar, minmax_normalize, random_state_dict

TOL = 1e-5  # the north-star tolerance for fp32 network outputs and value/reward estimates
# Hidden states end in (s - min) / (max - min + 1e-5) per channel; a channel whose range is ~1e-2
# amplifies fp32 rounding of the conv sums (1e-7) a hundredfold, so they get a wider absolute bound.
HTOL = 1e-4


def net_config(name):
    if name == 'resnet_tictactoe':
        return SimpleNamespace(network='resnet', observation_dim=[3, 3, 3], n_actions=9, n_stacked_observations=1,
                               blocks=1, channels=16, reduced_channels_reward=8, reduced_channels_policy=8,
                               reduced_channels_value=8, resnet_fc_reward_layers=[16], resnet_fc_policy_layers=[16],
                               resnet_fc_value_layers=[16], downsample=False, support_limit=1)
    if name == 'mlp_cartpole':
        return SimpleNamespace(network='mlp', observation_dim=[1, 1, 4], n_actions=2, n_stacked_observations=1,
                               encoding_size=8, fc_reward_layers=[16], fc_policy_layers=[16], fc_value_layers=[16],
                               fc_representation_layers=[], fc_hidden_state_layers=[16], support_limit=10)
    if name == 'resnet_downsample':
        return SimpleNamespace(network='resnet', observation_dim=[3, 48, 48], n_actions=9, n_stacked_observations=1,
                               blocks=1, channels=16, reduced_channels_reward=4, reduced_channels_policy=4,
                               reduced_channels_value=4, resnet_fc_reward_layers=[16], resnet_fc_policy_layers=[16],
                               resnet_fc_value_layers=[16], downsample=True, support_limit=5)
    return SimpleNamespace(network='mlp', observation_dim=[1, 1, 32], n_actions=18, n_stacked_observations=1,
                           encoding_size=64, fc_reward_layers=[64], fc_policy_layers=[64], fc_value_layers=[64],
                           fc_representation_layers=[], fc_hidden_state_layers=[64], support_limit=30)


def load_net(name):
    z = load(f'net_{name}.npz')
    sd = {k[3:]: torch.as_tensor(z[k]) for k in z.files if k.startswith('sd_')}
    return z, MuZeroNetwork(net_config(name), sd)


@pytest.mark.parametrize('name', ['mlp_cartpole', 'resnet_tictactoe', 'mlp_synthetic', 'resnet_downsample'])
def test_inference_matches_reference_modules(name):
    z, net = load_net(name)
    obs = torch.as_tensor(z['obs'], dtype=torch.float32)
    logits, hidden, value = net.initial_inference(obs)
    assert hidden.shape == z['init_hidden'].shape
    np.testing.assert_allclose(logits.numpy(), z['init_logits'], atol=TOL, rtol=0)
    np.testing.assert_allclose(hidden.numpy(), z['init_hidden'], atol=HTOL, rtol=0)
    np.testing.assert_allclose(value.numpy(), z['init_value'], atol=TOL, rtol=0)
    assert value.dtype == torch.float64
    enc = net.encode_actions(torch.as_tensor(z['actions']))
    h_in = torch.as_tensor(z['init_hidden'])
    l2, h2, v2, r2 = net.recurrent_inference(h_in, enc)
    np.testing.assert_allclose(l2.numpy(), z['rec_logits'], atol=TOL, rtol=0)
    np.testing.assert_allclose(h2.numpy(), z['rec_hidden'], atol=HTOL, rtol=0)
    np.testing.assert_allclose(v2.numpy(), z['rec_value'], atol=TOL, rtol=0)
    np.testing.assert_allclose(r2.numpy(), z['rec_reward'], atol=TOL, rtol=0)
    # predictions are those of the INPUT state (network.py:121): same as the initial ones
    np.testing.assert_allclose(l2.numpy(), z['init_logits'], atol=TOL, rtol=0)
    l3, h3, v3, r3 = net.recurrent_inference(torch.as_tensor(z['rec_hidden']), enc)
    np.testing.assert_allclose(h3.numpy(), z['rec2_hidden'], atol=HTOL, rtol=0)
    np.testing.assert_allclose(r3.numpy(), z['rec2_reward'], atol=TOL, rtol=0)
    np.testing.assert_allclose(v3.numpy(), z['rec2_value'], atol=TOL, rtol=0)


def test_recorded_search_outputs_reproduced():
    """The per-simulation outputs recorded while the reference searched are reproduced from
    (hidden_states[parent], action) by the lowered network."""
    for name in ('mlp_cartpole', 'resnet_tictactoe'):
        z, net = load_net(name)
        p = 'r0_'
        parents, actions = z[p + 'parents'][1:], z[p + 'action_from_parent'][1:]
        hid = torch.as_tensor(z[p + 'hidden_states'].reshape(len(parents) + 1, -1))[parents]
        l, h, v, r = net.recurrent_flat(hid, net.encode_actions(torch.as_tensor(actions)))
        np.testing.assert_allclose(l.numpy(), z[p + 'rec_logits'], atol=TOL, rtol=0)
        np.testing.assert_allclose(h.numpy(), z[p + 'rec_hidden'].reshape(len(parents), -1), atol=HTOL, rtol=0)
        np.testing.assert_allclose(v.numpy(), z[p + 'rec_value'], atol=TOL, rtol=0)
        np.testing.assert_allclose(r.numpy(), z[p + 'rec_reward'], atol=TOL, rtol=0)


def test_support_to_scalar_and_normalisation_vectors():
    z = load('misc.npz')
    for L in (1, 10, 300):
        got = support_to_scalar(torch.as_tensor(z[f'sts_logits_{L}']), L).numpy()
        np.testing.assert_allclose(got, z[f'sts_out_{L}'], rtol=1e-12, atol=1e-12)
    got = minmax_normalize(torch.as_tensor(z['norm_mlp_in']), 1).numpy()
    assert np.array_equal(got, z['norm_mlp_out'])
    x = torch.as_tensor(z['norm_res_in'])
    got = minmax_normalize(x.reshape(x.shape[0], -1), x.shape[1]).numpy().reshape(z['norm_res_out'].shape)
    assert np.array_equal(got, z['norm_res_out'])


def test_random_state_dict_has_reference_names_and_shapes():
    for name in ('mlp_cartpole', 'resnet_tictactoe', 'mlp_synthetic'):
        z = load(f'net_{name}.npz')
        ref = {k[3:]: z[k].shape for k in z.files if k.startswith('sd_')}
        sd = random_state_dict(net_config(name), seed=1)
        assert set(sd) == set(ref)
        for k in ref:
            assert tuple(sd[k].shape) == tuple(ref[k]), k
        MuZeroNetwork(net_config(name), sd)  # compiles


@pytest.mark.parametrize('name', ['mlp_cartpole', 'mlp_synthetic', 'resnet_tictactoe'])
def test_packed_program_checker_matches_reference_outputs(name):
    """The packed (level-scheduled, transposed) form of the network, run by the C checker, against
    the reference modules' recorded outputs."""
    from muzero_amd.packing import pack_network
    from oracle import oracle as orc
    z, net = load_net(name)
    packed = pack_network(net)
    assert packed.recurrent.macs == net.recurrent_macs()
    hid = z['init_hidden'].reshape(len(z['actions']), -1)
    lg, nh, val, rew = orc.net_recurrent_batch(packed, hid, z['actions'])
    np.testing.assert_allclose(lg, z['rec_logits'], atol=TOL, rtol=0)
    np.testing.assert_allclose(nh, z['rec_hidden'].reshape(nh.shape), atol=HTOL, rtol=0)
    np.testing.assert_allclose(val, z['rec_value'], atol=TOL, rtol=0)
    np.testing.assert_allclose(rew, z['rec_reward'], atol=TOL, rtol=0)


def test_downsampled_network_lowers_dynamics_and_prediction_only():
    """`downsample` on (network.py:228-325): the representation stays a stack of torch convolutions over the full
    frames, the programs every simulation runs are dense matrices over the 3 x 3 downsampled state, and a random
    checkpoint has the reference's parameter names and shapes."""
    z, net = load_net('resnet_downsample')
    assert net.rep.steps is None and net.hidden_shape == (16, 3, 3) and net.obs_size == 3 * 48 * 48
    assert all(s.W.shape[1] <= 16 * 9 + 9 for s in list(net.dyn.steps) + list(net.pred.steps))
    sd = random_state_dict(net_config('resnet_downsample'), seed=3)
    ref = {k[3:]: z[k].shape for k in z.files if k.startswith('sd_')}
    assert {k: tuple(v.shape) for k, v in sd.items()} == ref


def test_action_plane_smaller_than_action_count_is_rejected():
    """The ResNet dynamics input encodes an action as one cell of a board-sized plane
    (games/tictactoe.py:66-69): more actions than cells cannot be encoded."""
    from muzero_amd.network import MuZeroNetwork
    cfg = net_config('resnet_tictactoe')
    cfg.observation_dim = [3, 2, 2]
    with pytest.raises(ValueError):
        MuZeroNetwork(cfg, seed=0)


def test_network_with_more_jobs_than_a_wave_holds_gets_no_tile_schedule():
    """A wave of the tile-resident matrix-core kernels keeps its job descriptors in one register per field, 64 jobs at
    most: a network that needs more (here 300 column tiles dealt to the 4 waves of a 16-tree tile) gets an EMPTY
    schedule for that tile size - the engine then evaluates it level by level - instead of an error."""
    from muzero_amd.packing import pack_network_mfma, DESC_INTS
    ncfg = SimpleNamespace(network='resnet', observation_dim=[3, 4, 4], n_actions=9, n_stacked_observations=1, blocks=3,
                           channels=24, reduced_channels_reward=8, reduced_channels_policy=8, reduced_channels_value=8,
                           resnet_fc_reward_layers=[16], resnet_fc_policy_layers=[16], resnet_fc_value_layers=[16],
                           downsample=False, support_limit=10)
    pm, io = pack_network_mfma(MuZeroNetwork(ncfg, seed=0))
    h = pm.sched[:4]
    n8, n4 = (h[2] - h[1]) // DESC_INTS, (len(pm.sched) - h[3]) // DESC_INTS
    assert n8 > 0 and n4 == 0
    assert not pm.sched[h[2]:h[3]].any()          # job ranges of the empty schedule: all zero
    assert len(pm.ctinfo) // 3 >= 256             # the column tiles are all there for the level-by-level kernels
