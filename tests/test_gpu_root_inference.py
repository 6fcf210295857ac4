"""GPU: network.initial_inference (network.py:86-103) on the device - representation network, hidden-state scaling,
prediction network and support_to_scalar through the matrix-core level kernels (mz_initial_inference_mfma) - against
the outputs the reference's own modules recorded for the same weights and observations (tests/golden/net_*.npz) and,
on ragged batches, against the lowered network run by torch."""
import numpy as np
import pytest
import torch

from helpers import make_config
from test_network_lowering import load_net, net_config
from muzero_amd.network import MuZeroNetwork

pytestmark = pytest.mark.gpu
TOL, HTOL = 1e-5, 1e-4


def _engine(net, B, gumbel=True):
    from muzero_amd.engine import SearchEngine
    from muzero_amd.packing import pack_root_network_mfma
    eng = SearchEngine(make_config(gumbel, net.A, 8, 2, 1), net.H, B, seed=0)
    eng.set_root_network_mfma(*pack_root_network_mfma(net))
    return eng


@pytest.mark.parametrize('name', ['mlp_cartpole', 'resnet_tictactoe', 'mlp_synthetic'])
def test_initial_inference_matches_the_reference_modules(name):
    z, net = load_net(name)
    obs = np.asarray(z['obs'], np.float32)
    B = obs.shape[0]
    eng = _engine(net, B)
    logits, hidden, value = eng.initial_inference_mfma(obs)
    np.testing.assert_allclose(logits.cpu().numpy(), z['init_logits'], atol=TOL, rtol=0)
    np.testing.assert_allclose(hidden.cpu().numpy(), z['init_hidden'].reshape(B, -1), atol=HTOL, rtol=0)
    np.testing.assert_allclose(value.cpu().numpy(), z['init_value'], atol=TOL, rtol=0)
    eng.close()


@pytest.mark.parametrize('name,B', [('resnet_tictactoe', 1), ('resnet_tictactoe', 333), ('mlp_synthetic', 70)])
def test_initial_inference_on_ragged_batches(name, B):
    z, net0 = load_net(name)
    net = MuZeroNetwork(net_config(name), net0.state_dict()).to('cuda')
    rng = np.random.default_rng(B)
    obs = rng.uniform(-1, 1, size=(B, net.obs_size)).astype(np.float32)
    eng = _engine(net, 400)                      # a context larger than the batch
    logits, hidden, value = eng.initial_inference_mfma(obs)
    with torch.no_grad():
        l2, h2, v2 = net.initial_flat(torch.as_tensor(obs, device='cuda'))
    np.testing.assert_allclose(logits.cpu().numpy(), l2.cpu().numpy(), atol=TOL, rtol=0)
    np.testing.assert_allclose(hidden.cpu().numpy(), h2.cpu().numpy(), atol=HTOL, rtol=0)
    np.testing.assert_allclose(value.cpu().numpy(), v2.cpu().numpy(), atol=TOL, rtol=0)
    # the same call again (the activation rows are reused) and a smaller batch after a larger one
    l3, h3, v3 = eng.initial_inference_mfma(obs[: max(1, B // 2)])
    np.testing.assert_array_equal(l3.cpu().numpy(), logits.cpu().numpy()[: max(1, B // 2)])
    np.testing.assert_array_equal(h3.cpu().numpy(), hidden.cpu().numpy()[: max(1, B // 2)])
    np.testing.assert_array_equal(v3.cpu().numpy(), value.cpu().numpy()[: max(1, B // 2)])
    eng.close()


def test_initial_inference_needs_both_programs():
    from muzero_amd.engine import SearchEngine, EngineError
    z, net = load_net('resnet_tictactoe')
    eng = SearchEngine(make_config(True, net.A, 8, 2, 1), net.H, 8, seed=0)
    with pytest.raises(EngineError):
        eng.initial_inference_mfma(np.zeros((8, net.obs_size), np.float32))
    eng.close()
