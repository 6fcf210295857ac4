"""GPU: the training step against three optimisation steps performed with the reference's own helper
functions (tests/golden/trainer.npz), and the whole device loop self-play -> trajectories -> batch -> step."""
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from tiny_net import TinyNet

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), 'golden')


def test_training_steps_match_reference_helpers():
    from muzero_amd.trainer import Trainer
    z = np.load(os.path.join(GOLD, 'trainer.npz'))
    for ci in range(int(z['n_cases'])):
        A, L, U, Bs, adam = (int(x) for x in z[f'c{ci}_meta'])
        cfg = SimpleNamespace(lr=0.007, lr_decay_rate=0.9, lr_decay_steps=10, weight_decay=1e-4, momentum=0.9,
                              support_limit=L, value_loss_weight=0.25, optimizer='Adam' if adam else 'SGD')
        net = TinyNet(4, A, 8, L, seed=ci)
        net.load_state_dict({k: torch.as_tensor(z[f'c{ci}_w0_{k}']) for k in net.state_dict()})
        net.to('cuda')
        tr = Trainer(net, cfg)
        for step in range(3):
            batch = tuple(torch.as_tensor(z[f'c{ci}_s{step}_{n}'], device='cuda') for n in ('obs', 'acts', 'vt', 'rt', 'pt'))
            got = tr.step(batch)
            want = z[f'c{ci}_losses'][step]
            np.testing.assert_allclose(got, want[:4], rtol=2e-5, atol=1e-6)
            assert abs(tr.optimizer.param_groups[0]['lr'] - want[4]) < 1e-12
        for k, v in net.state_dict().items():
            np.testing.assert_allclose(v.cpu().numpy(), z[f'c{ci}_w3_{k}'], rtol=1e-4, atol=2e-6, err_msg=k)
        assert tr.training_step == 3


def test_device_loop_selfplay_to_training_step():
    """Self-play on the device, finished games into the trajectory store, a sampled batch, one step."""
    import bench
    from muzero_amd.network import MuZeroNetwork
    from muzero_amd.self_play import BatchedSelfPlay
    from muzero_amd.targets import TrajectoryStore
    from muzero_amd.trainer import Trainer
    cfg = bench.search_config(False, 2, 10, 1, 0.997)
    cfg.max_moves = 12
    net = MuZeroNetwork(bench.cartpole_net_config(), seed=0)
    sp = BatchedSelfPlay(cfg, 'cartpole', net, 64, seed=0)
    games = sp.play(14)
    assert len(games) >= 32
    store = TrajectoryStore(games, 2)
    np.random.seed(0)
    batch = store.sample(32, 5, 0.997, 5)
    obs, act, val, rew, pol = batch
    assert obs.shape == (32, 1, 1, 4) and act.shape == (32, 6, 2) and pol.shape == (32, 6, 2)
    tcfg = SimpleNamespace(lr=0.01, lr_decay_rate=0.9, lr_decay_steps=100, weight_decay=0.0, momentum=0.9,
                           support_limit=10, value_loss_weight=1.0, optimizer='Adam')
    student = TinyNet(4, 2, 8, 10, seed=1).to('cuda')
    tr = Trainer(student, tcfg)
    first = tr.step(batch)[0]
    for _ in range(30):
        last = tr.step(batch)[0]
    assert np.isfinite(first) and last < first       # the same batch is being fitted
