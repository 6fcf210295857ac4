"""Where the host's time goes in one drop-in MCTS.search call (one TicTacToe root, 25 PUCT simulations, host inputs and
outputs): cProfile over 200 calls, sorted by own time.   python scripts/profile_search_single.py"""
import cProfile, os, pstats, sys, time
import numpy as np, torch
sys.path.insert(0, '.')
import bench
from muzero_amd.network import MuZeroNetwork
sys.path.insert(0, os.path.join(bench.REPO, 'muzero_amd'))
from mcts.mcts import MCTS

cfg = bench.search_config(False, 9, 25, 2, 1)
cfg.dirichlet_alpha = 0.2
net = MuZeroNetwork(bench.tictactoe_net_config(), seed=0)
obs0 = np.stack([np.zeros((3, 3)), np.zeros((3, 3)), np.full((3, 3), -1.0)])
enc = lambda a: np.eye(9)[a].reshape(3, 3)  # noqa: E731
np.random.seed(0)
m = MCTS(cfg)
for _ in range(20):
    m.search(net, obs0, np.ones(9), enc, 0, 1.0)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200):
    m.search(net, obs0, np.ones(9), enc, 0, 1.0)
torch.cuda.synchronize()
print('ms per call: %.3f' % ((time.perf_counter() - t0) / 200 * 1e3))
pr = cProfile.Profile()
pr.enable()
for _ in range(200):
    m.search(net, obs0, np.ones(9), enc, 0, 1.0)
pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(25)
