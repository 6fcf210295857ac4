"""Where the host's time goes in the drop-in batch call (MCTS.search_batch, 16384 TicTacToe roots x 100 Gumbel simulations):
cProfile over 20 calls, sorted by own time.   python scripts/profile_search_batch.py"""
import cProfile, os, pstats, sys, time
import numpy as np, torch
sys.path.insert(0, '.')
import bench
from muzero_amd.network import MuZeroNetwork
sys.path.insert(0, os.path.join(bench.REPO, 'muzero_amd'))
from mcts.mcts import MCTS

B, A, S = 16384, 9, 100
cfg = bench.search_config(True, A, S, 2, 1)
net = MuZeroNetwork(bench.tictactoe_net_config(), seed=0)
rng = np.random.default_rng(1)
obs = torch.as_tensor(bench.tictactoe_roots(rng, B).reshape(B, 27), device='cuda')
legal, tp = np.ones((B, A)), np.zeros(B, np.int32)
np.random.seed(0)
m = MCTS(cfg)
for _ in range(5):
    m.search_batch(net, obs, legal, tp, 1.0)
torch.cuda.synchronize()
t0 = time.perf_counter()
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    m.search_batch(net, obs, legal, tp, 1.0)
pr.disable()
torch.cuda.synchronize()
print('ms per call: %.3f' % ((time.perf_counter() - t0) / 20 * 1e3))
pstats.Stats(pr).sort_stats('tottime').print_stats(18)
