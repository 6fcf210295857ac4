"""GPU box: is the Atari-shaped level-by-level search bound by the host's launch rate?  Times the call that issues a
search's 600 launches (it returns when they are queued) against the time until the device has finished them."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, '.')
import bench
from muzero_amd.engine import SearchEngine
from muzero_amd.network import MuZeroNetwork
from muzero_amd.packing import pack_network_mfma

B, S, A = 1024, 200, 18
cfg = bench.search_config(False, A, S, 1, 0.997)
net = MuZeroNetwork(bench.atari_like_net_config(), seed=0).to('cuda')
eng = SearchEngine(cfg, net.H, B, seed=0)
eng.set_network_mfma(*pack_network_mfma(net))
rng = np.random.default_rng(1)
obs = torch.as_tensor(rng.standard_normal((B, 128)).astype(np.float32), device='cuda')
lg0, hid0, val0 = net.initial_flat(obs)
di = rng.dirichlet([0.25] * A, size=B)
prior = eng.root_prepare(lg0, np.ones((B, A)), di, 0.25)
issue, total = [], []
for rep in range(8):
    eng.reset(prior, val0, hid0, np.zeros(B, np.int32), np.ones((B, A)))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.search_mfma()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    issue.append((t1 - t0) * 1e3); total.append((t2 - t0) * 1e3)
print('launches issued in %.2f ms (median), search finished after %.2f ms: the host %s' %
      (np.median(issue), np.median(total), 'keeps ahead of the device' if np.median(issue) < 0.8 * np.median(total) else 'is the bound'))
