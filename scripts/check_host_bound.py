"""Diagnostic: host time to enqueue one cfg2 step (roots + search with the final rule) against the time the GPU needs
for it - the benchmark is only meaningful while the first stays below the second."""
import sys, time, os
sys.path.insert(0, '.')

import numpy as np, torch
import bench
from muzero_amd.engine import SearchEngine
from muzero_amd.network import MuZeroNetwork
from muzero_amd.packing import pack_network
B=256
cfg = bench.search_config(False, 2, 50, 1, 0.997)
net = MuZeroNetwork(bench.cartpole_net_config(), seed=0)
packed = pack_network(net)
eng = SearchEngine(cfg, packed.H, B, seed=0)
eng.set_network(packed)
dev=eng.device
rng = np.random.default_rng(1)
obs = torch.as_tensor(rng.uniform(-0.05, 0.05, size=(B, 4)).astype(np.float32), device=dev)
legal = torch.ones((B,2),dtype=torch.uint8,device=dev); tp=torch.zeros(B,dtype=torch.int32,device=dev)
di = torch.as_tensor(rng.dirichlet([0.25]*2,size=B),device=dev); u=torch.as_tensor(rng.random(B),device=dev)
def step():
    eng.begin_net(obs, legal, tp, di, 0.25, None, tree_id0=0)
    return eng.search_net_act(1.0, u)
for _ in range(50): step()
torch.cuda.synchronize()
N=2000
# a short burst first: with few launches in flight the queue never pushes back, so this is the host's own cost
torch.cuda.synchronize()
b0=time.perf_counter()
for _ in range(20): step()
b1=time.perf_counter()
torch.cuda.synchronize()
print('host cost per step (burst of 20) us', (b1-b0)/20*1e6)
t0=time.perf_counter()
for _ in range(N): step()
t1=time.perf_counter()
torch.cuda.synchronize()
t2=time.perf_counter()
print('enqueue per step us', (t1-t0)/N*1e6, 'total per step us', (t2-t0)/N*1e6)
