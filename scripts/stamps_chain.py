"""Diagnostic: where the waves of the chained whole search spend their cycles on the Atari-shaped workload (needs the
MZ_STAMPS build: python -m muzero_amd.build --stamps).  Network wave 0 and tree wave 4 of every workgroup, cycles
summed over the search and divided by the simulations."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, '.')
os.environ['MUZERO_AMD_LIB'] = os.path.abspath(os.environ.get('MZ_STAMPS_LIB', 'muzero_amd/libmuzero_amd_stamps.so'))
os.environ['MZ_SEARCH_CHAIN'] = '2'
import bench
from muzero_amd.engine import SearchEngine, lib
from muzero_amd.network import MuZeroNetwork
from muzero_amd.packing import pack_network_mfma

B, S, A = int(os.environ.get('TREES', 1024)), int(os.environ.get('SIMS', 200)), 18
cfg = bench.search_config(False, A, S, 1, 0.997)
net = MuZeroNetwork(bench.atari_like_net_config(), seed=0)
if os.environ.get('PEAKED'):   # deep trees: one action preferred by a wide margin
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    sd[sorted(k for k in sd if 'policy_network' in k and k.endswith('bias'))[-1]][7] += 6.0
    net = MuZeroNetwork(bench.atari_like_net_config(), sd)
net = net.to('cuda')
eng = SearchEngine(cfg, net.H, B, seed=0)
eng.set_network_mfma(*pack_network_mfma(net))
rng = np.random.default_rng(1)
obs = torch.as_tensor(rng.standard_normal((B, 128)).astype(np.float32), device='cuda')
lg0, hid0, val0 = net.initial_flat(obs)
di = rng.dirichlet([0.25] * A, size=B)
prior = eng.root_prepare(lg0, np.ones((B, A)), di, 0.25)
for rep in range(3):
    eng.reset(prior, val0, hid0, np.zeros(B, np.int32), np.ones((B, A)))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); eng.search_mfma(); e1.record(); torch.cuda.synchronize()
print('search %.3f ms (diagnostic build: read the shares, not the length)' % e0.elapsed_time(e1))
eng.check()
buf = np.zeros((256, 16), np.int64)
f = lib().mz_debug_read_chain_stamps
f.argtypes = [C.c_void_p, C.c_int]
assert f(buf.ctypes.data_as(C.c_void_p), buf.size) == 0
live = buf[buf[:, :8].sum(1) > 0]
nn = ['look (wait for inputs)', 'staging fetch', 'staged -> barrier arrive', 'barrier', 'matrix instructions', 'outputs']
# slots: 0 look, 1 fetch, 2 barrier, 3 mfma, 4 out
names_net = ['wait for inputs', 'staging fetch', 'barrier', 'matrix instructions', 'outputs + stores']
x = live[:, :5] / S
print('network wave 0, cycles per simulation (mean / max over workgroups): ' +
      '  '.join('%s %.0f/%.0f' % (n, v, w) for n, v, w in zip(names_net, x.mean(0), x.max(0))) + '  total %.0f' % x.sum(1).mean())
names_tree = ['wait for outputs', 'sweep', 'epilogue', 'expansion + backup', 'descent', 'inputs staged + stored']
y = live[:, 8:14] / S
print('tree wave 4, cycles per simulation: ' +
      '  '.join('%s %.0f/%.0f' % (n, v, w) for n, v, w in zip(names_tree, y.mean(0), y.max(0))) + '  total %.0f' % y.sum(1).mean())
