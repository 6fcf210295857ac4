"""Diagnostic: phase cycles of the level-by-level whole search on the Atari-shaped workload (needs the MZ_STAMPS build:
python -m muzero_amd.build --stamps).  Slots: level kernels 0/1 of the first 256 blocks, tree kernel of the first 256."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, '.')
os.environ['MUZERO_AMD_LIB'] = os.path.abspath(os.environ.get('MZ_STAMPS_LIB', 'muzero_amd/libmuzero_amd_stamps.so'))
os.environ.setdefault('MZ_LEVEL_RANGES', '1')
import bench
from muzero_amd.engine import SearchEngine, lib
from muzero_amd.network import MuZeroNetwork
from muzero_amd.packing import pack_network_mfma

B, S, A = 1024, int(os.environ.get('SIMS', 40)), 18
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
eng.reset(eng.root_prepare(lg0, np.ones((B, A)), di, 0.25), val0, hid0, np.zeros(B, np.int32), np.ones((B, A)))
eng.search_mfma()
buf = np.zeros((1024, 8), np.int64)
f = lib().mz_debug_read_net_stamps
f.argtypes = [C.c_void_p, C.c_int]
assert f(buf.ctypes.data_as(C.c_void_p), buf.size) == 0
lv = ['descriptor + loads issued', 'inputs copied to LDS', 'barrier', 'matrix instructions']
for l in range(2):
    x = buf[l * 256:(l + 1) * 256, :4]
    print(f'k_net_level_rows level {l}, cycles per block (mean / max over 256 blocks): ' +
          '  '.join('%s %.0f/%.0f' % (n, v, w) for n, v, w in zip(lv, x.mean(0), x.max(0))) +
          '  (since the kernel began: arguments there after %.0f, descriptor after %.0f)' % tuple(buf[l * 256:(l + 1) * 256, 4:6].mean(0)))
tr = ['selection + tables', 'epilogue', 'expansion + backup', 'descent', 'inputs staged']
x = buf[512:768]
print('k_leaf_backup_select (last simulation but one), cycles per block: ' +
      '  '.join('%s %.0f/%.0f' % (n, v, w) for n, v, w in zip(tr, x[:, :5].mean(0), x[:, :5].max(0))) +
      '  depth of the new selection %.2f' % x[:, 5].mean() +
      '  (cached decisions: levels scored in full %.2f per descent, walk %.0f/%.0f cycles)' % (x[:, 6].mean(), x[:, 7].mean(), x[:, 7].max()))
