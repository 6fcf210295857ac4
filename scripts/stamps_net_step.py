"""Diagnostic: phase cycles of the matrix-core leaf-evaluation kernel k_net_step (needs the MZ_STAMPS build:
python -m muzero_amd.build --stamps)."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, '.')
os.environ['MUZERO_AMD_LIB'] = os.path.abspath(os.environ.get('MZ_STAMPS_LIB', 'muzero_amd/libmuzero_amd_stamps.so'))
os.environ['MZ_NET_PATH'] = 'tile'
import bench
from muzero_amd.engine import SearchEngine, lib
from muzero_amd.network import MuZeroNetwork
from muzero_amd.packing import pack_network_mfma

B, S, A = 16384, 8, 9
cfg = bench.search_config(True, A, S, 2, 1)
net = MuZeroNetwork(bench.tictactoe_net_config(), seed=0).to('cuda')
eng = SearchEngine(cfg, net.H, B, seed=0)
eng.set_network_mfma(*pack_network_mfma(net))
rng = np.random.default_rng(1)
obs = torch.as_tensor(bench.tictactoe_roots(rng, B).reshape(B, 27), device='cuda')
lg0, hid0, val0 = net.initial_flat(obs)
eng.reset(eng.root_prepare(lg0, np.ones((B, A))), val0, hid0, np.zeros(B, np.int32), np.ones((B, A)), rng.gumbel(size=(B, A)))
eng.select(want_hidden=False)
for _ in range(S):
    eng.expand_backup_net(select_next=True)
buf = np.zeros((512, 8), np.int64)
f = lib().mz_debug_read_net_stamps
f.argtypes = [C.c_void_p, C.c_int]
assert f(buf.ctypes.data_as(C.c_void_p), buf.size) == 0
names = ['zero the tile', 'read the schedule', 'stage inputs + first weights', 'barrier', 'levels', 'epilogue']
print('k_net_step, cycles per block (mean over 512 blocks): ' + '  '.join('%s %.0f' % (n, v) for n, v in zip(names, buf[:, :6].mean(0))))
