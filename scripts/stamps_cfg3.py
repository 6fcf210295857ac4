"""Diagnostic: per-phase cycle shares of the matrix-core search kernel (needs the -DMZ_STAMPS build)."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, '.')
os.environ['MUZERO_AMD_LIB'] = os.path.abspath(os.environ.get('MZ_STAMPS_LIB', 'muzero_amd/libmuzero_amd_stamps.so'))
import bench
from muzero_amd.engine import SearchEngine, lib
from muzero_amd.network import MuZeroNetwork
from muzero_amd.packing import pack_network_mfma

B, S, A = 16384, 100, 9
cfg = bench.search_config(True, A, S, 2, 1)
net = MuZeroNetwork(bench.tictactoe_net_config(), seed=0).to('cuda')
eng = SearchEngine(cfg, net.H, B, seed=0)
eng.set_network_mfma(*pack_network_mfma(net))
rng = np.random.default_rng(1)
boards = rng.integers(0, 3, size=(B, 9)) - 1
planes = np.stack([(boards == -1), (boards == 1), np.ones_like(boards) * -1], axis=1).astype(np.float32)
obs = torch.as_tensor(planes.reshape(B, 27), device='cuda')
g = rng.gumbel(size=(B, A))
for _ in range(2):
    lg0, hid0, val0 = net.initial_flat(obs)
    prior = eng.root_prepare(lg0, np.ones((B, A)))
    eng.reset(prior, val0, hid0, np.zeros(B, np.int32), np.ones((B, A)), g)
    eng.search_mfma()
buf = np.zeros((B, S), np.int32)
lib().mz_debug_read_path(C.c_void_p(eng._ctx), buf.ctypes.data_as(C.c_void_p), C.c_size_t(buf.nbytes))
cached = os.environ.get('MZ_SEARCH_CACHED', '1') != '0'
n = 5 if cached else 4
fine = buf.view(np.int64).reshape(B, S // 2)[:, 5:21].astype(np.float64) if cached else None
st = buf.view(np.int64).reshape(B, S // 2)[:, :n].astype(np.float64)
names = ['walk/descend+stage', 'barrier wait', 'network', 'epilogue+backup'] + (['decisions'] if cached else [])
print('shader-clock cycles per simulation, mean over trees: ' +
      '  '.join('%s %.0f' % (k, v) for k, v in zip(names, st.mean(0) / S)) + '  total %.0f' % (st.sum(1).mean() / S))
if fine is not None:
    print('   of which: walk %.0f (the rest of the first phase stages the inputs)  epilogue %.0f  expansion + backup %.0f  '
          'wait for the stores %.0f  barrier %.0f  record loads of the re-decision (lane 0 of each tree) %.0f' % tuple(fine[:, :6].mean(0) / S))
    print('   decisions evaluated with the reference operations (no certified lead): %.4f per lane and simulation; waves that ran that path: %.3f per simulation' % tuple(fine[:, 6:8].mean(0) / S))
    print('   inside one decision (cycles from its start, cumulative): loads %.0f  completed values + range %.0f  certified argmax %.0f  end %.0f' % tuple(fine[:, [5, 8, 9, 10]].mean(0) / S) + '  (the decision starts %.0f cycles into its phase; the record fetch has issued its first eight loads after %.0f and holds their data after %.0f, ends after %.0f: between the start of the phase and the fetch lies the wait for the stores of the backup)' % tuple(fine[:, [11, 12, 13, 14]].mean(0) / S))
lv = np.zeros(8 * 1024, np.int64)
f = lib().mz_debug_read_net_stamps
f.argtypes = [C.c_void_p, C.c_int]
if f(lv.ctypes.data_as(C.c_void_p), lv.size) == 0:
    x = lv[7000:7000 + 8 * 16].reshape(8, 8, 2) / (2.0 * S)    # [wave][level][jobs, barrier wait]; two searches ran above
    print('   network levels of block 0, cycles per simulation, jobs/wait at the barrier per wave:')
    for l in range(8):
        if x[:, l].any():
            print('     L%d  ' % l + '  '.join('%5.0f/%-5.0f' % (x[w, l, 0], x[w, l, 1]) for w in range(8)))
    j = lv[7200:7264].reshape(8, 8).astype(np.float64)
    if j[:, 3].any():
        print('   jobs of block 0 per wave: count per simulation, cycles per job before / in / after the matrix instructions:')
        print('     ' + '  '.join('%.1f: %.0f/%.0f/%.0f' % (j[w, 3] / (2 * S), j[w, 0] / j[w, 3], j[w, 1] / j[w, 3], j[w, 2] / j[w, 3]) for w in range(8)))
