"""Diagnostic: per-phase cycle shares of the fused search kernel (needs the -DMZ_STAMPS build)."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, '.')
os.environ['MUZERO_AMD_LIB'] = os.path.abspath('muzero_amd/libmuzero_amd_stamps.so')
import bench
from muzero_amd.engine import SearchEngine, lib
from muzero_amd.network import MuZeroNetwork
from muzero_amd.packing import pack_network

for B, lanes in ((256, 64), (8192, 16)):
    cfg = bench.search_config(False, 2, 50, 1, 0.997)
    net = MuZeroNetwork(bench.cartpole_net_config(), seed=0)
    packed = pack_network(net)
    eng = SearchEngine(cfg, packed.H, B, seed=0, lanes_per_tree=lanes)
    eng.set_network(packed)
    rng = np.random.default_rng(1)
    obs = rng.uniform(-0.05, 0.05, size=(B, 4)).astype(np.float32)
    di = rng.dirichlet([0.25] * 2, size=B)
    for _ in range(3):
        eng.begin_net(obs, np.ones((B, 2)), np.zeros(B, np.int32), di, 0.25)
        eng.search_net()
    buf = np.zeros((B, 50), np.int32)
    lib().mz_debug_read_path(C.c_void_p(eng._ctx), buf.ctypes.data_as(C.c_void_p), C.c_size_t(buf.nbytes))
    st = buf.view(np.int64).reshape(B, 25)[:, :6].astype(np.float64)
    tot = st.sum(1).mean()
    print(f'B={B} lanes={lanes}: cycles/search {tot:.0f}  select {st[:,0].mean()/tot:.2%}  dense {st[:,1].mean()/tot:.2%} '
          f'epilogue {st[:,2].mean()/tot:.2%}  expand+backup {st[:,3].mean()/tot:.2%}  [select, dense, sts, backup, stage-in, minmax] per sim: '
          + ' '.join(f'{x/50:.0f}' for x in st.mean(0)))
