"""Builds the HIP engine in-tree: muzero_amd/libmuzero_amd.so (gfx950 only)."""
from __future__ import annotations

import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'libmuzero_amd.so')
OBJ_DIR = os.path.join(HERE, 'build')
# translation units: the C ABI with the tree / fused-network kernels, and the matrix-core kernels
SOURCES = ['engine.hip', 'mfma_kernels.hip', 'chain_kernels.hip', 'cached_kernels_a.hip', 'cached_kernels_b.hip',
           'cached_kernels_c.hip']
HEADERS = sorted(f for f in os.listdir(CSRC) if f.endswith('.h')) + [os.path.join('..', '..', 'include', 'muzero_amd.h')]
# -ffp-contract=off: the tree arithmetic must round exactly like the reference's NumPy expressions.
FLAGS = ['--offload-arch=gfx950', '-ffp-contract=off', '-fPIC', '-std=c++17']
# Optimisation level per translation unit, as measured (same box, both builds): the whole-search kernels of the matrix-core
# units are larger than the 64 KB instruction cache, and optimising for size makes them 0.7 - 0.9 % faster (TicTacToe
# 16384 x 100: 7.15 -> 7.10 ms, Atari-shaped 1024 x 200: 7.84 -> 7.78 ms); the register-resident tree kernels of
# engine.hip lose 6 % that way (CartPole 256 x 50) and stay at -O3.
# The cached-decision search kernels gain another 0.5 % from the compiler's iterative scheduler for instruction-level
# parallelism (16384 x 100: 6.768 -> 6.730 ms; "max-ilp" 6.737, "max-memory-clause" 6.816); the level-by-level kernels of
# mfma_kernels.hip lose 1.6 - 2.7 % under each of the three and keep the default scheduler.
_CACHED = '-Os -mllvm --amdgpu-sched-strategy=iterative-ilp'
OPT = {'engine.hip': '-O3', 'chain_kernels.hip': '-O3', 'cached_kernels_a.hip': _CACHED, 'cached_kernels_b.hip': _CACHED,
       'cached_kernels_c.hip': _CACHED}
OPT_DEFAULT = '-Os'


def source_fingerprint() -> str:
    """sha256 (16 hex digits) over the compiler flags and the kernel sources - every .hip and .h under csrc/ and the
    public header - in name order.  Measurements that belong to one state of the kernels (HBM traffic from the counter passes) carry it, so
    that a reader can tell whether they still describe the code at hand."""
    import hashlib
    h = hashlib.sha256()
    h.update(repr((FLAGS, sorted(OPT.items()), OPT_DEFAULT)).encode())   # the same sources built differently are other kernels
    for f in sorted(SOURCES + HEADERS):
        h.update(os.path.basename(f).encode())
        with open(os.path.join(CSRC, f), 'rb') as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def needs_build(lib: str = LIB) -> bool:
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


# headers a translation unit does NOT include (everything else counts as a dependency)
_TREE_ONLY = {'search_net.h', 'search_reg.h', 'env_step.h', 'targets.h'}
NOT_INCLUDED = {'engine.hip': {'net_mfma.h', 'search_cached.h', 'search_chain.h', 'cached_launch.h'},
                'mfma_kernels.hip': _TREE_ONLY | {'search_chain.h', 'cached_launch.h'},
                'chain_kernels.hip': _TREE_ONLY | {'search_cached.h', 'cached_launch.h'},
                'cached_kernels_a.hip': _TREE_ONLY | {'search_chain.h'},
                'cached_kernels_b.hip': _TREE_ONLY | {'search_chain.h'},
                'cached_kernels_c.hip': _TREE_ONLY | {'search_chain.h'}}


def _stale(src: str, obj: str) -> bool:
    if not os.path.exists(obj):
        return True
    t = os.path.getmtime(obj)
    deps = [src] + [h for h in HEADERS if os.path.basename(h) not in NOT_INCLUDED.get(src, set())]
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in deps)


def build(force: bool = False, defines=(), lib: str = LIB) -> str:
    """Compiles the translation units that changed (all of them with `force`), in parallel, and links the shared
    library.  `defines`: extra -D flags (the diagnostic build with phase stamps uses MZ_STAMPS and another `lib`)."""
    if force or needs_build(lib):
        hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
        os.makedirs(OBJ_DIR, exist_ok=True)
        tag = '_'.join(defines).replace('=', '')
        objs = [os.path.join(OBJ_DIR, s.replace('.hip', f'{"_" + tag if tag else ""}.o')) for s in SOURCES]
        todo = [(s, o) for s, o in zip(SOURCES, objs) if force or _stale(s, o)]

        def compile_one(pair):
            src, obj = pair
            subprocess.check_call([hipcc, *FLAGS, *OPT.get(src, OPT_DEFAULT).split(), *[f'-D{x}' for x in defines], '-c',
                                   os.path.join(CSRC, src), '-o', obj])
        with ThreadPoolExecutor(max(len(todo), 1)) as pool:
            list(pool.map(compile_one, todo))
        subprocess.check_call([hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', *objs, '-o', lib])
    return lib


if __name__ == '__main__':
    import sys
    force = '--changed' not in sys.argv
    if '--stamps' in sys.argv:
        print(build(force=force, defines=('MZ_STAMPS',), lib=os.path.join(HERE, 'libmuzero_amd_stamps.so')))
    else:
        print(build(force=force))
