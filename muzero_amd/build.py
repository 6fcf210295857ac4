"""Builds the HIP engine in-tree: muzero_amd/libmuzero_amd.so (gfx950 only)."""
from __future__ import annotations

import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'libmuzero_amd.so')
SOURCES = ['engine.hip']
HEADERS = sorted(f for f in os.listdir(CSRC) if f.endswith('.h')) + [os.path.join('..', '..', 'include', 'muzero_amd.h')]
# -ffp-contract=off: the tree arithmetic must round exactly like the reference's NumPy expressions.
FLAGS = ['--offload-arch=gfx950', '-O3', '-ffp-contract=off', '-fPIC', '-shared', '-std=c++17']


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def build(force: bool = False) -> str:
    if force or needs_build():
        hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
        cmd = [hipcc, *FLAGS, *[os.path.join(CSRC, s) for s in SOURCES], '-o', LIB]
        subprocess.check_call(cmd)
    return LIB


if __name__ == '__main__':
    print(build(force=True))
