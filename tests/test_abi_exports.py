"""CPU-only: the C-ABI library loads and exports every symbol include/muzero_amd.h declares."""
import ctypes
import os
import re

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(REPO, 'include', 'muzero_amd.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(mz_[a-z_]+)\s*\(', text)))


def test_header_symbols_exported():
    from muzero_amd import engine
    L = engine.lib()
    names = _declared()
    assert len(names) >= 15
    for n in names:
        assert hasattr(L, n), f'{n} declared in include/muzero_amd.h but not exported'
    assert set(names) == set(engine.EXPORTS)
    assert L.mz_version() >= 1


def test_considered_visits_host_entry():
    from muzero_amd import engine
    from helpers import load
    z = load('misc.npz')
    for m, S in z['cv_pairs']:
        assert engine.considered_visits(int(m), int(S)) == tuple(int(x) for x in z[f'cv_{m}_{S}'])


def test_create_rejects_bad_config_without_gpu_work():
    from muzero_amd import engine
    cfg = engine.MzConfig()
    cfg.n_actions, cfg.n_simulations, cfg.players, cfg.max_batch = 65, 10, 1, 4
    assert not engine.lib().mz_create(ctypes.byref(cfg))
    assert b'n_actions' in engine.lib().mz_last_error()
    cfg.n_actions, cfg.players = 4, 3
    assert not engine.lib().mz_create(ctypes.byref(cfg))
    assert b'players' in engine.lib().mz_last_error()
