"""CPU, world_size 2, gloo: the sharding / timing / gather helpers the multi-GPU path uses."""
import os
import socket

import numpy as np
import torch
import torch.multiprocessing as mp

from muzero_amd import distributed as mzdist


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    r, w = mzdist.init('gloo')
    assert (r, w) == (rank, world)
    B_total, A, OBS = 11, 3, 4          # 11 trees over 2 ranks: uneven shards (6 and 5)
    lo, hi = mzdist.shard_range(B_total, rank, world)
    b = hi - lo
    # every rank's slice of a global, rank-independent result
    g_action = torch.arange(B_total, dtype=torch.int32)
    g_value = torch.arange(B_total, dtype=torch.float64) * 0.5
    g_probs = torch.arange(B_total * A, dtype=torch.float64).reshape(B_total, A)
    g_obs = torch.arange(B_total * OBS, dtype=torch.float32).reshape(B_total, OBS) * 0.25
    g_reward = torch.arange(B_total, dtype=torch.float32) - 3.0
    g_done = (torch.arange(B_total) % 3 == 0).to(torch.uint8)
    g_tp = (torch.arange(B_total) % 2).to(torch.int32)
    assert b == (6 if rank == 0 else 5)
    res = mzdist.gather_root_stats(g_action[lo:hi], g_value[lo:hi], g_probs[lo:hi], dst=0)
    # the whole per-move record of self-play, twice through the same (preallocated) exchange
    sg = mzdist.StepGather(b, {'probs': A, 'obs_before': OBS}, device='cpu', dst=0)
    assert sg.sizes == [6, 5]
    full = dict(action=g_action, value=g_value, probs=g_probs, obs_before=g_obs, reward=g_reward, done=g_done,
                to_play=g_tp)
    for rep in range(2):
        rec = sg.gather({k: (v + rep if k != 'done' else v)[lo:hi] for k, v in full.items()})
        if rank == 0:
            for k, v in full.items():
                want = v + rep if k != 'done' else v
                assert rec[k].dtype == v.dtype and torch.equal(rec[k], want), k
        else:
            assert rec is None
    mzdist.barrier()
    slow = mzdist.max_over_ranks(1.0 + rank)
    total = mzdist.sum_over_ranks(float(b))
    assert slow == float(world) and total == float(B_total)
    # weight refresh: every rank ends up with rank 0's tensors, dtypes kept
    sd = {'w': torch.full((3, 2), float(rank + 1)), 'b': torch.arange(4, dtype=torch.float32) * (rank + 1),
          'steps': torch.tensor(7 * (rank + 1), dtype=torch.int64)}
    out = mzdist.broadcast_weights(sd, src=0)
    assert out is sd and sd['steps'].dtype == torch.int64 and int(sd['steps']) == 7
    assert torch.equal(sd['w'], torch.full((3, 2), 1.0)) and torch.equal(sd['b'], torch.arange(4, dtype=torch.float32))
    if rank == 0:
        a, v, p = res
        assert torch.equal(a, g_action) and torch.equal(v, g_value) and torch.equal(p, g_probs)
        open(os.path.join(out_dir, 'ok'), 'w').write('ok')
    else:
        assert res is None
    torch.distributed.destroy_process_group()


def test_two_ranks_gloo(tmp_path):
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / 'ok').read_text() == 'ok'


def test_shard_range_covers_everything():
    for n in (0, 1, 7, 8, 65536):
        for world in (1, 2, 3, 8):
            spans = [mzdist.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
