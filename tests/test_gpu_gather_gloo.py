"""GPU, two processes on one card, gloo: the per-move gather of self-play records (muzero_amd/distributed.py, in place
of the reference's per-actor pushes to the replay buffer, self_play.py:58) with the records on the DEVICE and the
exchange buffer on the host - the layout that MUZERO_AMD_DIST_BACKEND=gloo gives.  The values change every move: a
copy into the host buffer that had not finished when gloo read it would deliver the previous move's records."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

from muzero_amd import distributed as mzdist

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), LOCAL_RANK='0', WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port))
    mzdist.init('gloo')
    dev = torch.device('cuda:0')
    B_total, A, OBS, moves = 4099, 18, 64, 40      # uneven shards, records of a few hundred KB
    lo, hi = mzdist.shard_range(B_total, rank, world)
    b = hi - lo
    sg = mzdist.StepGather(b, {'probs': A, 'obs_before': OBS}, device=dev, dst=0)
    base = torch.arange(B_total, device=dev)
    bad = 0
    for mv in range(moves):
        # a kernel's worth of work in front of every record, so that the copies have something to overtake
        noise = torch.randn(512, 512, device=dev)
        noise = noise @ noise
        full = dict(action=(base + mv).to(torch.int32), value=(base * 0.5 + mv).to(torch.float64),
                    probs=(base[:, None] * A + torch.arange(A, device=dev) + mv).to(torch.float64),
                    obs_before=(base[:, None] * 0.25 + torch.arange(OBS, device=dev) - mv).to(torch.float32),
                    reward=(base - 3.0 * mv).to(torch.float32), done=((base + mv) % 3 == 0).to(torch.uint8),
                    to_play=((base + mv) % 2).to(torch.int32))
        rec = sg.gather({k: v[lo:hi] for k, v in full.items()})
        if rank == 0:
            for k, v in full.items():
                if not torch.equal(rec[k].to(dev), v):
                    bad += 1
        else:
            assert rec is None
    mzdist.barrier()
    if rank == 0:
        open(os.path.join(out_dir, 'result'), 'w').write(str(bad))
    torch.distributed.destroy_process_group()


def test_device_records_through_a_host_exchange(tmp_path):
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / 'result').read_text() == '0'
