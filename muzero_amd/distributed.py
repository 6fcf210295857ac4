"""Multi-GPU sharding of a batch of independent trees (one process per GPU).

Trees never exchange data during a search, so ranks only agree on who owns which trees
(block partition, global tree ids keep the counter-based noise independent of the GPU count)
and, for self-play, gather each step's root statistics on the replay rank.  Works with any
torch.distributed backend ("nccl" = RCCL over xGMI on MI355X, "gloo" on CPU in the tests).
"""
from __future__ import annotations

import os
from typing import Optional, Tuple

import torch
import torch.distributed as dist


def env_rank_world() -> Tuple[int, int, int]:
    return (int(os.environ.get('RANK', 0)), int(os.environ.get('LOCAL_RANK', 0)),
            int(os.environ.get('WORLD_SIZE', 1)))


def init(backend: Optional[str] = None) -> Tuple[int, int]:
    """Joins the process group described by RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT."""
    rank, local_rank, world = env_rank_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if backend is None:  # MUZERO_AMD_DIST_BACKEND=gloo rehearses the multi-rank path on a box with one GPU
            backend = os.environ.get('MUZERO_AMD_DIST_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')
        if backend == 'nccl':
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world


def shard_range(n_total: int, rank: int, world: int) -> Tuple[int, int]:
    """[start, stop) of the trees rank owns: contiguous blocks, remainder to the first ranks."""
    base, rem = divmod(int(n_total), int(world))
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def reduce_device(device):
    """Where small reduction operands live: on the GPU for RCCL, on the host otherwise."""
    return device if dist.is_initialized() and dist.get_backend() == 'nccl' else 'cpu'


def barrier():
    if dist.is_initialized():
        dist.barrier()


def max_over_ranks(value: float, device='cpu') -> float:
    if not dist.is_initialized():
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value: float, device='cpu') -> float:
    if not dist.is_initialized():
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


# The per-move record of self-play: the fields of the reference's GameHistory.save call (self_play.py:107-109,
# games/game.py:99-112) plus the episode-end flag.  Ordered by falling element size so that every field
# starts aligned inside the packed byte buffer.
RECORD_FIELDS = (('value', torch.float64), ('probs', torch.float64), ('obs_before', torch.float32),
                 ('reward', torch.float32), ('action', torch.int32), ('to_play', torch.int32), ('done', torch.uint8))


class StepGather:
    """Gathers every rank's per-move records on the replay rank `dst` (global tree order).

    One `gather` collective per move (RCCL send/recv over xGMI on GPUs; gloo on the host in the CPU tests)
    of one packed byte buffer per rank: field-major, every field padded to the largest shard so ranks with
    uneven shards (`shard_range` gives the first ranks one tree more) exchange equal sizes.  Shard sizes are
    exchanged once, at construction; buffers are allocated once and reused, so a move costs D2D copies into
    the send buffer and the collective.  Fields absent from a record (search-only callers have no
    observation / reward / done) travel as zeros.

    b: trees of this rank; widths: columns of the 2-D fields, e.g. {'probs': A, 'obs_before': obs_size}."""

    def __init__(self, b: int, widths: dict, device='cpu', dst: int = 0):
        self.b, self.dst = int(b), int(dst)
        self.widths = {name: int(widths.get(name, 1)) for name, _ in RECORD_FIELDS}
        self.active = dist.is_initialized() and dist.get_world_size() > 1
        self.out_device = device
        if not self.active:
            self.sizes = [self.b]
            return
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.device = device if dist.get_backend() == 'nccl' else 'cpu'
        mine = torch.tensor([self.b], dtype=torch.int64, device=self.device)
        every = [torch.zeros_like(mine) for _ in range(self.world)]
        dist.all_gather(every, mine)
        self.sizes = [int(x.item()) for x in every]
        self.rows = (max(self.sizes) + 7) // 8 * 8
        self.offsets, off = {}, 0
        for name, dtype in RECORD_FIELDS:
            self.offsets[name] = off
            off += self.rows * self.widths[name] * torch.empty(0, dtype=dtype).element_size()
        self.nbytes = off
        self.send = torch.zeros(self.nbytes, dtype=torch.uint8, device=self.device)
        self.recv = ([torch.zeros(self.nbytes, dtype=torch.uint8, device=self.device) for _ in range(self.world)]
                     if self.rank == self.dst else None)

    def _field(self, buf: torch.Tensor, name: str, dtype) -> torch.Tensor:
        n = self.rows * self.widths[name] * torch.empty(0, dtype=dtype).element_size()
        return buf[self.offsets[name]:self.offsets[name] + n].view(dtype).reshape(self.rows, self.widths[name])

    def gather(self, record: dict):
        """record: name -> tensor with leading dimension b (RECORD_FIELDS names; missing ones are sent as zeros).
        Returns on `dst` a dict name -> tensor over all ranks' trees in rank order, None on the other ranks."""
        if not self.active:
            return {name: record[name] for name, _ in RECORD_FIELDS if name in record}
        # Copies into the send buffer may be asynchronous only when the buffer lives on the records' device: the
        # collective is then ordered behind them on the current stream.  A host buffer (gloo) filled from device
        # records by a non-blocking copy could still hold the previous move's bytes when gloo reads it.
        for name, dtype in RECORD_FIELDS:
            x = record.get(name)
            if x is not None:
                dst_view = self._field(self.send, name, dtype)[:self.b]
                dst_view.copy_(x.reshape(self.b, self.widths[name]), non_blocking=x.device == dst_view.device)
        dist.gather(self.send, self.recv, dst=self.dst)
        if self.rank != self.dst:
            return None
        out = {}
        for name, dtype in RECORD_FIELDS:
            if name not in record:
                continue
            parts = [self._field(self.recv[r], name, dtype)[:self.sizes[r]] for r in range(self.world)]
            full = torch.cat(parts, dim=0).to(self.out_device)
            out[name] = full if record[name].dim() > 1 else full.reshape(-1)
        return out


def gather_root_stats(action: torch.Tensor, root_value: torch.Tensor, probs: torch.Tensor, dst: int = 0):
    """One-off form of StepGather for a search's root statistics: (action i32[b], root_value f64[b],
    probs f64[b, A]) of every rank on `dst`, in rank order; None on the other ranks.  Shards may be uneven."""
    if not dist.is_initialized():
        return action, root_value, probs
    g = StepGather(probs.shape[0], {'probs': probs.shape[1]}, device=probs.device, dst=dst)
    out = g.gather({'action': action, 'value': root_value, 'probs': probs})
    return None if out is None else (out['action'], out['value'], out['probs'])


def broadcast_weights(state_dict: dict, src: int = 0, device=None) -> dict:
    """The learner's weights to every search rank, as ONE broadcast of a flat float32 buffer (replaces the
    reference's per-actor `shared_storage.get_info('model_state_dict')` pull, self_play.py:37 /
    reanalyse.py:38-42).  Every rank passes a state dict of the same architecture (its current weights);
    the tensors are overwritten in place with `src`'s values and the dict is returned.  Non-float entries
    (BatchNorm's step counters) travel as float32 and are cast back."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return state_dict
    keys = sorted(state_dict)
    dev = device if device is not None else reduce_device(next(iter(state_dict.values())).device)
    flat = torch.cat([state_dict[k].detach().reshape(-1).to(torch.float32) for k in keys]).to(dev)
    dist.broadcast(flat, src=src)
    off = 0
    with torch.no_grad():
        for k in keys:
            t = state_dict[k]
            n = t.numel()
            t.copy_(flat[off:off + n].reshape(t.shape).to(device=t.device, dtype=t.dtype))
            off += n
    return state_dict
