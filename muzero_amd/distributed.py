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


def gather_root_stats(action: torch.Tensor, root_value: torch.Tensor, probs: torch.Tensor, dst: int = 0):
    """Gathers each rank's (action i32[b], root_value f64[b], probs f64[b, A]) on `dst`, in rank
    order (= global tree order for equal shards).  One packed all_gather: latency-bound, a few KB
    to a few hundred KB per step.  Returns the concatenated tensors on `dst`, None elsewhere."""
    if not dist.is_initialized():
        return action, root_value, probs
    b, A = probs.shape
    packed = torch.cat([action.to(torch.float64).reshape(b, 1), root_value.reshape(b, 1), probs], dim=1).contiguous()
    if dist.get_backend() != 'nccl':
        packed = packed.cpu()
    out = [torch.empty_like(packed) for _ in range(dist.get_world_size())]
    dist.all_gather(out, packed)
    if dist.get_rank() != dst:
        return None
    full = torch.cat(out, dim=0).to(probs.device)
    return full[:, 0].to(torch.int32), full[:, 1].contiguous(), full[:, 2:].contiguous()


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
