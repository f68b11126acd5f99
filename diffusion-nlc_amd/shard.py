"""Sample-level sharding of the sampling run over the GPUs of one node.

Independent samples never interact on the unconstrained path, so the run shards by WHOLE
reference-sized batches: rank r owns global batches r, r+W, r+2W, ...  Every rank replays the
reference's single host generator in the reference's order (one ``randn(batch_shape)`` per global
batch, image_sample.py:529 / src/experiments.py:268) and keeps only its own draws, so the
gathered result equals the single-process run bit for bit.  The only collective is one
all-gather of the finished samples (RCCL over xGMI with backend 'nccl'; 'gloo' in the CPU tests).
"""
from __future__ import annotations

import os
from typing import Callable, List, Optional, Sequence

import torch
import torch.distributed as dist


def owned_batches(n_batches: int, world: int, rank: int) -> List[int]:
    return list(range(rank, n_batches, world))


def draw_initial_noise(batch_shape: Sequence[int], n_batches: int, seed: int, world: int, rank: int) -> List[torch.Tensor]:
    """Replay the reference's generator for ALL global batches, keep this rank's (host tensors)."""
    gen = torch.Generator().manual_seed(seed)
    mine = set(owned_batches(n_batches, world, rank))
    out = []
    for j in range(n_batches):
        z = torch.randn(tuple(batch_shape), generator=gen)
        if j in mine:
            out.append(z)
    return out


def init_from_env(backend: Optional[str] = None) -> tuple:
    """(rank, world, local_rank) from torchrun's environment; initialises torch.distributed when world > 1."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if backend == "nccl":
            kw["device_id"] = torch.device("cuda", local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, world, local


def world_rank() -> tuple:
    """(world, rank) of the running job: the initialised process group if there is one, else a single process."""
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(), dist.get_rank()
    return 1, 0


def broadcast_object(obj, src: int = 0):
    """Small host object from rank ``src`` to every rank (directory names, the skip-if-exists decisions)."""
    world, _ = world_rank()
    if world == 1:
        return obj
    box = [obj]
    dist.broadcast_object_list(box, src=src)
    return box[0]


def barrier() -> None:
    if world_rank()[0] > 1:
        dist.barrier()


def replay_draws(gen: torch.Generator, shape: Sequence[int], n: int) -> None:
    """Advance the reference's single host generator past a batch another rank samples: ``n`` x randn(shape), discarded."""
    for _ in range(n):
        torch.randn(tuple(shape), generator=gen)


# ---- the exchange half of the sigma-net training (src/experiments.py:645-652) ------------------------------------------
# The reference wraps the sigma net in DistributedDataParallel(bucket_cap_mb=128, broadcast_buffers=False): constructing it
# broadcasts rank 0's parameters to every rank, and a backward pass outside `no_sync()` averages the gradients in 128 MB buckets.
# (Its training loop happens to run every forward under `no_sync()` (:682-686), so upstream only the initial broadcast ever
# crosses GPUs.)  Both collectives, on flat buckets, over whatever backend the job runs (RCCL over xGMI: ring all-reduce is
# per-link bound at ~153 GB/s, so few large buckets - the 61.4 M-parameter ADM sigma net is two of them).
def _buckets(tensors, bucket_bytes: int):
    cur, size = [], 0
    for t in tensors:
        n = t.numel() * t.element_size()
        if cur and (size + n > bucket_bytes or t.dtype != cur[0].dtype or t.device != cur[0].device):
            yield cur
            cur, size = [], 0
        cur.append(t)
        size += n
    if cur:
        yield cur


def broadcast_parameters_(tensors, src: int = 0, bucket_bytes: int = 128 << 20) -> None:
    """In place: every rank's ``tensors`` become rank ``src``'s (DDP's construction-time parameter sync), bucket by bucket."""
    world, _ = world_rank()
    if world == 1:
        return
    for b in _buckets(list(tensors), bucket_bytes):
        flat = torch.cat([t.reshape(-1) for t in b])
        dist.broadcast(flat, src=src)
        off = 0
        for t in b:
            t.copy_(flat[off:off + t.numel()].view_as(t))
            off += t.numel()


def allreduce_mean_(tensors, bucket_bytes: int = 128 << 20) -> None:
    """In place: every rank's ``tensors`` (gradients) become the mean over ranks (what DDP's backward hook does), bucket by
    bucket, one all-reduce each."""
    world, _ = world_rank()
    if world == 1:
        return
    for b in _buckets(list(tensors), bucket_bytes):
        flat = torch.cat([t.reshape(-1) for t in b])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        flat /= world
        off = 0
        for t in b:
            t.copy_(flat[off:off + t.numel()].view_as(t))
            off += t.numel()


def gather_samples(local: torch.Tensor, n_batches: int, world: int, rank: int) -> torch.Tensor:
    """All-gather per-rank results [n_local, B, ...] and restore global batch order -> [n_batches, B, ...].

    Ranks may own different numbers of batches when world does not divide n_batches; shorter ranks are
    padded to the maximum for the collective and the padding is dropped afterwards."""
    if world == 1:
        return local
    n_max = (n_batches + world - 1) // world
    pad = n_max - local.shape[0]
    if pad:
        local = torch.cat([local, local.new_zeros((pad,) + tuple(local.shape[1:]))], 0)
    gathered = local.new_empty((world * n_max,) + tuple(local.shape[1:]))
    dist.all_gather_into_tensor(gathered.view(-1), local.contiguous().view(-1))
    # global batch j was sampled by rank j % world as its (j // world)-th batch: one gather-by-index on the device
    j = torch.arange(n_batches, device=local.device)
    return gathered.index_select(0, (j % world) * n_max + j // world)


def sample_sharded(run_batch: Callable[[torch.Tensor], torch.Tensor], batch_shape: Sequence[int], n_batches: int,
                   seed: int, world: int, rank: int, device) -> torch.Tensor:
    """Run ``run_batch(z) -> x`` on this rank's batches and all-gather: returns [n_batches, B, ...] on ``device``."""
    zs = draw_initial_noise(batch_shape, n_batches, seed, world, rank)
    outs = [run_batch(z).to(device) for z in zs]
    local = torch.stack(outs) if outs else torch.empty((0,) + tuple(batch_shape), device=device)
    return gather_samples(local, n_batches, world, rank)
