"""Multi-GPU layer: one process per GPU, torch.distributed (backend "nccl" == RCCL over xGMI on ROCm).

The reference's only multi-GPU mechanism is single-process nn.DataParallel with padded inputs
(second/pytorch/train.py:203-206, second/data/preprocess.py:57-88).  MI355X-first replacement:
  * inference  -- frames are independent (batch index = column 0 of the indices): shard frames across ranks,
                  NO data-path collective (`shard_frames`);
  * training   -- each rank runs the reference's single-GPU path on its shard of the batch; gradients
                  (1.83 M parameters = 7.3 MB fp32 for car.fhd) are flattened into ONE bucket and averaged
                  with ONE all-reduce per step (`allreduce_gradients`): on a 7-link xGMI ring a 7.3 MB
                  bucket is latency-bound (~0.1 ms), so more buckets would only add launches.  BatchNorm
                  statistics stay per rank, as in the reference (DataParallel replicas never sync BN).
All helpers work with any backend (tests use gloo on CPU, world size 2).
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise the default process group from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torchrun)."""
    world = int(os.environ.get("WORLD_SIZE", 1))
    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, local_rank, world


def shard_frames(num_frames, rank, world):
    """Contiguous, balanced frame range [lo, hi) of this rank (per-frame data parallel inference)."""
    base, rem = divmod(num_frames, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def broadcast_parameters(module, src=0):
    """Make every rank start from rank `src`'s parameters and buffers (one flat broadcast)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return
    tensors = [p.data for p in module.parameters()] + [b.data for b in module.buffers() if b.is_floating_point()]
    if not tensors:
        return
    flat = torch.cat([t.reshape(-1).float() for t in tensors])
    dist.broadcast(flat, src)
    off = 0
    for t in tensors:
        n = t.numel()
        t.copy_(flat[off:off + n].view_as(t).to(t.dtype))
        off += n


def allreduce_gradients(module, average=True):
    """Average the gradients of all ranks with ONE all-reduce over a single flat bucket.
    Call after backward(), before clip_grad_norm_/optimizer.step (cf. train.py:318-325)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return 0
    grads = [p.grad for p in module.parameters() if p.grad is not None]
    if not grads:
        return 0
    flat = torch.cat([g.reshape(-1).float() for g in grads])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    if average:
        flat /= dist.get_world_size()
    off = 0
    for g in grads:
        n = g.numel()
        g.copy_(flat[off:off + n].view_as(g).to(g.dtype))
        off += n
    return flat.numel() * 4


def max_over_ranks(value, device=None):
    """max of a python float over ranks (bench timing)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
