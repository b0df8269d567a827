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


def world_size():
    """Ranks of the default process group (1 when torch.distributed is not initialised)."""
    import torch.distributed as dist
    return dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1


def init_from_env(backend=None, timeout_s=None):
    """Initialise the default process group from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torchrun).
    ``timeout_s`` (default SEC_DIST_TIMEOUT_S, else 4 h): collective timeout -- ranks != 0 sit in the gradient all-reduce
    while rank 0 evaluates and checkpoints (launch.py seam 5), which can outlast the 10-minute RCCL default."""
    import datetime
    if timeout_s is None:
        timeout_s = float(os.environ.get("SEC_DIST_TIMEOUT_S", 4 * 3600))
    timeout = datetime.timedelta(seconds=timeout_s)
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
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device("cuda", local_rank),
                                    timeout=timeout)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world, timeout=timeout)
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


class GradBucket:
    """ONE flat fp32 bucket holding the gradient of every trainable parameter, in parameter order, on every rank.

    The bucket is laid out from ``requires_grad`` (identical on all ranks), never from which ``.grad`` happen to exist: a
    parameter without a gradient on this rank this step (an empty shard, an unused head) contributes zeros, so the ranks
    always reduce buffers of the same length and layout.  After :meth:`allreduce` every fp32 ``p.grad`` is a view of the
    bucket -- later steps accumulate straight into it, so the flatten / unflatten copies disappear.

    Parameters that have no gradient on ANY rank: with ``track_presence=False`` (the sync-free device trainer) they end up
    with an all-zero gradient, so an optimizer with weight decay / moments still touches them; with ``track_presence=True``
    (the launcher around the reference's loop, whose optimizer skips ``grad is None`` parameters) one flag per parameter
    rides at the end of the same all-reduce and those parameters get ``grad = None`` back -- at the price of one small
    device-to-host read per step."""

    def __init__(self, module, track_presence=False):
        self.params = [p for p in module.parameters() if p.requires_grad]
        self.numel = sum(p.numel() for p in self.params)
        self.track_presence = bool(track_presence)
        dev = self.params[0].device if self.params else torch.device("cpu")
        self._buf = torch.zeros(self.numel + (len(self.params) if self.track_presence else 0), dtype=torch.float32, device=dev)
        self.flat = self._buf[:self.numel]
        self.flags = self._buf[self.numel:]
        self.views = []
        off = 0
        for p in self.params:
            self.views.append(self.flat[off:off + p.numel()].view_as(p))
            off += p.numel()

    def pack(self):
        present, srcs, dsts = [], [], []
        for p, v in zip(self.params, self.views):
            present.append(0.0 if p.grad is None else 1.0)
            if p.grad is None:
                v.zero_()
            elif p.grad.data_ptr() != v.data_ptr():
                if p.grad.dtype == v.dtype and p.grad.device == v.device:
                    srcs.append(p.grad)
                    dsts.append(v)
                else:
                    v.copy_(p.grad)
        if dsts:
            torch._foreach_copy_(dsts, srcs)        # a few multi-tensor launches instead of one copy per parameter
        if self.track_presence and present:
            self.flags.copy_(torch.tensor(present, dtype=torch.float32))
        return self._buf

    def unpack(self):
        absent = set()
        if self.track_presence and len(self.params):
            absent = {i for i, f in enumerate(self.flags.tolist()) if f == 0.0}
        for i, (p, v) in enumerate(zip(self.params, self.views)):
            if i in absent:
                p.grad = None                 # no rank produced a gradient: the optimizer skips it, as in a single process
            elif p.dtype == torch.float32:
                p.grad = v                    # a view of the bucket: the next backward accumulates into it in place
            else:
                if p.grad is None:
                    p.grad = torch.empty_like(p)
                p.grad.copy_(v)

    def zero_grad(self, set_to_none=True):
        """After the optimizer step.  ``set_to_none`` (default): every ``p.grad`` is dropped -- the next backward then HANDS its
        gradient tensors over (no kernel) instead of adding them into the bucket views one parameter at a time (~70 small
        launches per step on the SECOND networks), and :meth:`pack` gathers them with a few multi-tensor copies.
        ``set_to_none=False``: one fill for the bucket (= every fp32 ``p.grad`` stays a zeroed view that backward accumulates
        into); gradients of non-fp32 parameters are separate tensors and are dropped either way, otherwise the next
        :meth:`pack` would add the stale (already averaged) values."""
        if set_to_none:
            for p in self.params:
                p.grad = None
            return
        self._buf.zero_()
        for p, v in zip(self.params, self.views):
            if p.grad is not None and p.grad.data_ptr() != v.data_ptr():
                p.grad = None

    def reduce(self, average=True):
        """The collective alone, on the packed bucket (between :meth:`pack` and :meth:`unpack`; a step captured as two graphs
        issues exactly this between their replays)."""
        if dist.is_initialized() and dist.get_world_size() > 1:
            dist.all_reduce(self._buf, op=dist.ReduceOp.SUM)
            if average:
                self.flat /= dist.get_world_size()

    def allreduce(self, average=True):
        self.pack()
        self.reduce(average)
        self.unpack()
        return self.numel * 4


def allreduce_gradients(module, average=True):
    """Average the gradients of all ranks with ONE all-reduce over a single flat bucket (see :class:`GradBucket` for the
    layout rule).  Call after backward(), before clip_grad_norm_/optimizer.step (cf. train.py:318-325)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return 0
    bucket = getattr(module, "_sec_grad_bucket", None)
    trainable = [p for p in module.parameters() if p.requires_grad]
    if bucket is None or len(bucket.params) != len(trainable) or any(a is not b for a, b in zip(bucket.params, trainable)) or \
            (trainable and bucket.flat.device != trainable[0].device):
        bucket = GradBucket(module, track_presence=True)   # the caller's optimizer skips ``grad is None`` parameters
        module._sec_grad_bucket = bucket
    if bucket.numel == 0:
        return 0
    return bucket.allreduce(average)


def max_over_ranks(value, device=None):
    """max of a python float over ranks (bench timing)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
