"""Data-parallel gradient exchange for the block (one process per GPU, torch.distributed; backend "nccl" is RCCL over
xGMI on ROCm, "gloo" on CPU for tests).

The reference trains with DistributedDataParallel (U/engine/trainer.py:366-367): replicas, equal minibatch shards
(``batch // world_size``, trainer.py:379), one bucketed all-reduce(mean) of parameter gradients per step.  The block has no
cross-sample reduction, so its data path needs no collective; only its parameter gradients are exchanged:
47 KB (YOLOv8n) to 302 KB (l) per step.  At that size an all-reduce over xGMI is latency-bound, not bandwidth-bound, so
what matters is to hide its latency behind work that does not need the averaged gradients: ``GradExchange.start()``
launches the all-reduce of ONE flat bucket on a side stream, ``finish()`` joins it into the compute stream.  Callers put
independent work between the two -- bench.py overlaps it with the next step's parameter-free pooling kernel; a caller
that wants the overlap inside the step uses ``PyramidPlan.backward_params()`` (every parameter gradient complete),
``start()``, ``backward_inputs()`` (the big input-gradient kernel), ``finish()``.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.distributed as dist


def shard_batch(global_batch: int, world_size: int, rank: int) -> slice:
    """Equal contiguous shards, as DistributedSampler / trainer.py:379 (global batch must divide evenly)."""
    if global_batch % world_size:
        raise ValueError(f"global batch {global_batch} is not divisible by world size {world_size}")
    per = global_batch // world_size
    return slice(rank * per, (rank + 1) * per)


def payload_buckets(nbytes: int, device, cap_mb: float = 25.0):
    """fp32 buffers standing for the gradients of the layers OUTSIDE the hot path (backbone / neck / Detect: 11.9 MB for YOLOv8n ...
    130 MB for l), cut the way DistributedDataParallel cuts them: buckets of `cap_mb` (its default bucket_cap_mb = 25)."""
    n = max(0, int(nbytes) // 4)
    cap = max(1, int(cap_mb * (1 << 20)) // 4)
    return [torch.zeros(min(cap, n - o), dtype=torch.float32, device=device) for o in range(0, n, cap)]


class GradExchange:
    """All-reduce(mean) of flat gradient buckets, overlapped with whatever the caller enqueues between ``start()`` and
    ``finish()``.  One bucket (the blocks' own parameter gradients) or several (plus ``payload_buckets``: the rest of the model's
    gradients, which DDP puts through the same collective every step, U/engine/trainer.py:366-367): the buckets are reduced one
    after the other on ONE side stream, in order -- what DDP's reducer does with its ready buckets."""

    def __init__(self, bucket, group: Optional[dist.ProcessGroup] = None, single_rank_collective: bool = False):
        self.buckets = list(bucket) if isinstance(bucket, (list, tuple)) else [bucket]
        self.bucket = self.buckets[0]
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        # single_rank_collective: issue the collective even in a one-rank group (a no-op numerically).  For tests: it is the only way to run
        # the RCCL branch -- ReduceOp.AVG inside the collective, side stream, event join -- on a one-GPU box
        self.always = bool(single_rank_collective and dist.is_initialized())
        self.on_gpu = self.bucket.is_cuda
        self.side = torch.cuda.Stream(self.bucket.device) if self.on_gpu else None
        self._ready = torch.cuda.Event() if self.on_gpu else None
        self._work = None
        # RCCL averages inside the collective (one kernel fewer on the side stream than SUM + div_); gloo has no AVG
        self.avg_in_collective = bool(self.on_gpu and (self.world > 1 or self.always) and dist.get_backend(group) == "nccl")

    def nbytes(self) -> int:
        return sum(b.numel() * b.element_size() for b in self.buckets)

    def start(self):
        if self.world == 1 and not self.always:
            return
        if self.on_gpu:
            cur = torch.cuda.current_stream(self.bucket.device)
            self._ready.record(cur)                       # gradients complete at this point of the compute stream
            self.side.wait_event(self._ready)
            op = dist.ReduceOp.AVG if self.avg_in_collective else dist.ReduceOp.SUM
            with torch.cuda.stream(self.side):
                self._work = [dist.all_reduce(b, op=op, group=self.group, async_op=True) for b in self.buckets]
        else:
            self._work = [dist.all_reduce(b, op=dist.ReduceOp.SUM, group=self.group, async_op=True) for b in self.buckets]

    def finish(self):
        if (self.world == 1 and not self.always) or self._work is None:
            return
        if self.on_gpu:
            with torch.cuda.stream(self.side):
                for w in self._work:
                    w.wait()                              # orders the side stream after the collective
                if not self.avg_in_collective:
                    for b in self.buckets:
                        b.div_(self.world)                # DDP semantics: mean over replicas
            torch.cuda.current_stream(self.bucket.device).wait_stream(self.side)
        else:
            for w in self._work:
                w.wait()
            for b in self.buckets:
                b.div_(self.world)
        self._work = None
