"""Bucketed gradient all-reduce over the flat gradient arena (replaces DistributedDataParallel's
reducer, ref: task/task_cruller_pretrain.py:181-189,280-283).

The arena is laid out in forward order, so backward completes it from the end towards the start.
Buckets are contiguous slices cut from the END of the arena; ``on_ready(name)`` (called by the
backward sweep) launches the asynchronous all-reduce of every bucket that lies entirely at or after
entry ``name``.  With the RCCL backend each all-reduce runs on the process group's own stream and
is ordered after the kernels already enqueued on the compute stream, so communication overlaps the
rest of backward; ``finish()`` makes the compute stream wait for all of them.  The sum is NOT
divided here: 1/world_size is folded into the unscale/clip coefficient of the optimiser kernel.
xGMI is a full mesh of point-to-point links, so buckets are large (default 64 MiB) -- few, big
collectives -- rather than DDP's 25 MiB.

The collectives are issued through torch.distributed (backend "nccl" = RCCL), as the reference's DDP does: stream, channel count and
event chaining are ProcessGroupNCCL's.  RCCL's kernels hold CUs while a bucket is in flight; the persistent GEMMs cope by pulling
their tiles from ticket counters (include/crl.h crl_gemm_set_schedule), so a launch that finds CUs taken slows down by about the CU
fraction.  PIXPARSE_AMD_RCCL_CUS=n additionally makes them launch on 256 - n CUs from the first bucket of a backward sweep until
finish() (a whole number of tile rounds on the CUs that are left; worth it only when the collectives are long -- default 0, unset).
"""
import os
from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


class BucketedGradReducer:
    def __init__(self, arena, world_size: int, bucket_bytes: int = 64 << 20, group=None, active: Optional[bool] = None,
                 op=dist.ReduceOp.SUM):
        """active: run the collectives (default: world_size > 1). A single rank started by torchrun passes True so that
        the N = 1 run exercises the same RCCL calls as N = 8. `op` is SUM in production (tests pass a pre-multiplied
        sum to make the collective observable on one rank)."""
        self.arena, self.world_size, self.group, self.op = arena, world_size, group, op
        self.active = (world_size > 1) if active is None else bool(active)
        n = max(1, bucket_bytes // 4)
        total = arena.total
        self.buckets: List[Tuple[int, int]] = []
        end = total
        while end > 0:
            start = max(0, end - n)
            self.buckets.append((start, end))
            end = start
        self.enabled = True
        self._next = 0
        self._works = []
        self.reserved_cus = int(os.environ.get('PIXPARSE_AMD_RCCL_CUS', '0') or 0) if arena.p.is_cuda else 0
        self._reserved_on = False

    def _reserve(self, on: bool):
        if self.reserved_cus and on != self._reserved_on:
            from .. import ops
            ops.gemm_set_reserved_cus(self.reserved_cus if on else 0)
            self._reserved_on = on

    def broadcast_params(self, src: int = 0):
        """DDP constructor semantics (C1): every rank starts from rank 0's parameters."""
        if self.active:
            dist.broadcast(self.arena.p, src=src, group=self.group)

    def begin(self):
        self._next = 0
        self._works = []

    def _fire_until(self, offset: int):
        if not self.enabled or not self.active:
            return
        g = self.arena.g
        while self._next < len(self.buckets) and self.buckets[self._next][0] >= offset:
            s, e = self.buckets[self._next]
            self._reserve(True)
            self._works.append(dist.all_reduce(g[s:e], op=self.op, group=self.group, async_op=True))
            self._next += 1

    def on_ready(self, name: str):
        self._fire_until(self.arena.entries[name].offset)

    def finish(self):
        self._fire_until(0)
        for w in self._works:
            w.wait()
        self._works = []
        self._reserve(False)

    def grad_divisor(self) -> float:
        return float(self.world_size) if (self.enabled and self.world_size > 1) else 1.0
