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
fraction.  On top of that the persistent GEMMs launch on 256 - n CUs from the first bucket of a backward sweep until finish() (a whole
number of tile rounds on the CUs that are left: measured on one GPU with 16 / 32 CUs taken, +6.9 / +7.6 % per step against +13 / +15 %
without the reservation, profiles/r3_cu_contention.txt).  n = PIXPARSE_AMD_RCCL_CUS when set (0 switches the reservation off); otherwise,
whenever world_size > 1, NCCL_MAX_NCHANNELS when that is set (RCCL runs one workgroup = one CU per channel), else 16.

Diagnosis of a multi-GPU run (bench.py puts it on its JSON line): `stats()` returns the bucket geometry, the CUs reserved and
`comm_exposed_ms` -- per optimiser step, the time the compute stream spends in finish() waiting for collectives that have not
completed when backward is done (event before the first wait -> event after the last one; 0 when everything overlapped).
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
        self.bucket_bytes = int(bucket_bytes)
        self.reserved_cus = self._auto_reserved_cus() if arena.p.is_cuda else 0
        self._reserved_on = False
        self._exposed = []             # (event before the first wait, event after the last) of every finish() since reset_stats()
        self._finishes = 0

    def _auto_reserved_cus(self) -> int:
        env = os.environ.get('PIXPARSE_AMD_RCCL_CUS')
        if env not in (None, ''):
            return max(0, int(env))
        if not (self.active and self.world_size > 1):
            return 0
        ch = os.environ.get('NCCL_MAX_NCHANNELS')
        n = int(ch) if ch and ch.isdigit() and int(ch) > 0 else 16
        return min(n, 64)

    def reset_stats(self):
        self._exposed = []
        self._finishes = 0

    def stats(self) -> dict:
        """call after a device synchronisation (reads event timings)"""
        ms = [a.elapsed_time(b) for a, b in self._exposed]
        return {'buckets': len(self.buckets), 'bucket_bytes': self.bucket_bytes, 'reserved_cus': self.reserved_cus, 'reductions': self._finishes,
                'comm_exposed_ms': round(sum(ms) / max(1, len(ms)), 3) if ms else 0.0,
                'comm_exposed_ms_max': round(max(ms), 3) if ms else 0.0}

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
        timed = self.active and self.enabled and self._works and self.arena.g.is_cuda and len(self._exposed) < 4096
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        for w in self._works:
            w.wait()
        if timed:
            e1.record()
            self._exposed.append((e0, e1))
        if self._works:
            self._finishes += 1
        self._works = []
        self._reserve(False)

    def grad_divisor(self) -> float:
        return float(self.world_size) if (self.enabled and self.world_size > 1) else 1.0
