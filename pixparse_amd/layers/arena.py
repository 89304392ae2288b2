"""Flat parameter arenas: one fp32 tensor for all parameters, one for all gradients (so the
optimiser tail and the gradient all-reduce see contiguous memory), one bf16 shadow of the
parameters (what every GEMM reads; refreshed by the fused AdamW kernel), plus AdamW m / v.

Entries are laid out in FORWARD order (encoder first, LM-head/embedding region before the decoder
layers), so during backward the gradient arena completes from its END towards its start -- the
bucketed reducer (framework/reducer.py) fires contiguous buckets as the backward sweep passes them.
"""
from collections import OrderedDict
from dataclasses import dataclass
from typing import Dict, Optional, Tuple

import torch

ALIGN = 64  # elements; keeps every entry 256-B aligned in fp32 and 128-B aligned in bf16


@dataclass
class Entry:
    name: str
    shape: Tuple[int, ...]
    offset: int
    numel: int
    alloc: int  # >= numel (padded rows, e.g. the vocab rounded up for the LM-head GEMM)


class ParamArena:
    def __init__(self):
        self.entries: "OrderedDict[str, Entry]" = OrderedDict()
        self.total = 0
        self.p: Optional[torch.Tensor] = None
        self.g: Optional[torch.Tensor] = None
        self.m: Optional[torch.Tensor] = None
        self.v: Optional[torch.Tensor] = None
        self.pb: Optional[torch.Tensor] = None

    # ---- layout
    def add(self, name: str, shape, alloc_numel: Optional[int] = None) -> None:
        assert self.p is None, 'arena already materialised'
        assert name not in self.entries, name
        numel = 1
        for s in shape:
            numel *= int(s)
        alloc = max(numel, alloc_numel or 0)
        self.entries[name] = Entry(name, tuple(int(s) for s in shape), self.total, numel, alloc)
        self.total += (alloc + ALIGN - 1) // ALIGN * ALIGN

    def materialize(self, device='cpu') -> None:
        self.p = torch.zeros(self.total, dtype=torch.float32, device=device)

    # ---- views
    def _view(self, flat: torch.Tensor, name: str, padded: bool = False) -> torch.Tensor:
        e = self.entries[name]
        if padded:
            return flat[e.offset:e.offset + e.alloc]
        return flat[e.offset:e.offset + e.numel].view(e.shape)

    def param(self, name, padded=False):
        return self._view(self.p, name, padded)

    def grad(self, name, padded=False):
        return self._view(self.g, name, padded)

    def shadow(self, name, padded=False):
        return self._view(self.pb, name, padded)

    def span(self, first: str, last: str) -> Tuple[int, int]:
        """[start, end) element range covering entries first..last (inclusive, layout order)."""
        a, b = self.entries[first], self.entries[last]
        return a.offset, b.offset + (b.alloc + ALIGN - 1) // ALIGN * ALIGN

    # ---- device / training state
    def to(self, device) -> None:
        for k in ('p', 'g', 'm', 'v', 'pb'):
            t = getattr(self, k)
            if t is not None:
                setattr(self, k, t.to(device))

    def apply_(self, fn) -> None:
        for k in ('p', 'g', 'm', 'v'):
            t = getattr(self, k)
            if t is not None:
                setattr(self, k, fn(t))
        if self.pb is not None:
            self.pb = self.pb.to(self.p.device)

    def alloc_training_state(self) -> None:
        dev = self.p.device
        if self.g is None:
            self.g = torch.zeros(self.total, dtype=torch.float32, device=dev)
        if self.m is None:
            self.m = torch.zeros(self.total, dtype=torch.float32, device=dev)
            self.v = torch.zeros(self.total, dtype=torch.float32, device=dev)

    def alloc_shadow(self) -> None:
        if self.pb is None or self.pb.device != self.p.device:
            self.pb = torch.empty(self.total, dtype=torch.bfloat16, device=self.p.device)

    def named_shapes(self) -> Dict[str, Tuple[int, ...]]:
        return {k: e.shape for k, e in self.entries.items()}
