"""One device arena per training chain: what lets G chains share every launch of the equaliser step.

``dccn_eq_train_step_grouped`` (include/dccn.h, "chain groups") carries several independent equaliser chains -- the reference
driver's per-modulation / per-variant jobs, dev/py/run_local_ofdm.py:61-118 -- in one launch sequence: the chain index is a grid
dimension and a kernel reaches chain g's buffers by adding ONE byte offset to the pointers it was given for chain 0.  That works
when every chain keeps all of its device buffers in a single arena with the same internal layout.  :class:`ChainArena` is that
arena: a bump allocator the trainer, its fused plans, the fused generator and the epoch loop draw from when one is handed to
them; allocations happen in the same order for every chain, sizes that depend on the modulation are reserved for 16-QAM
(``reserve``), and :func:`check_same_layout` verifies the result (the C entry points verify it again, pointer by pointer).
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import numpy as np
import torch

ALIGN = 256


class ChainArena:
    def __init__(self, nbytes: int, device):
        self.device = torch.device(device)
        self.buf = torch.zeros(int(nbytes), dtype=torch.uint8, device=self.device)
        if self.buf.data_ptr() % ALIGN:
            raise RuntimeError("device allocation is not %d-byte aligned" % ALIGN)
        self.off = 0
        self.log: List[tuple] = []          # (offset, reserved bytes) per allocation, in order

    @property
    def base(self) -> int:
        return self.buf.data_ptr()

    def _take(self, nbytes: int, reserve_bytes: int) -> int:
        off = (self.off + ALIGN - 1) // ALIGN * ALIGN
        size = max(int(nbytes), int(reserve_bytes))
        if off + size > self.buf.numel():
            raise MemoryError("chain arena too small: %d + %d > %d bytes" % (off, size, self.buf.numel()))
        self.off = off + size
        self.log.append((off, size))
        return off

    def empty(self, *shape, dtype=torch.float32, reserve: Optional[int] = None) -> torch.Tensor:
        """a tensor inside the arena (contents: zeros the first time; never recycled).  ``reserve``: element count to set
        aside when the shape depends on the modulation (the layout must not)"""
        if len(shape) == 1 and isinstance(shape[0], (tuple, list, torch.Size)):
            shape = tuple(shape[0])
        n = int(np.prod(shape)) if len(shape) else 1
        item = torch.empty(0, dtype=dtype).element_size()
        off = self._take(n * item, (reserve or 0) * item)
        return self.buf[off:off + n * item].view(dtype).view(*shape)

    def zeros(self, *shape, dtype=torch.float32, reserve: Optional[int] = None) -> torch.Tensor:
        t = self.empty(*shape, dtype=dtype, reserve=reserve)
        t.zero_()
        return t

    def place(self, t, reserve: Optional[int] = None) -> torch.Tensor:
        """a copy of ``t`` (tensor or array) inside the arena"""
        t = torch.as_tensor(t)
        v = self.empty(tuple(t.shape), dtype=t.dtype, reserve=reserve)
        v.copy_(t)
        return v


def empty(arena: Optional[ChainArena], *shape, dtype=torch.float32, device=None, reserve: Optional[int] = None):
    return arena.empty(*shape, dtype=dtype, reserve=reserve) if arena is not None else torch.empty(*shape, dtype=dtype, device=device)


def zeros(arena: Optional[ChainArena], *shape, dtype=torch.float32, device=None, reserve: Optional[int] = None):
    return arena.zeros(*shape, dtype=dtype, reserve=reserve) if arena is not None else torch.zeros(*shape, dtype=dtype, device=device)


def place(arena: Optional[ChainArena], t, reserve: Optional[int] = None):
    return arena.place(t, reserve=reserve) if arena is not None else t


def check_same_layout(arenas: Sequence[ChainArena]):
    """every chain allocated the same (offset, size) sequence: the precondition of the grouped launches"""
    for a in arenas[1:]:
        if a.log != arenas[0].log:
            k = next((i for i, (x, y) in enumerate(zip(a.log, arenas[0].log)) if x != y), min(len(a.log), len(arenas[0].log)))
            raise RuntimeError("chain arenas differ at allocation %d: %s vs %s (%d / %d allocations)"
                               % (k, a.log[k:k + 1], arenas[0].log[k:k + 1], len(a.log), len(arenas[0].log)))
