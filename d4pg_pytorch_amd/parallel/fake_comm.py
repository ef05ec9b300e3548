"""In-process stand-in for a process group: ``world`` threads, barrier-synchronised
sum all-reduce and broadcast on shared tensors.  Same interface as
``TorchDistComm`` so data-parallel logic can be tested without spawning
processes or touching a network backend."""
from __future__ import annotations

import threading
from typing import List

import torch

from .comm import Comm


class _Shared:
    def __init__(self, world: int):
        self.world = world
        self.barrier = threading.Barrier(world)
        self.slots: List = [None] * world
        self.lock = threading.Lock()


class ThreadComm(Comm):
    def __init__(self, shared: _Shared, rank: int):
        self.shared, self.rank, self.world = shared, rank, shared.world

    @staticmethod
    def group(world: int) -> List["ThreadComm"]:
        sh = _Shared(world)
        return [ThreadComm(sh, r) for r in range(world)]

    def all_reduce_sum_(self, t: torch.Tensor) -> torch.Tensor:
        sh = self.shared
        sh.slots[self.rank] = t.detach().clone()
        sh.barrier.wait()
        total = torch.stack([sh.slots[r] for r in range(self.world)]).sum(0)   # same order on every rank
        sh.barrier.wait()
        with torch.no_grad():
            t.copy_(total)
        return t

    def broadcast_(self, t: torch.Tensor, src: int = 0) -> torch.Tensor:
        sh = self.shared
        if self.rank == src:
            sh.slots[src] = t.detach().clone()
        sh.barrier.wait()
        with torch.no_grad():
            t.copy_(sh.slots[src])
        sh.barrier.wait()
        return t

    def all_gather_into_(self, out: torch.Tensor, inp: torch.Tensor) -> torch.Tensor:
        sh = self.shared
        if inp.is_cuda:
            torch.cuda.current_stream(inp.device).synchronize()     # stream-ordered producers of inp
        sh.slots[self.rank] = inp.detach().reshape(-1).clone()
        if inp.is_cuda:
            torch.cuda.current_stream(inp.device).synchronize()
        sh.barrier.wait()
        with torch.no_grad():
            out.view(-1).copy_(torch.cat([sh.slots[r] for r in range(self.world)]))
        if out.is_cuda:
            torch.cuda.current_stream(out.device).synchronize()
        sh.barrier.wait()
        return out

    def barrier(self) -> None:
        self.shared.barrier.wait()

    def all_reduce_max_scalar(self, v: float, device="cpu") -> float:
        sh = self.shared
        sh.slots[self.rank] = float(v)
        sh.barrier.wait()
        out = max(sh.slots[: self.world])
        sh.barrier.wait()
        return out
