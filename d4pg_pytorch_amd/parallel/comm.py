"""Collective communication for data-parallel learners.

The reference has no collectives at all (one learner process, one device).
This framework scales the learner as one process per GPU; the ranks exchange

* the flat critic-gradient range   (all-reduce, before the critic Adam step),
* the flat actor-gradient range    (all-reduce, before the actor Adam step),
* the initial parameters           (broadcast from rank 0),
* a tiny control vector            (all-reduce: stop flag / step agreement),

all as single contiguous fp32 buffers (0.7-0.9 MB): on xGMI these messages are
latency-bound, so one flat bucket per network is the right granularity.
That is ``dp_mode: grad_allreduce`` (:func:`attach_data_parallel`).  The default
``dp_mode: replicated`` (``replicated.py``) exchanges the sampled rows instead
(one all-gather of ~20 KB per rank and update, overlapped with the update in flight).

:class:`TorchDistComm` wraps ``torch.distributed`` (backend ``nccl`` == RCCL on
ROCm, ``gloo`` on CPU).  :class:`LocalComm` is the world-size-1 no-op.
:class:`ThreadComm` (``fake_comm.py``) is an in-process stand-in for tests.
"""
from __future__ import annotations

import contextlib
import os
import sys
from typing import Optional

import torch


class Comm:
    rank: int = 0
    world: int = 1

    def all_reduce_sum_(self, t: torch.Tensor) -> torch.Tensor:
        raise NotImplementedError

    def broadcast_(self, t: torch.Tensor, src: int = 0) -> torch.Tensor:
        raise NotImplementedError

    def all_gather_into_(self, out: torch.Tensor, inp: torch.Tensor) -> torch.Tensor:
        """``out[rank * n:(rank + 1) * n] = inp`` of every rank (``n = inp.numel()``)."""
        raise NotImplementedError

    def barrier(self) -> None:
        raise NotImplementedError

    def all_reduce_max_scalar(self, v: float, device="cpu") -> float:
        raise NotImplementedError

    def close(self) -> None:
        pass


class LocalComm(Comm):
    def all_reduce_sum_(self, t):
        return t

    def broadcast_(self, t, src=0):
        return t

    def all_gather_into_(self, out, inp):
        if out.data_ptr() != inp.data_ptr():
            out.view(-1)[: inp.numel()].copy_(inp.reshape(-1))
        return out

    def barrier(self):
        pass

    def all_reduce_max_scalar(self, v, device="cpu"):
        return float(v)


@contextlib.contextmanager
def _stdout_to_stderr():
    """File-descriptor level redirect: RCCL prints a version banner to stdout when it creates its
    first communicator, which would corrupt the output of programs that print machine-readable
    results there."""
    sys.stdout.flush()
    saved = os.dup(1)
    try:
        os.dup2(2, 1)
        yield
    finally:
        sys.stdout.flush()
        os.dup2(saved, 1)
        os.close(saved)


class TorchDistComm(Comm):
    """``torch.distributed`` process group: RCCL over xGMI on GPUs, gloo on CPU."""

    def __init__(self, backend: Optional[str] = None, rank: Optional[int] = None, world: Optional[int] = None,
                 device: Optional[torch.device] = None, init_method: Optional[str] = None,
                 timeout_s: float = 600.0):
        import datetime
        import torch.distributed as dist
        self.dist = dist
        self.device = torch.device(device) if device is not None else torch.device("cpu")
        if backend is None:
            # D4PG_DIST_BACKEND=gloo: rehearse a multi-rank job whose ranks share one GPU (RCCL refuses
            # two ranks on a device); collectives on GPU tensors are then staged through the host
            backend = os.environ.get("D4PG_DIST_BACKEND") or ("nccl" if self.device.type == "cuda" else "gloo")
        self.backend = backend
        self._owns = False
        if not dist.is_initialized():
            kwargs = {}
            if rank is not None:
                kwargs.update(rank=rank, world_size=world)
            if init_method is None:
                os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
                os.environ.setdefault("MASTER_PORT", "29531")
                init_method = "env://"
            if backend == "nccl":
                kwargs["device_id"] = self.device
            with _stdout_to_stderr():
                dist.init_process_group(backend=backend, init_method=init_method,
                                        timeout=datetime.timedelta(seconds=timeout_s), **kwargs)
                # create the communicator now (it is otherwise created by the first collective)
                warm = torch.zeros(1, device=self.device if backend == "nccl" else "cpu")
                dist.all_reduce(warm)
                if warm.is_cuda:
                    torch.cuda.synchronize(self.device)
            self._owns = True
        self.rank = dist.get_rank()
        self.world = dist.get_world_size()

    def _staged(self, t: torch.Tensor) -> bool:
        return self.backend != "nccl" and t.is_cuda

    def all_reduce_sum_(self, t):
        if self._staged(t):
            h = t.detach().cpu()                 # synchronises the current stream
            self.dist.all_reduce(h, op=self.dist.ReduceOp.SUM)
            with torch.no_grad():
                t.copy_(h)
            return t
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return t

    def broadcast_(self, t, src=0):
        if self._staged(t):
            h = t.detach().cpu()
            self.dist.broadcast(h, src=src)
            with torch.no_grad():
                t.copy_(h)
            return t
        self.dist.broadcast(t, src=src)
        return t

    def all_gather_into_(self, out, inp):
        if self._staged(inp):
            h = inp.detach().reshape(-1).cpu()
            parts = [torch.empty_like(h) for _ in range(self.world)]
            self.dist.all_gather(parts, h)
            with torch.no_grad():
                out.view(-1).copy_(torch.cat(parts))
            return out
        self.dist.all_gather_into_tensor(out.view(-1), inp.reshape(-1))
        return out

    def barrier(self):
        if self.backend == "nccl":
            self.dist.barrier(device_ids=[self.device.index or 0])
        else:
            self.dist.barrier()

    def all_reduce_max_scalar(self, v, device=None):
        t = torch.tensor([float(v)], dtype=torch.float64, device=self.device if self.backend == "nccl" else "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def close(self):
        if self._owns and self.dist.is_initialized():
            self.dist.destroy_process_group()
            self._owns = False


#: One GPU, us per update by batch size (Walker2d shape, ``profiles/README.md``): what a rank of the replicated mode pays
#: for the GLOBAL batch, and a rank of the gradient-exchange mode for its LOCAL batch.
UPDATE_US = {"fp32": {256: 78.0, 512: 86.0, 768: 99.0, 1024: 107.0, 2048: 144.0, 2064: 175.0, 3072: 186.0, 4096: 202.0, 8192: 358.0},
             "bf16": {256: 52.0, 512: 55.0, 1024: 59.0, 2048: 64.0, 4096: 90.0, 8192: 131.0}}
#: what the gradient-exchange mode costs on top of the update and its two all-reduces (two more launches that apply the
#: summed gradients, gradients through memory): measured over a world of one, where an in-graph RCCL all-reduce is no
#: kernel at all (0.2 us in the probe): 85.4 - 77.5 us (fp32), 59.0 - 52.2 us (bf16) (bench.py --dp-selftest)
EXCHANGE_OVERHEAD_US = {"fp32": 8.0, "bf16": 7.0}
#: Without a live communicator to time (``resolve_dp_mode`` called with ``allreduce_us=None``): the largest global batch that is
#: replicated, by precision; an all-reduce of a 0.8 MB gradient bucket is then taken to cost 35 us.
REPLICATED_MAX_GLOBAL_BATCH = {"fp32": 2048, "bf16": 8192}


def update_us(batch: int, compute_dtype: str = "fp32") -> float:
    """Interpolated single-GPU update time at ``batch`` rows (linear between the measured sizes, linear in the batch
    beyond the largest)."""
    tab = UPDATE_US["bf16" if compute_dtype == "bf16" else "fp32"]
    xs = sorted(tab)
    if batch <= xs[0]:
        return tab[xs[0]]
    for lo, hi in zip(xs, xs[1:]):
        if batch <= hi:
            return tab[lo] + (tab[hi] - tab[lo]) * (batch - lo) / (hi - lo)
    return tab[xs[-1]] * batch / xs[-1]


def probe_allreduce_us(comm: "Comm", numel: int, device, iters: int = 20, in_graph: bool = False) -> float:
    """Time of one sum all-reduce of ``numel`` floats on the LIVE communicator, the maximum over the ranks so that
    every rank decides from the same number.  This is what replaces a guessed xGMI latency in ``dp_mode: auto``.

    Default: from the host, the way every other collective of the job is issued — median of ``iters`` (barrier,
    all-reduce, synchronise), MINUS the median of the same sequence around a one-element device operation, i.e. minus
    what launching and waiting costs by itself (15 - 20 us): what is left is what the collective adds to a stream.
    ``in_graph=True`` (RCCL only): ``iters`` dependent all-reduces captured as one graph and replayed, device time
    between two events — the way the gradient exchange runs them; 0.2 us at a world of one, where RCCL launches nothing.
    (Not the default: a communicator that cannot run captured collectives would hang the probe, and the probe runs
    before anything else has been measured; ``bench.py`` records it later, under its watchdog.)
    """
    import time
    buf = torch.zeros(int(numel), dtype=torch.float32, device=device)
    cuda = torch.device(device).type == "cuda"
    for _ in range(3):
        comm.all_reduce_sum_(buf)
    if cuda:
        torch.cuda.synchronize(device)
    us = None
    if in_graph and cuda and isinstance(comm, TorchDistComm) and comm.backend == "nccl":
        try:
            us = _graph_allreduce_us(comm, buf, device, iters)
        except Exception:      # capture refused (an RCCL build without graph support): time from the host instead
            us = None
            torch.cuda.synchronize(device)
    if us is None:
        one = torch.zeros(1, dtype=torch.float32, device=device)

        def median_us(op) -> float:
            times = []
            for _ in range(iters):
                comm.barrier()
                t0 = time.perf_counter()
                op()
                if cuda:
                    torch.cuda.synchronize(device)
                times.append(time.perf_counter() - t0)
            times.sort()
            return times[len(times) // 2] * 1e6
        total = median_us(lambda: comm.all_reduce_sum_(buf))
        launch = median_us(lambda: one.add_(1.0)) if cuda else 0.0
        us = max(0.0, total - launch)
    return comm.all_reduce_max_scalar(us)


def _graph_allreduce_us(comm: "Comm", buf: torch.Tensor, device, iters: int) -> float:
    stream = torch.cuda.Stream(device=device)
    stream.wait_stream(torch.cuda.current_stream(device))
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(stream):
        with torch.cuda.graph(g, stream=stream, capture_error_mode="thread_local"):
            for _ in range(iters):
                comm.all_reduce_sum_(buf)
        g.replay()                                  # the first replay pays the upload of the executable graph
        stream.synchronize()
        comm.barrier()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        g.replay()
        e1.record(stream)
        stream.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def resolve_dp_mode(mode: str, local_batch: int, world: int, compute_dtype: str = "fp32",
                    allreduce_us: Optional[float] = None) -> str:
    """``auto`` -> the cheaper of the two modes per update.  ``replicated`` pays the single-GPU update time of the
    GLOBAL batch; ``grad_allreduce`` that of the local batch, two all-reduces of a gradient bucket on the critical
    path (``allreduce_us`` each: measured on the live communicator by :func:`probe_allreduce_us` at attach time) and
    the two launches that apply the summed gradients (``EXCHANGE_OVERHEAD_US``).  Without a measurement the static limits of ``REPLICATED_MAX_GLOBAL_BATCH``."""
    if mode != "auto":
        return mode
    if world <= 1:
        return "replicated"
    dt = "bf16" if compute_dtype == "bf16" else "fp32"
    if allreduce_us is None:
        return "replicated" if int(local_batch) * int(world) <= REPLICATED_MAX_GLOBAL_BATCH[dt] else "grad_allreduce"
    replicated = update_us(int(local_batch) * int(world), dt)
    exchange = update_us(int(local_batch), dt) + 2.0 * float(allreduce_us) + EXCHANGE_OVERHEAD_US[dt]
    return "replicated" if replicated <= exchange else "grad_allreduce"


def attach_data_parallel(core, comm: Comm) -> None:
    """Make ``core`` a data-parallel replica: identical start, summed gradients.

    * parameters, targets and Adam moments are broadcast from rank 0;
    * each rank draws a different RNG stream for replay sampling;
    * ``core.grad_sync`` all-reduces a flat gradient range in place such that it
      ends up holding the gradient of the mean loss over the GLOBAL batch.  The
      HIP core already scales its loss by 1/(world*B), so a plain sum is right;
      the torch core averages over its local batch, so the sum is divided by world.
    """
    # D4PG_DP_SELFTEST=1: keep the gradient exchange (three graph segments, two all-reduces, separate
    # optimiser kernels) over a world of one, to exercise and time that path on a single GPU
    if comm.world == 1 and not (os.environ.get("D4PG_DP_SELFTEST") and isinstance(comm, TorchDistComm)):
        core.grad_sync = None
        return
    with torch.no_grad():
        for t in (core.online.flat, core.target.flat, core.adam_m, core.adam_v):
            comm.broadcast_(t.data if hasattr(t, "data") else t, src=0)
    scaled_locally = hasattr(core, "set_world")
    if scaled_locally:
        core.set_world(comm.world)
        st = core.read_dev_state()
        core._write_dev_state(step=core.step, seed=(st.seed + 0x9E3779B97F4A7C15 * (comm.rank + 1)),
                              keep_replay=True)
        def sync_sum(g: torch.Tensor) -> None:
            comm.all_reduce_sum_(g)
        # RCCL collectives are stream-ordered: the update graph captures them (HipCore._run); anything that
        # goes through the host (gloo staging, the in-process fake) stays between graph segments
        sync_sum.capturable = isinstance(comm, TorchDistComm) and comm.backend == "nccl" and \
            not os.environ.get("D4PG_DP_SEGMENTED")
        core.grad_sync = sync_sum
    else:
        inv = 1.0 / comm.world

        def sync(g: torch.Tensor) -> None:
            comm.all_reduce_sum_(g)
            g.mul_(inv)
        core.grad_sync = sync
