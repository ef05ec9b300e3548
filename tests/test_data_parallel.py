"""Data-parallel learners without a cluster: an in-process fake collective and a
real 2-rank gloo process group on the CPU.  Oracle: one rank on the concatenated batch."""
import multiprocessing as mp
import os
import threading

import pytest
import torch

from d4pg_pytorch_amd.config import default_config
from d4pg_pytorch_amd.models.core import Batch, TorchCore
from d4pg_pytorch_amd.parallel.comm import LocalComm, attach_data_parallel
from d4pg_pytorch_amd.parallel.fake_comm import ThreadComm


def _cfg(model, B, seed=5):
    return default_config(env="Walker2d-v2", model=model, device="cpu", dense_size=32, batch_size=B,
                          v_min=-5.0, v_max=5.0, random_seed=seed)


def _batch(B, seed, S=17, A=6):
    g = torch.Generator().manual_seed(seed)
    return Batch(torch.randn(B, S, generator=g), torch.rand(B, A, generator=g) * 2 - 1, torch.randn(B, generator=g),
                 torch.randn(B, S, generator=g), (torch.rand(B, generator=g) < 0.2).float(),
                 torch.full((B,), 0.95), None, torch.arange(B))


def _split(b, r, w):
    n = b.size // w
    sl = slice(r * n, (r + 1) * n)
    return Batch(b.state[sl], b.action[sl], b.reward[sl], b.next_state[sl], b.done[sl], b.gamma[sl], None, b.indices[sl])


@pytest.mark.parametrize("model", ["d4pg", "d3pg"])
def test_two_fake_ranks_equal_one_rank_on_the_full_batch(model):
    world, B = 2, 32
    single = TorchCore(_cfg(model, B), "cpu")
    comms = ThreadComm.group(world)
    cores = [TorchCore(_cfg(model, B // world, seed=100 + r), "cpu") for r in range(world)]   # different inits
    batches = [_batch(B, 10 + i) for i in range(4)]
    errs = []

    def run(r):
        try:
            attach_data_parallel(cores[r], comms[r])          # broadcast from rank 0 -> identical start
            for b in batches:
                cores[r].update(_split(b, r, world))
        except Exception as e:    # pragma: no cover
            errs.append(e)
            comms[r].shared.barrier.abort()

    # rank 0 must start from the single-rank weights
    with torch.no_grad():
        cores[0].online.flat.copy_(single.online.flat)
        cores[0].target.flat.copy_(single.target.flat)
    ts = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    [t.start() for t in ts]
    [t.join(120) for t in ts]
    assert not errs, errs
    for b in batches:
        single.update(b)
    torch.testing.assert_close(cores[0].online.flat.detach(), cores[1].online.flat.detach(), rtol=0, atol=0)
    torch.testing.assert_close(cores[0].online.flat.detach(), single.online.flat.detach(), rtol=0, atol=2e-6)
    torch.testing.assert_close(cores[1].target.flat, single.target.flat, rtol=0, atol=2e-6)


def test_world_one_is_a_no_op():
    core = TorchCore(_cfg("d4pg", 8), "cpu")
    attach_data_parallel(core, LocalComm())
    assert core.grad_sync is None


def _gloo_rank(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(1)
    from d4pg_pytorch_amd.parallel.comm import TorchDistComm
    comm = TorchDistComm(backend="gloo", rank=rank, world=world, device=torch.device("cpu"))
    core = TorchCore(_cfg("d4pg", 16, seed=200 + rank), "cpu")
    attach_data_parallel(core, comm)
    for i in range(3):
        core.update(_split(_batch(32, 10 + i), rank, world))
    comm.barrier()
    assert comm.all_reduce_max_scalar(float(rank)) == world - 1
    torch.save({"online": core.online.flat.detach(), "target": core.target.flat}, os.path.join(out_dir, f"r{rank}.pt"))
    comm.close()


def test_two_gloo_processes_stay_bit_identical(tmp_path):
    ctx = mp.get_context("spawn")
    port = 29700 + os.getpid() % 200
    ps = [ctx.Process(target=_gloo_rank, args=(r, 2, port, str(tmp_path))) for r in range(2)]
    [p.start() for p in ps]
    [p.join(180) for p in ps]
    assert all(p.exitcode == 0 for p in ps), [p.exitcode for p in ps]
    a = torch.load(tmp_path / "r0.pt", weights_only=True)
    b = torch.load(tmp_path / "r1.pt", weights_only=True)
    assert torch.equal(a["online"], b["online"]) and torch.equal(a["target"], b["target"])
    # and they match a single rank that saw both shards (rank 0's init is what was broadcast)
    single = TorchCore(_cfg("d4pg", 32, seed=200), "cpu")
    for i in range(3):
        single.update(_batch(32, 10 + i))
    torch.testing.assert_close(a["online"], single.online.flat.detach(), rtol=0, atol=2e-6)


def test_dp_mode_auto_replicates_only_small_global_batches():
    from d4pg_pytorch_amd.config import default_config
    from d4pg_pytorch_amd.parallel.comm import REPLICATED_MAX_GLOBAL_BATCH, resolve_dp_mode
    assert default_config(env="Pendulum-v0")["dp_mode"] == "auto"
    assert REPLICATED_MAX_GLOBAL_BATCH == {"fp32": 2048, "bf16": 8192}
    assert resolve_dp_mode("auto", 256, 8) == "replicated"
    assert resolve_dp_mode("auto", 1024, 8) == "grad_allreduce"            # fp32 (the default): 8192 rows are too many
    assert resolve_dp_mode("auto", 1024, 8, "bf16") == "replicated"
    assert resolve_dp_mode("auto", 4096, 8, "bf16") == "grad_allreduce"
    assert resolve_dp_mode("auto", 4096, 8) == "grad_allreduce"
    assert resolve_dp_mode("auto", 16384, 1) == "replicated"         # one rank has nobody to exchange with
    assert resolve_dp_mode("grad_allreduce", 256, 2) == "grad_allreduce"
    assert resolve_dp_mode("replicated", 4096, 8) == "replicated"


def test_dp_mode_auto_follows_the_measured_all_reduce_time():
    """With a probe the choice is a cost comparison: update(global batch) against update(local batch) + two all-reduces
    + the stand-alone optimiser kernels."""
    from d4pg_pytorch_amd.parallel.comm import EXCHANGE_OVERHEAD_US, resolve_dp_mode, update_us
    assert update_us(256) == 78.0 and update_us(128) == 78.0
    assert update_us(640) == pytest.approx((86.0 + 99.0) / 2)
    assert update_us(16384) == pytest.approx(2 * 358.0)                    # linear in the batch past the table
    # headline shape, 8 ranks x 256 rows: replicated 2048 rows = 144 us; exchange = 78 + 8 + 2 t
    assert resolve_dp_mode("auto", 256, 8, "fp32", allreduce_us=25.0) == "grad_allreduce"      # 136 < 144
    assert resolve_dp_mode("auto", 256, 8, "fp32", allreduce_us=35.0) == "replicated"          # 156 > 144
    # a slow fabric (gloo over loopback: milliseconds) replicates even batches the static limit would exchange
    assert resolve_dp_mode("auto", 4096, 8, "fp32") == "grad_allreduce"
    assert resolve_dp_mode("auto", 4096, 8, "fp32", allreduce_us=5000.0) == "replicated"
    # and a fast one exchanges where the static limit would replicate
    assert resolve_dp_mode("auto", 1024, 2, "fp32") == "replicated"
    assert resolve_dp_mode("auto", 1024, 2, "fp32", allreduce_us=5.0) == "grad_allreduce"      # 107 + 8 + 10 < 144
    # bf16 at 8 x 256: the update itself is so cheap (64 us for 2048 rows) that only an all-reduce under 2.5 us would pay
    t_break = (update_us(2048, "bf16") - update_us(256, "bf16") - EXCHANGE_OVERHEAD_US["bf16"]) / 2
    assert 0 < t_break < 5
    assert resolve_dp_mode("auto", 256, 8, "bf16", allreduce_us=0.2) == "grad_allreduce"
    assert resolve_dp_mode("auto", 256, 8, "bf16", allreduce_us=10.0) == "replicated"
    assert resolve_dp_mode("grad_allreduce", 256, 8, "bf16", allreduce_us=1e9) == "grad_allreduce"


def test_probe_times_the_live_communicator_and_agrees_across_ranks():
    import threading
    from d4pg_pytorch_amd.parallel.comm import LocalComm, probe_allreduce_us
    from d4pg_pytorch_amd.parallel.fake_comm import ThreadComm
    assert probe_allreduce_us(LocalComm(), 1000, "cpu", iters=3) >= 0.0
    comms, out = ThreadComm.group(4), [None] * 4

    def run(r):
        out[r] = probe_allreduce_us(comms[r], 50_000, "cpu", iters=5)

    ts = [threading.Thread(target=run, args=(r,)) for r in range(4)]
    [t.start() for t in ts]
    [t.join(60) for t in ts]
    assert all(o is not None and o > 0 for o in out), out
    assert len(set(out)) == 1, out          # the maximum over the ranks: every rank decides from the same number
