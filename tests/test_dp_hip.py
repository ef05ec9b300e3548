"""Gradient-exchange data parallelism of the fused HIP learner cores (GPU only).

* two thread-ranks on ONE GPU with the in-process collective fake: the all-reduced update of two
  half-batches equals one learner's update on the concatenated batch (``set_world``'s 1/(world*B)
  scaling, the stand-alone optimiser + packed-copy refresh, the three-segment path);
* a world-1 RCCL group: the all-reduces captured INTO the update graph make bit for bit the updates
  that the same collectives issued between graph segments make;
* real RCCL over two GPUs (both data-parallel modes), skipped on boxes with fewer than two GPUs.
"""
import os
import threading

import pytest
import torch

from d4pg_pytorch_amd.config import default_config

pytestmark = pytest.mark.gpu


def _cfg(B, dtype, graph=False, model="d4pg", **kw):
    base = dict(env="BipedalWalker-v2", model=model, batch_size=B, device="cuda", v_min=-10.0, v_max=10.0,
                replay_mem_size=40_000, num_steps_train=1000, compute_dtype=dtype, use_hip_graph=graph, random_seed=7)
    base.update(kw)
    return default_config(**base)


def _batch(cfg, B, seed, device="cuda"):
    from d4pg_pytorch_amd.models.core import Batch
    g = torch.Generator().manual_seed(seed)
    S, A = cfg["state_dim"], cfg["action_dim"]
    return Batch(state=torch.randn(B, S, generator=g), action=torch.rand(B, A, generator=g) * 2 - 1,
                 reward=torch.randn(B, generator=g), next_state=torch.randn(B, S, generator=g),
                 done=(torch.rand(B, generator=g) < 0.1).float(),
                 gamma=torch.full((B,), 0.95), weights=torch.rand(B, generator=g) + 0.5,
                 indices=torch.arange(B)).to(device)


def _rows_of(b, lo, hi):
    from d4pg_pytorch_amd.models.core import Batch
    return Batch(b.state[lo:hi], b.action[lo:hi], b.reward[lo:hi], b.next_state[lo:hi], b.done[lo:hi],
                 b.gamma[lo:hi], b.weights[lo:hi], b.indices[lo:hi])


@pytest.mark.parametrize("dtype,model,world", [("fp32", "d4pg", 2), ("fp32", "d3pg", 2), ("bf16", "d4pg", 2),
                                               ("fp32", "d4pg", 4), ("bf16", "d4pg", 4)])
def test_thread_ranks_grad_allreduce_equal_one_learner_on_the_concatenated_batch(dtype, model, world):
    from d4pg_pytorch_amd.models.core import build_core
    from d4pg_pytorch_amd.parallel.comm import attach_data_parallel
    from d4pg_pytorch_amd.parallel.fake_comm import ThreadComm
    Bl, steps = 128 if world == 2 else 64, 5
    cfg_l, cfg_g = _cfg(Bl, dtype, model=model), _cfg(Bl * world, dtype, model=model)
    batches = [_batch(cfg_g, Bl * world, 50 + i) for i in range(steps)]
    comms = ThreadComm.group(world)
    cores, prios, errs = [None] * world, [None] * world, []

    def run(r):
        try:
            torch.cuda.set_device(0)
            core = build_core(cfg_l, "cuda", seed=100 + r)       # different inits: rank 0's is broadcast
            attach_data_parallel(core, comms[r])
            assert core.grad_sync is not None and core.world == world
            for b in batches:
                p = core.update(_rows_of(b, r * Bl, (r + 1) * Bl)).clone()
            torch.cuda.synchronize()
            cores[r], prios[r] = core, p
        except Exception as e:    # pragma: no cover
            errs.append(e)
            comms[r].shared.barrier.abort()

    ts = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    [t.start() for t in ts]
    [t.join(180) for t in ts]
    assert not errs, errs
    # both ranks applied the same summed gradients to the same parameters
    for r in range(1, world):
        assert torch.equal(cores[0].online.flat, cores[r].online.flat)
        assert torch.equal(cores[0].target.flat, cores[r].target.flat)
    single = build_core(cfg_g, "cuda", seed=100)
    for b in batches:
        p_single = single.update(b).clone()
    torch.cuda.synchronize()
    # same update up to the order of the batch sum (two halves summed, then added) ...
    lr = cfg_g["critic_learning_rate"]
    tol = 2e-5 if dtype == "fp32" else 0.6 * lr      # ... and, for bf16, to sign flips of noise-level gradients
    diff = (cores[0].online.flat - single.online.flat).abs()
    assert diff.max().item() < tol * (1 if dtype == "fp32" else steps), diff.max().item()
    assert diff.mean().item() < (2e-6 if dtype == "fp32" else 0.05 * lr)
    torch.testing.assert_close(torch.cat(prios), p_single, rtol=2e-3 if dtype == "fp32" else 5e-2, atol=1e-5)
    # the packed copies the phase kernels read were kept current by the stand-alone optimiser kernel
    want = torch.zeros_like(cores[0].shadow_online)
    if dtype == "fp32":
        from d4pg_pytorch_amd.models.nets import refresh_shadows32
        refresh_shadows32(cores[0].online.flat.detach(), want, cores[0].layout, cores[0].sl)
    else:
        from d4pg_pytorch_amd.models.nets import refresh_shadows
        refresh_shadows(cores[0].online.flat.detach(), want, cores[0].layout, cores[0].sl)
    assert torch.equal(cores[0].shadow_online, want)


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_rccl_world_one_in_graph_all_reduce_equals_segmented(dtype, monkeypatch):
    """The gradient exchange as nodes of the update graph (32 updates per replay) against the same
    collectives issued from the host between three graph segments per update."""
    from d4pg_pytorch_amd.models.core import build_core
    from d4pg_pytorch_amd.parallel.comm import TorchDistComm, attach_data_parallel
    from d4pg_pytorch_amd.replay.hbm_replay import HbmReplay
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(30300 + os.getpid() % 100), RANK="0", WORLD_SIZE="1")
    monkeypatch.setenv("D4PG_DP_SELFTEST", "1")
    comm = TorchDistComm(device=torch.device("cuda", 0), rank=0, world=1)
    outs = {}
    try:
        for mode in ("in_graph", "in_graph_flat", "segmented"):
            if mode == "segmented":
                monkeypatch.setenv("D4PG_DP_SEGMENTED", "1")
            cfg = _cfg(256, dtype, graph=True, sample_ahead=8)
            if mode == "in_graph_flat":      # the element-wise optimiser kernel instead of the tile-shaped apply launch
                cfg["dp_optimizer"] = "flat"
            core = build_core(cfg, "cuda")
            rp = HbmReplay(cfg, "cuda", core.dev_state)
            core.attach_replay(rp)
            g = torch.Generator().manual_seed(1)
            rows = torch.randn(10_000, rp.row_ld, generator=g)
            rows[:, rp.S + rp.A + 1] = (torch.rand(10_000, generator=g) < 0.05).float()
            rows[:, rp.S + rp.A + 2] = 0.95
            rp.insert(rows.cuda())
            attach_data_parallel(core, comm)
            assert core.grad_sync is not None
            done = 0
            while done < 20:
                done += core.update_from_replay_many(20 - done)
            torch.cuda.synchronize()
            assert core._sync_in_graph() == (mode != "segmented")
            assert core.read_dev_state().step_done == 20
            outs[mode] = (core.online.flat.clone(), core.target.flat.clone())
    finally:
        comm.close()
    for other in ("segmented", "in_graph_flat"):
        assert torch.equal(outs["in_graph"][0], outs[other][0]), other
        assert torch.equal(outs["in_graph"][1], outs[other][1]), other


# --------------------------------------------------------------------------- real RCCL, two GPUs
def _rccl_worker(rank, world, port, mode, dtype, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch as th
    from d4pg_pytorch_amd.models.core import build_core
    from d4pg_pytorch_amd.parallel.comm import TorchDistComm, attach_data_parallel
    from d4pg_pytorch_amd.parallel.replicated import attach_replicated
    from d4pg_pytorch_amd.replay.hbm_replay import HbmReplay
    th.cuda.set_device(rank)
    dev = th.device("cuda", rank)
    comm = TorchDistComm(device=dev, rank=rank, world=world)
    Bl, steps = 128, 12
    try:
        if mode == "grad_allreduce":
            cfg_l, cfg_g = _cfg(Bl, dtype, graph=True), _cfg(Bl * world, dtype, graph=True)
            core = build_core(cfg_l, dev, seed=100 + rank)
            attach_data_parallel(core, comm)
            batches = [_batch(cfg_g, Bl * world, 50 + i, dev) for i in range(steps)]
            for b in batches:
                core.update(_rows_of(b, rank * Bl, (rank + 1) * Bl))
            th.cuda.synchronize(dev)
            ok = core._sync_in_graph()
            if rank == 0:
                single = build_core(cfg_g, dev, seed=100)
                for b in batches:
                    single.update(b)
                th.cuda.synchronize(dev)
                diff = (core.online.flat - single.online.flat).abs().max().item()
                th.save({"diff": diff, "in_graph": ok}, os.path.join(out_dir, "r0.pt"))
            # all ranks hold the same parameters
            chk = core.online.flat.detach().double().sum().reshape(1).clone()
            ref = chk.clone()
            comm.broadcast_(ref, src=0)
            assert float((chk - ref).abs().item()) == 0.0
        else:
            cfg_g = _cfg(Bl * world, dtype, graph=True, sample_ahead=4)
            core = build_core(cfg_g, dev, seed=100 + rank)
            rp = HbmReplay(cfg_g, dev, core.dev_state)
            g = th.Generator().manual_seed(10 + rank)
            rows = th.randn(10_000, rp.row_ld, generator=g)
            rows[:, rp.S + rp.A + 1] = (th.rand(10_000, generator=g) < 0.05).float()
            rows[:, rp.S + rp.A + 2] = 0.95
            rp.insert(rows.to(dev))
            dp = attach_replicated(core, rp, comm, Bl, chunk=4)
            done = 0
            while done < steps:
                done += dp.update_many(steps - done)
            drift = dp.check_consistency()
            if rank == 0:
                th.save({"diff": drift, "in_graph": True}, os.path.join(out_dir, "r0.pt"))
    finally:
        comm.close()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (RCCL refuses two ranks on one device)")
@pytest.mark.parametrize("mode,dtype", [("grad_allreduce", "fp32"), ("grad_allreduce", "bf16"), ("replicated", "fp32")])
def test_rccl_two_gpus(mode, dtype, tmp_path):
    import torch.multiprocessing as mp
    port = 30500 + os.getpid() % 400
    ctx = mp.start_processes(_rccl_worker, args=(2, port, mode, dtype, str(tmp_path)), nprocs=2, join=False,
                             start_method="spawn")
    import time
    deadline = time.time() + 150.0          # a collective that never completes must fail the test, not hang the suite
    while not ctx.join(timeout=5.0):
        if time.time() > deadline:
            for p in ctx.processes:
                p.kill()
            pytest.fail("two-rank RCCL run did not finish within 150 s")
    res = torch.load(tmp_path / "r0.pt", weights_only=True)
    if mode == "replicated":
        assert res["diff"] == 0.0
    else:
        assert res["in_graph"]
        assert res["diff"] < (2e-5 if dtype == "fp32" else 12 * 0.6 * 5e-4)
