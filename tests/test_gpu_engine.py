"""GPU topology end to end: HipCore learner + HBM replay fed by CPU explorers through
shared-memory rings, GPU exploiter with batched inference (GPU only)."""
import glob
import os

import numpy as np
import pytest
import torch

from d4pg_pytorch_amd.config import default_config
from d4pg_pytorch_amd.engine import load_engine
from d4pg_pytorch_amd.utils.checkpoint import load_learner_bundle
from d4pg_pytorch_amd.utils.logger import read_events

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("model,per", [("d4pg", 1), ("d3pg", 0)])
def test_hbm_topology_trains_on_the_gpu(tmp_path, model, per):
    cfg = dict(env="Pendulum-v0", model=model, num_agents=3, batch_size=128, num_steps_train=400,
               replay_mem_size=20_000, n_step_returns=5, device="cuda", agent_device="cpu", dense_size=128,
               v_min=-20, v_max=0, replay_memory_prioritized=per, metrics_interval=50,
               weight_publish_interval=50, envs_per_agent=4, num_episode_save=1,
               results_path=str(tmp_path / "results"))
    engine = load_engine(cfg)
    assert engine.config["replay_device"] == "hbm"
    exp = engine.train()
    assert engine.final_update_step == 400
    assert all(c == 0 for c in engine.exit_codes.values()), engine.exit_codes
    assert "sampler" not in engine.exit_codes            # sampling happens on the device
    ev = read_events(os.path.join(exp, "learner"))
    tags = {e[2] for e in ev}
    assert {"learner/policy_loss", "learner/value_loss", "data_struct/replay_buffer", "perf/updates_per_s"} <= tags
    assert all(np.isfinite(e[3]) for e in ev)
    assert max(e[3] for e in ev if e[2] == "data_struct/replay_buffer") >= 128
    assert max(e[3] for e in ev if e[2] == "data_struct/ring_dropped") == 0      # lossless ingest
    b = load_learner_bundle(os.path.join(exp, "learner", "checkpoint_final.pt"))
    assert b["update_step"] == 400 and all(torch.isfinite(v).all() for v in b["critic"].values())
    assert glob.glob(os.path.join(exp, "agent_0", "local_episode_*_reward_*.pt"))


@pytest.mark.parametrize("width", [300, 72])
def test_any_hidden_width_runs_the_fused_kernels_through_the_whole_topology(tmp_path, width):
    """The reference takes any ``dense_size`` (models/d4pg/networks.py:10,23-25; its own test file uses 100).  Widths
    with compile-time 4x4x1 kernels (300) and widths without (72: the run-time-width 16x16x4 kernels) both run the
    fused pipeline: learner (graph, PER through the HBM replay), GPU exploiter with single-launch inference and device
    explorers, end to end."""
    from d4pg_pytorch_amd.actors.gpu_actor import GpuActor
    from d4pg_pytorch_amd.models.core import build_core
    cfg = dict(env="Pendulum-v0", model="d4pg", num_agents=2, batch_size=128, num_steps_train=600,
               replay_mem_size=20_000, n_step_returns=5, device="cuda", agent_device="cuda", dense_size=width,
               v_min=-20, v_max=0, replay_memory_prioritized=1, metrics_interval=100, weight_publish_interval=50,
               envs_per_agent=4, gpu_rollout_envs=32, gpu_rollout_segment=32, num_episode_save=1000,
               results_path=str(tmp_path / "results"))
    core = build_core(default_config(**{k: v for k, v in cfg.items() if k != "results_path"}), "cuda")
    assert type(core).__name__ == "HipCoreF32" and core.wide == (width == 72)
    assert GpuActor(3, 1, width, max_batch=8).launches_per_act == 1
    del core
    engine = load_engine(cfg)
    exp = engine.train()
    assert engine.final_update_step == 600
    assert all(c == 0 for c in engine.exit_codes.values()), engine.exit_codes
    ev = read_events(os.path.join(exp, "learner"))
    assert all(np.isfinite(e[3]) for e in ev)
    losses = [e[3] for e in ev if e[2] == "learner/value_loss"]
    # the critic fits its targets: the cross-entropy of an untrained critic against a projected target is ln 51 = 3.9
    assert len(losses) >= 4 and max(losses[1:]) < 1.0, losses
    assert engine.final_env_steps >= 32 * 32
    b = load_learner_bundle(os.path.join(exp, "learner", "checkpoint_final.pt"))
    assert any(tuple(v.shape) == (width, width) for v in b["critic"].values())
    assert all(torch.isfinite(v).all() for v in b["critic"].values())


def test_gpu_actor_matches_cpu_actor():
    from d4pg_pytorch_amd.actors.gpu_actor import GpuActor
    from d4pg_pytorch_amd.actors.native import CpuActor
    from d4pg_pytorch_amd.models.nets import ActorPolicy, init_net_
    S, A, Hd, E = 24, 4, 400, 48
    pol = ActorPolicy(S, A, Hd)
    init_net_(pol.view, 0.3, torch.Generator().manual_seed(1))
    cpu, gpu = CpuActor(S, A, Hd), GpuActor(S, A, Hd, max_batch=E)
    cpu.load_flat(pol.flat.numpy())
    gpu.load_flat(pol.flat.numpy())
    x = np.random.default_rng(0).standard_normal((E, S)).astype(np.float32)
    np.testing.assert_allclose(gpu.act(x), cpu.act(x), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(gpu.act(x[:5]), cpu.act(x[:5]), rtol=1e-4, atol=1e-5)
    assert set(gpu.state_dict()) == set(cpu.state_dict())


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_data_parallel_wiring_on_one_gpu(dtype):
    """world=1 NCCL(RCCL) group on the single GPU: the segmented-graph path with the
    all-reduce between segments must reproduce the fused single-graph path (which also pits the
    stand-alone Adam/Polyak/packed-copy kernel against the one fused into the weight-gradient kernel)."""
    from d4pg_pytorch_amd.models.core import build_core
    from d4pg_pytorch_amd.parallel.comm import TorchDistComm
    from d4pg_pytorch_amd.replay.hbm_replay import HbmReplay
    port = 29800 + os.getpid() % 100 + (100 if dtype == "bf16" else 0)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
    comm = TorchDistComm(device=torch.device("cuda", 0), rank=0, world=1)
    try:
        cfg = default_config(env="Walker2d-v2", model="d4pg", batch_size=256, device="cuda", v_min=0.0, v_max=50.0,
                             replay_mem_size=50_000, num_steps_train=100, compute_dtype=dtype)
        rows = torch.randn(10_000, 44, device="cuda")
        rows[:, 17 + 6 + 1] = 0.0
        rows[:, 17 + 6 + 2] = 0.95
        outs = {1: [], 10: []}
        for use_sync in (False, True):
            core = build_core(cfg, "cuda")
            rp = HbmReplay(cfg, "cuda", core.dev_state)
            core.attach_replay(rp)
            rp.insert(rows)
            if use_sync:
                core.grad_sync = lambda g: comm.all_reduce_sum_(g)    # forces the 3-graph path
            for it in range(1, 11):
                core.update_from_replay()
                if it in outs:
                    torch.cuda.synchronize()
                    outs[it].append((core.online.flat.clone(), core.target.flat.clone()))
            if dtype == "bf16":
                # the shadows kept by the stand-alone optimiser kernel == a rebuild from the masters
                from d4pg_pytorch_amd.models.nets import refresh_shadows
                want = torch.zeros_like(core.shadow_online)
                refresh_shadows(core.online.flat.detach(), want, core.layout, core.sl)
                assert torch.equal(core.shadow_online, want)
        # the optimiser runs in two different kernels (fused into the weight-gradient kernel / stand-alone) that share
        # one contraction-free definition of the arithmetic (adam_polyak_math, csrc/common.h): bit-identical
        for it in (1, 10):
            assert torch.equal(outs[it][0][0], outs[it][1][0]), (it, (outs[it][0][0] - outs[it][1][0]).abs().max().item())
            assert torch.equal(outs[it][0][1], outs[it][1][1]), it
    finally:
        comm.close()


def _ipc_child(_rank, handle, numel, S, A, Hd, out_path, device="cuda:0"):
    import numpy as _np
    import torch as th
    from d4pg_pytorch_amd.actors.gpu_actor import GpuActor
    from d4pg_pytorch_amd.parallel.hip_ipc import SharedDeviceBuffer
    th.cuda.set_device(th.device(device))
    shared = SharedDeviceBuffer.open(handle, numel, device)
    ga = GpuActor(S, A, Hd, max_batch=16, device=device)
    ga.attach_shared(shared, 1)                       # the target half
    x = _np.random.default_rng(0).standard_normal((16, S)).astype(_np.float32)
    outs = []
    for version in (1, 2):
        assert ga.refresh_shared(version)
        outs.append(th.from_numpy(ga.act(x)))
    th.save({"acts": th.stack(outs), "sum": float(shared.tensor.double().sum().item())}, out_path)
    shared.close()


def test_hip_ipc_weights_are_read_in_place_by_another_process(tmp_path):
    """The learner's [online | target] actor slots, exported through a HIP IPC handle, drive a GpuActor in a
    second process: zero host copies (what share_memory() gives the reference's exploiter)."""
    import torch.multiprocessing as mp
    from d4pg_pytorch_amd.actors.native import CpuActor
    from d4pg_pytorch_amd.models.nets import ActorPolicy, init_net_
    from d4pg_pytorch_amd.parallel.hip_ipc import SharedDeviceBuffer
    S, A, Hd = 24, 4, 400
    pols = [ActorPolicy(S, A, Hd) for _ in range(4)]
    for i, p in enumerate(pols):
        init_net_(p.view, 0.3, torch.Generator().manual_seed(10 + i))
    n = pols[0].layout.numel
    shared = SharedDeviceBuffer.create(4 * n, "cuda:0")
    shared.tensor.copy_(torch.cat([p.flat for p in pols]).cuda())      # slot 0: [p0 | p1], slot 1: [p2 | p3]
    torch.cuda.synchronize()
    out = str(tmp_path / "child.pt")
    mp.start_processes(_ipc_child, args=(shared.handle, shared.numel, S, A, Hd, out), nprocs=1, join=True,
                       start_method="spawn")
    res = torch.load(out, weights_only=True)
    x = np.random.default_rng(0).standard_normal((16, S)).astype(np.float32)
    for k, pol in enumerate((pols[1], pols[3])):       # version 1 -> slot 0, version 2 -> slot 1; target halves
        cpu = CpuActor(S, A, Hd)
        cpu.load_flat(pol.flat.numpy())
        np.testing.assert_allclose(res["acts"][k].numpy(), cpu.act(x), rtol=1e-4, atol=1e-5)
    assert abs(res["sum"] - float(torch.cat([p.flat for p in pols]).double().sum())) < 1e-6
    shared.close()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs")
def test_hip_ipc_weights_are_read_over_xgmi_by_an_agent_on_another_gpu(tmp_path):
    """The same slots mapped as PEER memory by a process whose GpuActor lives on the second GPU (how an exploiter on
    another device follows the learner without a copy through the host)."""
    import torch.multiprocessing as mp
    from d4pg_pytorch_amd.actors.native import CpuActor
    from d4pg_pytorch_amd.models.nets import ActorPolicy, init_net_
    from d4pg_pytorch_amd.parallel.hip_ipc import SharedDeviceBuffer
    if not torch.cuda.can_device_access_peer(1, 0):
        pytest.skip("no peer access between GPU 1 and GPU 0")
    S, A, Hd = 24, 4, 400
    pols = [ActorPolicy(S, A, Hd) for _ in range(4)]
    for i, p in enumerate(pols):
        init_net_(p.view, 0.3, torch.Generator().manual_seed(20 + i))
    shared = SharedDeviceBuffer.create(4 * pols[0].layout.numel, "cuda:0")
    shared.tensor.copy_(torch.cat([p.flat for p in pols]).cuda())
    torch.cuda.synchronize()
    out = str(tmp_path / "child.pt")
    mp.start_processes(_ipc_child, args=(shared.handle, shared.numel, S, A, Hd, out, "cuda:1"), nprocs=1, join=True,
                       start_method="spawn")
    res = torch.load(out, weights_only=True)
    x = np.random.default_rng(0).standard_normal((16, S)).astype(np.float32)
    for k, pol in enumerate((pols[1], pols[3])):
        cpu = CpuActor(S, A, Hd)
        cpu.load_flat(pol.flat.numpy())
        np.testing.assert_allclose(res["acts"][k].numpy(), cpu.act(x), rtol=1e-4, atol=1e-5)
    shared.close()


def test_weight_publisher_hands_over_without_blocking():
    from d4pg_pytorch_amd.models.core import build_core
    from d4pg_pytorch_amd.models.learner import WeightPublisher
    from d4pg_pytorch_amd.parallel.ipc import ControlBlock, SlabReader, WeightSlab
    cfg = default_config(env="Pendulum-v0", model="d4pg", batch_size=64, dense_size=128, device="cuda",
                         v_min=-20.0, v_max=0.0)
    core = build_core(cfg, "cuda")
    n = core.layout.actor.numel
    with ControlBlock.create(8) as ctrl, WeightSlab.create(2 * n) as slab:
        pub = WeightPublisher(core, slab, ctrl, "cuda")
        reader = SlabReader(slab)
        for k in range(1, 6):
            with torch.no_grad():
                core.online.flat.add_(0.01)           # "an update"
            want = torch.cat([core.online.actor_flat().detach(), core.target.actor_flat().detach()]).cpu()
            pub.publish()
            assert pub.issued == k
            with torch.no_grad():
                core.online.flat.add_(1.0)            # later work on the update stream must not leak into the snapshot
                core.online.flat.sub_(1.0)
            pub.poll(block=True)
            assert ctrl.load(ControlBlock.GPU_WEIGHT_VERSION) == k and reader.poll()
            assert torch.equal(torch.from_numpy(reader.buf), want)
            assert torch.equal(pub.stage[(k - 1) % 2].cpu(), want)
        # several publications without a poll in between: nothing is lost, order is kept
        pub.publish()
        pub.publish()
        pub.publish()
        pub.poll(block=True)
        assert pub.completed == pub.issued == 8 and ctrl.load(ControlBlock.GPU_WEIGHT_VERSION) == 8


def test_device_explorers_and_ipc_exploiter_end_to_end(tmp_path):
    """No CPU explorer at all: the replay is filled by environments stepped on the learner's GPU; the GPU
    exploiter follows the learner's weights through the HIP IPC buffer."""
    cfg = dict(env="Pendulum-v0", model="d4pg", num_agents=1, batch_size=128, num_steps_train=600,
               replay_mem_size=100_000, n_step_returns=5, device="cuda", dense_size=128,
               v_min=-20, v_max=0, metrics_interval=100, weight_publish_interval=50, num_episode_save=1,
               gpu_rollout_envs=256, gpu_rollout_segment=16, gpu_rollout_interval=32,
               results_path=str(tmp_path / "results"))
    engine = load_engine(cfg)
    assert engine.config["num_agents"] == 1
    exp = engine.train()
    assert engine.final_update_step == 600
    assert all(c == 0 for c in engine.exit_codes.values()), engine.exit_codes
    assert set(engine.exit_codes) == {"learner-0", "agent-0"}
    assert engine.final_env_steps >= 256 * 16 * 10
    ev = read_events(os.path.join(exp, "learner"))
    tags = {e[2] for e in ev}
    assert {"learner/policy_loss", "agent/reward_device_explorers", "data_struct/replay_buffer"} <= tags
    assert max(e[3] for e in ev if e[2] == "data_struct/replay_buffer") >= 256 * 16 * 10
    agent_tags = {e[2] for e in read_events(os.path.join(exp, "agent-0"))}
    assert "agent/shared_device_weights" in agent_tags
