#!/usr/bin/env python
"""Headline benchmark: learner updates/sec on synthetic transitions.

    python bench.py --gpus N --steps K --warmup W

One *step* is one complete learner update of the named configuration: on-device
replay sampling + gather, target actor/critic forward, categorical projection,
critic loss/backward/Adam, actor loss through the updated critic, actor
backward/Adam, Polyak on both targets (and the priority write-back with --per).
Nothing is skipped or cached; weights are random-init, the replay is filled
with synthetic transitions of the environment's observation/action shape.

Configurations:
  N == 1 : LunarLanderContinuous-v2 D4PG, batch 256, n-step 5, 51 atoms, 400-wide MLPs
  N  > 1 : Walker2d-v2 D4PG, data-parallel learners over RCCL, batch 256 per rank (weak
           scaling: every rank draws 256 rows per update from its own HBM replay shard;
           one synchronous update consumes the global batch of N x 256 rows)
Other shapes: --env Ant-v2 --batch 4096 --model d3pg --per ...

Precision: the headline (``value``, ``dtype: fp32``) is the reference's own precision — every product an exact
f32 multiply-add on the matrix cores.  For N == 1 the same run also times the bf16-operand pipeline and
reports it as ``bf16_value`` / ``bf16_ms_per_step``, and the reference's op sequence in eager PyTorch-ROCm as
``eager_torch_value`` (``speedup_vs_eager_torch``; ``--no-bf16`` skips both).  A request of fewer than 256 steps is timed as 15
regions of exactly that many updates each (``repeats``), and the median region is reported.

Data-parallel modes (``--dp-mode``, N > 1):
  replicated      the sampled rows are all-gathered (one ~20 KB/rank collective, one update
                  ahead on a side stream) and every rank runs the update on the global
                  batch as ONE graph; no gradient traffic.  At small batches a learner update is
                  latency bound, not throughput bound (fp32: 77 us at batch 256, 86 at 512, 105 at
                  1024, 147 at 2048 on one MI355X; bf16: 52 / 55 / 61 / 67), so replicating the
                  arithmetic competes with two latency-bound gradient all-reduces on the critical path.
  grad_allreduce  every rank runs the update on its 256 rows and the critic and actor
                  gradient buckets (0.7-0.9 MB each) are all-reduced before each Adam step; with RCCL
                  the two all-reduces are nodes of the update's graph (32 updates per replay).
  both (default)  times the two back to back; ``value`` is the faster one (``dp_mode_selected``), both are reported
                  (``replicated_value``, ``grad_allreduce_value``) together with what the framework's static rule
                  picks (``dp_mode_auto``: replicated up to a global batch of 2048 rows in fp32, 8192 in bf16; gradient all-reduce beyond).
Both make the update a single learner would make on the concatenated global batch.
For N > 1 every rank also times two single-GPU runs without collectives on its own GPU:
``single_gpu_same_config_value`` (the per-rank configuration, batch 256: what the N-GPU ``value``
divided by N compares with, since the N == 1 default is the LunarLander shape) and
``single_gpu_global_batch_value`` (ONE GPU running the job's global batch of N x 256 rows per update,
in the same batch-256 units).  ``speedup_vs_single_gpu_global_batch`` = ``value`` over the latter is
what the extra GPUs buy the LEARNER; an update of these 400-wide MLPs is a latency-bound chain of small
layers, so at small per-rank batches that ratio is near or below 1 and the extra GPUs pay for themselves
through replay capacity, ingest bandwidth and exploiters, not through learner arithmetic.

For N > 1 the driver launches this file under torch.distributed.run (one rank
per GPU).  Rank 0 prints ONE JSON line.  ``value`` aggregates all ranks:
N x (synchronous updates/s), i.e. batch-256 learner updates per second summed
over the GPUs; ``sync_updates_per_sec`` is the optimiser-step rate of the job.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

# RCCL shares device buffers between the ranks' processes through dmabuf IPC handles; the legacy IPC
# mode fails with "hipIpcGetMemHandle: invalid argument" on hosts whose driver only offers dmabuf.
# Must be in the environment before the HIP runtime starts.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3000)
    ap.add_argument("--warmup", type=int, default=300)
    ap.add_argument("--env", default="auto")
    ap.add_argument("--model", default="d4pg", choices=["d4pg", "d3pg", "ddpg"])
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--per", action="store_true", help="prioritised replay (HBM sum-tree) instead of uniform")
    ap.add_argument("--replay-size", type=int, default=1_000_000)
    ap.add_argument("--prefill", type=int, default=200_000)
    ap.add_argument("--dtype", default="fp32", choices=["fp32", "bf16"],
                    help="fp32 (the reference's precision, the headline): fused row-local pipeline on exact f32 MFMA; "
                         "bf16: the same pipeline with bf16 MFMA operands (fp32 masters / accumulate)")
    ap.add_argument("--no-bf16", action="store_true",
                    help="N == 1: do not also time the bf16 pipeline (reported as bf16_value next to the fp32 headline)")
    ap.add_argument("--repeats", type=int, default=0,
                    help="timed regions of exactly --steps updates each; the median is reported "
                         "(0: 15 when --steps < 256, so that a short request is not a 1 ms sample, else 1)")
    ap.add_argument("--layers", action="store_true", help="fp32 with one launch per layer (the unfused HIP kernels)")
    ap.add_argument("--baseline", action="store_true",
                    help="time the unfused eager-PyTorch update (reference op sequence) instead")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--tile-rows", default="auto", choices=["auto", "4", "8", "16", "32"])
    ap.add_argument("--mfma", default="auto", choices=["auto", "4x4", "16x16"],
                    help="fp32 phase kernels: v_mfma_f32_4x4x1 (compiled hidden widths) or v_mfma_f32_16x16x4 (any width)")
    ap.add_argument("--occupancy", default="auto", choices=["auto", "1", "2"],
                    help="fp32 16x16x4 phase kernels: workgroups per CU (2: the lean build; auto: from 256 row tiles)")
    ap.add_argument("--hidden", type=int, default=400, help="dense_size (400: the reference's)")
    ap.add_argument("--dp-mode", default="both", choices=["both", "replicated", "grad_allreduce"],
                    help="data-parallel scheme for N > 1 (see the module docstring)")
    ap.add_argument("--chunk", type=int, default=0,
                    help="updates per drawn-ahead chunk of batches = per graph replay (and per all-gather for N > 1); "
                         "0: the configuration default (sample_ahead)")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the extra measurements beside the headline (N == 1: prioritised replay, the scalar critic and "
                         "the Ant-shape batch-4096 stress in fp32; N > 1: rollout workers on ranks >= 1)")
    ap.add_argument("--dp-selftest", action="store_true",
                    help="with one GPU: run the N > 1 code path (both data-parallel modes) over a world-size-1 "
                         "RCCL group, to check the wiring on a single-GPU box")
    return ap.parse_args()


def synthetic_rows(n: int, S: int, A: int, row_ld: int, gamma_n: float, device, seed: int) -> torch.Tensor:
    g = torch.Generator(device=device).manual_seed(seed)
    rows = torch.zeros(n, row_ld, device=device)
    rows[:, :S] = torch.randn(n, S, device=device, generator=g)
    rows[:, S:S + A] = torch.rand(n, A, device=device, generator=g) * 2 - 1
    rows[:, S + A] = torch.randn(n, device=device, generator=g)
    rows[:, S + A + 1] = (torch.rand(n, device=device, generator=g) < 0.02).float()
    rows[:, S + A + 2] = gamma_n
    rows[:, S + A + 3: S + A + 3 + S] = torch.randn(n, S, device=device, generator=g)
    return rows


def main() -> None:
    args = parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world == 1 and args.gpus > 1:
        # not launched by torch.distributed.run: spawn the ranks ourselves (before any GPU use)
        import subprocess
        port = 29500 + (os.getpid() % 2000)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))

    from d4pg_pytorch_amd.config import default_config
    from d4pg_pytorch_amd.models.core import build_core
    from d4pg_pytorch_amd.parallel.comm import LocalComm, TorchDistComm, attach_data_parallel, resolve_dp_mode
    from d4pg_pytorch_amd.replay.hbm_replay import HbmReplay

    # one GPU per rank; ranks wrap around only when a multi-rank job is rehearsed on fewer GPUs
    # (D4PG_DIST_BACKEND=gloo, see parallel/comm.py)
    dev_index = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if world == 1 and args.dp_selftest:
        os.environ["D4PG_DP_SELFTEST"] = "1"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(29400 + os.getpid() % 500))
        comm = TorchDistComm(device=device, rank=0, world=1)
    else:
        comm = TorchDistComm(device=device) if world > 1 else LocalComm()

    repeats = args.repeats if args.repeats > 0 else (15 if args.steps < 256 else 1)
    env = args.env
    if env == "auto":
        env = "LunarLanderContinuous-v2" if world == 1 else "Walker2d-v2"
    vrange = {"LunarLanderContinuous-v2": (-200.0, 200.0), "Walker2d-v2": (0.0, 5000.0),
              "HalfCheetah-v2": (0.0, 10000.0), "BipedalWalker-v2": (-100.0, 300.0),
              "Ant-v2": (0.0, 4000.0), "Pendulum-v0": (-1000.0, 0.0)}.get(env, (-100.0, 100.0))
    cfg = default_config(env=env, model=args.model, batch_size=args.batch, n_step_returns=5,
                         dense_size=args.hidden, num_atoms=51, v_min=vrange[0], v_max=vrange[1],
                         device="cuda", replay_mem_size=args.replay_size,
                         replay_memory_prioritized=int(args.per), use_hip_graph=not args.no_graph,
                         compute_dtype=args.dtype, phase_tile_rows=args.tile_rows, phase_mfma=args.mfma, phase_occupancy=args.occupancy,
                         num_steps_train=max(1, args.steps * max(1, repeats) + args.warmup),
                         fused_learner="torch" if args.baseline else ("layers" if args.layers else "auto"), random_seed=2019)
    if args.chunk > 0:
        cfg["sample_ahead"] = args.chunk
    S, A = cfg["state_dim"], cfg["action_dim"]
    gamma_n = cfg["discount_rate"] ** cfg["n_step_returns"]

    def fill(replay):
        n_fill = min(args.prefill, args.replay_size)
        chunk = 50_000
        for off in range(0, n_fill, chunk):
            n = min(chunk, n_fill - off)
            replay.insert(synthetic_rows(n, S, A, replay.row_ld, gamma_n, device, seed=1000 * (rank + 1) + off))

    def timed(one_step, many=None):
        """``one_step()`` makes one update; ``many(n)`` makes up to n and returns the count (whole
        drawn-ahead chunks of updates are one graph replay).  Exactly ``warmup`` then exactly
        ``steps`` updates are made."""
        def run(total):
            done = 0
            while done < total:
                if many is not None:
                    done += many(total - done)
                else:
                    one_step()
                    done += 1
        run(args.warmup)
        times = []
        for _ in range(repeats):
            comm.barrier()
            torch.cuda.synchronize(device)
            t0 = time.perf_counter()
            run(args.steps)
            comm.barrier()
            torch.cuda.synchronize(device)
            times.append(comm.all_reduce_max_scalar(time.perf_counter() - t0))
        times.sort()
        return times[len(times) // 2]


    def quick_value(overrides, steps, warm):
        """updates/s of a variation of the run's configuration (one GPU, no collectives): a short extra measurement."""
        c = default_config(**{**cfg, **overrides, "num_steps_train": steps + warm + 64})
        core_q = build_core(c, device)
        rp = HbmReplay(c, device, core_q.dev_state)
        core_q.attach_replay(rp)
        Sq, Aq = c["state_dim"], c["action_dim"]
        for off in range(0, 100_000, 50_000):
            rp.insert(synthetic_rows(50_000, Sq, Aq, rp.row_ld, gamma_n, device, seed=77 + off))
        many = getattr(core_q, "update_from_replay_many", None)

        def run(total):
            done = 0
            while done < total:
                if many is not None:
                    done += many(total - done)
                else:
                    core_q.update_from_replay()
                    done += 1
        run(warm)
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        run(steps)
        torch.cuda.synchronize(device)
        dt_q = time.perf_counter() - t0
        del core_q, rp
        torch.cuda.empty_cache()
        return steps / dt_q

    def rollout_workers_extra(seconds=4.0):
        """N > 1: what the other GPUs buy as ROLLOUT devices.  Rank 0 is the learner (this run's per-rank
        configuration, HBM replay); every other rank is a rollout worker (actors/rollout_worker.py) stepping 4096
        device environments of the task on its own GPU, reading the online actor out of rank 0's HIP-IPC slots and
        delivering its rows into rank 0's HBM (parallel/row_mailbox.py).  Returns environment steps/s of the job and
        the learner's update rate beside that ingest."""
        import torch.distributed as dist
        from d4pg_pytorch_amd.parallel.ipc import ControlBlock
        envs, seg = 4096, 32
        DONE_SLOT = ControlBlock.TOOL_SLOT0     # a free fixed slot of the control block: workers that have finished
        rcfg = default_config(**{**cfg, "gpu_rollout_envs": envs, "gpu_rollout_segment": seg, "env_backend": "native",
                                 "rollout_devices": list(range(1, world)), "replay_mem_size": 2_000_000,
                                 "num_steps_train": 10 ** 9})
        name = "d4pg_bench_roll_%s" % os.environ.get("MASTER_PORT", "0")
        out = {}
        if rank == 0:
            from d4pg_pytorch_amd.models.learner import WeightPublisher
            from d4pg_pytorch_amd.parallel.hip_ipc import SharedDeviceBuffer, write_handle
            from d4pg_pytorch_amd.parallel.row_mailbox import RowMailbox
            from d4pg_pytorch_amd.parallel.ipc import cleanup_names
            cleanup_names([name])               # a segment an earlier, killed run on this port left behind
            ctrl = ControlBlock.create(n_workers=world + 2, name=name)
            core_r = build_core(rcfg, device)
            rp = HbmReplay(rcfg, device, core_r.dev_state)
            core_r.attach_replay(rp)
            rp.insert(synthetic_rows(100_000, S, A, rp.row_ld, gamma_n, device, seed=5))
            shared = SharedDeviceBuffer.create(4 * core_r.layout.actor.numel, device)
            write_handle(ctrl.name, shared.handle, shared.numel, dev_index)
            pub = WeightPublisher(core_r, None, ctrl, device, shared)
            mailbox = RowMailbox.create(ctrl.name, world - 1, envs * seg, rp.row_ld, device)
            pub.publish()
            pub.poll(block=True)
            dist.barrier()
            many = core_r.update_from_replay_many
            for _ in range(4):
                many(32)
            torch.cuda.synchronize(device)
            rows0, t0, done, last_pub = mailbox.rows_received, time.perf_counter(), 0, 0
            while time.perf_counter() - t0 < seconds:
                done += many(32)
                mailbox.poll(rp)
                if done - last_pub >= 100:
                    pub.publish()
                    last_pub = done
                pub.poll()
            torch.cuda.synchronize(device)
            el = time.perf_counter() - t0
            pub.poll(block=True)
            out = {"env_steps_per_sec": round((mailbox.rows_received - rows0) / el, 1),
                   "learner_updates_per_sec_beside_rollouts": round(done / el, 2),
                   "rollout_workers": world - 1, "rollout_envs_per_worker": envs, "rollout_segment_steps": seg,
                   "rollout_env": env + " (native device simulator)"}
            ctrl.request_stop()
            # the workers report in through the control block, polled from Python: a worker that died must not leave
            # this rank inside a collective it can never leave (the headline of the run is still to be printed)
            t_end = time.perf_counter() + 30.0
            while ctrl.load(DONE_SLOT) < world - 1 and time.perf_counter() < t_end:
                time.sleep(0.01)
            if ctrl.load(DONE_SLOT) < world - 1:
                out["rollout_workers_error"] = "%d of %d workers did not report back" % (world - 1 - ctrl.load(DONE_SLOT), world - 1)
            mailbox.close()
            shared.close()
            for suffix in (".hipipc", ".rows.hipipc"):
                try:
                    os.unlink(os.path.join("/dev/shm", ctrl.name.strip("/") + suffix))
                except OSError:
                    pass
            ctrl.close()
        else:
            from d4pg_pytorch_amd.actors.rollout_worker import run_rollout_worker
            dist.barrier()                       # the control block, the weight handle and the mailbox exist
            ctrl = ControlBlock.attach(name)
            try:
                run_rollout_worker(rcfg, ctrl, rank - 1, dev_index, hb_slot=rank)
            finally:
                ctrl.add(DONE_SLOT, 1)
                ctrl.close()
        return out

    extra = {}
    dp_mode = "none"
    metrics_primary = None

    def emit(core, dt, dp_mode, metrics_primary):
        """Rank 0 prints the one JSON line of the run."""
        metrics = metrics_primary if metrics_primary is not None else core.read_metrics()
        finite = all(map(lambda v: v == v and abs(v) != float("inf"), (metrics["value_loss"], metrics["policy_loss"])))
        if rank == 0:
            sync_ups = args.steps / dt
            value = sync_ups * world
            out = {
                "metric": "learner updates/sec",
                "value": round(value, 2),
                "unit": "batch-%d learner updates/s (sum over ranks)" % args.batch,
                "n_gpus": world,
                "steps": args.steps,
                "warmup": args.warmup,
                "repeats": repeats,
                "ms_per_step": round(1000.0 * dt / args.steps, 5),
                "higher_is_better": True,
                "scaling": "weak",
                "vs_baseline": None,
                "dtype": args.dtype if not args.baseline else "fp32",
                "data": "synthetic transitions (random-init weights, %d-row HBM replay prefilled with %d rows)"
                        % (args.replay_size, min(args.prefill, args.replay_size)),
                "sync_updates_per_sec": round(sync_ups, 2),
                "transitions_per_sec": round(sync_ups * args.batch * world, 1),
                "losses_finite": bool(finite),
                "value_loss": metrics["value_loss"],
                "policy_loss": metrics["policy_loss"],
                "implementation": "eager-torch-unfused" if args.baseline else
                                  (type(core).__name__ + ":" +
                                   {"HipCoreF32": "hip-fused-rowlocal-fp32-mfma", "HipCoreBf16": "hip-fused-rowlocal-bf16",
                                    "HipCore": "hip-per-layer-fp32-mfma"}.get(type(core).__name__, "?")
                                   + ("-16x16x4" if getattr(core, "wide", False) else "")
                                   + ("" if args.no_graph else "+hipgraph")),
                "config": {"model": "%s %s (S=%d, A=%d, H=%d%s)" % (
                               env, args.model.upper(), S, A, args.hidden, ", 51 atoms" if args.model == "d4pg" else ""),
                           "global_batch": args.batch * world, "seq_len": 1,
                           "n_step": 5, "replay": "per-hbm-sumtree" if args.per else "uniform-hbm",
                           "parallelism": "dp%d" % world + {
                               "none": "", "replicated": " (sharded HBM replay, all-gathered batches, replicated update)",
                               "grad_allreduce": " (gradient all-reduce)"}[dp_mode]},
            }
            out.update(extra)
            if "single_gpu_global_batch_value" in extra:
                out["speedup_vs_single_gpu_global_batch"] = round(value / extra["single_gpu_global_batch_value"], 3)
            if dp_mode == "grad_allreduce":
                out["grad_allreduce_in_graph"] = bool(getattr(core, "_sync_in_graph", lambda: False)())
            print(json.dumps(out), flush=True)

    if args.baseline:
        from d4pg_pytorch_amd.models.core import Batch, TorchCore
        core = TorchCore(cfg, device)
        attach_data_parallel(core, comm)
        dp_mode = "grad_allreduce" if world > 1 else "none"
        rows = synthetic_rows(args.prefill, S, A, 2 * S + A + 3, gamma_n, device, seed=rank + 1)

        def one_step():
            # reference-equivalent data path: uniform host-free sampling with torch ops
            idx = torch.randint(0, rows.shape[0], (args.batch,), device=device)
            r = rows[idx]
            b = Batch(r[:, :S], r[:, S:S + A], r[:, S + A], r[:, S + A + 3:S + A + 3 + S], r[:, S + A + 1],
                      r[:, S + A + 2], None, idx)
            core.update(b)
        dt = timed(one_step)
    else:
        from d4pg_pytorch_amd.models.core import build_core
        from d4pg_pytorch_amd.parallel.replicated import attach_replicated
        from d4pg_pytorch_amd.replay.hbm_replay import HbmReplay

        def run_grad_allreduce():
            core = build_core(cfg, device)
            replay = HbmReplay(cfg, device, core.dev_state)
            core.attach_replay(replay)
            attach_data_parallel(core, comm)
            fill(replay)
            return core, timed(core.update_from_replay, getattr(core, "update_from_replay_many", None))

        def run_replicated():
            core = build_core(dict(cfg, batch_size=args.batch * world), device)
            replay = HbmReplay(cfg, device, core.dev_state)
            fill(replay)
            dp = attach_replicated(core, replay, comm, args.batch, chunk=int(cfg["sample_ahead"]))
            t = timed(dp.update, dp.update_many)
            drift = dp.check_consistency()
            assert drift == 0.0, f"learner replicas diverged (checksum difference {drift})"
            return core, t

        if world > 1 or args.dp_selftest:
            # What ONE GPU does alone, no collectives: (a) the per-rank configuration (batch B) and (b) the
            # job's GLOBAL batch (world x B rows per update).  (b) is the honest denominator for a
            # data-parallel learner: N GPUs are worth having for the learner only if the N-GPU job beats it.
            real_comm, comm = comm, LocalComm()
            try:
                _, dt_1 = run_grad_allreduce()
                extra["single_gpu_same_config_value"] = round(args.steps / dt_1, 2)
            except Exception as exc:
                extra["single_gpu_same_config_error"] = repr(exc)[:200]
            torch.cuda.empty_cache()
            if world > 1:
                cfg_rank, cfg = cfg, dict(cfg, batch_size=args.batch * world)
                try:
                    _, dt_g = run_grad_allreduce()
                    extra["single_gpu_global_batch_value"] = round(args.steps / dt_g * world, 2)
                    extra["single_gpu_global_batch_ms_per_step"] = round(1000.0 * dt_g / args.steps, 5)
                except Exception as exc:
                    extra["single_gpu_global_batch_error"] = repr(exc)[:200]
                cfg = cfg_rank
                torch.cuda.empty_cache()
            comm = real_comm
        if world == 1 and not args.dp_selftest:
            core, dt = run_grad_allreduce()
            if args.dtype == "fp32" and not args.no_bf16 and not args.layers:
                # the same configuration on the bf16-operand pipeline, for reference next to the headline
                metrics_primary = core.read_metrics()
                cfg_main, cfg = cfg, dict(cfg, compute_dtype="bf16")
                try:
                    _, dt_b = run_grad_allreduce()
                    extra["bf16_value"] = round(args.steps / dt_b, 2)
                    extra["bf16_ms_per_step"] = round(1000.0 * dt_b / args.steps, 5)
                except Exception as exc:
                    extra["bf16_error"] = repr(exc)[:200]
                cfg = cfg_main
                # ... and the reference's op sequence in eager PyTorch on this GPU (the reference publishes no
                # numbers: this is the same-hardware yardstick).  At most 200 updates, it is ~30x slower.
                try:
                    from d4pg_pytorch_amd.models.core import Batch, TorchCore
                    tcore = TorchCore(cfg, device)
                    trows = synthetic_rows(min(args.prefill, 50_000), S, A, 2 * S + A + 3, gamma_n, device, seed=7)

                    def eager_step():
                        idx = torch.randint(0, trows.shape[0], (args.batch,), device=device)
                        r = trows[idx]
                        tcore.update(Batch(r[:, :S], r[:, S:S + A], r[:, S + A], r[:, S + A + 3:S + A + 3 + S],
                                           r[:, S + A + 1], r[:, S + A + 2], None, idx))
                    n_e = max(20, min(200, args.steps))
                    for _ in range(10):
                        eager_step()
                    torch.cuda.synchronize(device)
                    t0 = time.perf_counter()
                    for _ in range(n_e):
                        eager_step()
                    torch.cuda.synchronize(device)
                    dt_e = (time.perf_counter() - t0) / n_e
                    extra["eager_torch_value"] = round(1.0 / dt_e, 2)
                    extra["eager_torch_ms_per_step"] = round(1000.0 * dt_e, 5)
                    extra["speedup_vs_eager_torch"] = round(dt_e / (dt / args.steps), 2)
                except Exception as exc:
                    extra["eager_torch_error"] = repr(exc)[:200]
            if (args.dtype == "fp32" and not args.no_extras and not args.layers and not args.per
                    and args.model == "d4pg" and env == "LunarLanderContinuous-v2" and args.batch == 256):
                # the other headline configurations of the reference's algorithm family, fp32, same run (short):
                # prioritised replay, the scalar critic (D3PG / DDPG) and the Ant-shape batch-4096 stress
                metrics_primary = metrics_primary if metrics_primary is not None else core.read_metrics()
                for key, ov, n_q, w_q in (("per_value", {"replay_memory_prioritized": 1}, 1600, 320),
                                          ("d3pg_value", {"model": "d3pg"}, 1600, 320)):
                    try:
                        extra[key] = round(quick_value(ov, n_q, w_q), 2)
                    except Exception as exc:
                        extra[key + "_error"] = repr(exc)[:200]
                try:
                    ups = quick_value({"env": "Ant-v2", "batch_size": 4096, "v_min": 0.0, "v_max": 4000.0,
                                       "state_dim": 111, "action_dim": 8}, 320, 96)
                    extra["ant4096_fp32_ms_per_step"] = round(1000.0 / ups, 5)
                    extra["ant4096_fp32_transitions_per_sec"] = round(ups * 4096, 1)
                    ups3 = quick_value({"env": "Ant-v2", "batch_size": 4096, "model": "d3pg", "state_dim": 111,
                                        "action_dim": 8}, 320, 96)      # the scalar critic on the same shape
                    extra["ant4096_d3pg_fp32_ms_per_step"] = round(1000.0 / ups3, 5)
                except Exception as exc:
                    extra["ant4096_fp32_error"] = repr(exc)[:200]
                # "env steps/sec", the second stated metric, on one GPU: 4096 device-resident environments of this shape's
                # task (policy with OU noise + dynamics + n-step rows, two launches per environment step, one captured
                # graph per 32-step segment) feeding an HBM replay, nothing else on the GPU
                try:
                    from d4pg_pytorch_amd.actors.gpu_rollout import DeviceRollout
                    from d4pg_pytorch_amd.envs.device_envs import make_device_env
                    ecfg = default_config(**{**cfg, "env_backend": "native", "gpu_rollout_envs": 4096, "replay_mem_size": 1_000_000})
                    ecore = build_core(ecfg, device)
                    erp = HbmReplay(ecfg, device, ecore.dev_state)
                    ero = DeviceRollout(ecfg, make_device_env(ecfg, 4096, device, 11), erp, device, segment=32, seed=12)
                    ero.refresh_weights(ecore.online.flat.data_ptr(), ecore.layout.actor.base)
                    for _ in range(3):
                        ero.collect()
                    torch.cuda.synchronize(device)
                    t0 = time.perf_counter()
                    n_env = sum(ero.collect() for _ in range(20))
                    torch.cuda.synchronize(device)
                    extra["env_steps_per_sec"] = round(n_env / (time.perf_counter() - t0), 1)
                    extra["env_steps_config"] = "4096 device environments of %s (native simulator), 32-step segments" % env
                    del ero, erp, ecore
                    torch.cuda.empty_cache()
                except Exception as exc:
                    extra["env_steps_error"] = repr(exc)[:200]
                # "episodes to reward target", the third stated metric, on the task that needs no data file: the built-in
                # pendulum entirely on the GPU (8 device environments feed the HBM replay at 8 steps per update, the
                # greedy TARGET policy is evaluated on 64 fresh episodes every 1000 updates; a random policy scores
                # about -1200, -300 is a pendulum that swings up and stays up)
                try:
                    from d4pg_pytorch_amd.utils.learning import train_on_device
                    r = train_on_device("d4pg", "fp32", updates=20_000, seed=103, eval_every=1000, device=device)
                    hit = r["first_eval_step_above_-300"]
                    extra["pendulum_target_return"] = -300.0
                    extra["pendulum_updates_to_target"] = hit
                    extra["pendulum_episodes_to_target"] = None if hit is None else int((2048 + 8 * hit) // 200)
                    extra["pendulum_final_return"] = r["final_return"]
                    extra["pendulum_learn_wall_s"] = r["wall_s"]
                except Exception as exc:
                    extra["pendulum_learn_error"] = repr(exc)[:200]
        elif args.dp_mode != "both":
            dp_mode = args.dp_mode
            core, dt = run_replicated() if dp_mode == "replicated" else run_grad_allreduce()
        else:
            # both modes are timed, the one dp_mode: auto picks on this machine first so that nothing the other run
            # does can affect it; `value` is that mode's, the faster of the two is reported next to it
            # (`faster_dp_mode`, `faster_dp_mode_value`: a job is free to set dp_mode)
            import threading
            # what dp_mode: auto picks on THIS machine: the framework times one all-reduce of the critic's gradient
            # bucket on the live communicator (parallel/comm.py::probe_allreduce_us) and compares the two modes' costs
            from d4pg_pytorch_amd.models.nets import NetLayout
            from d4pg_pytorch_amd.parallel.comm import probe_allreduce_us
            bucket = NetLayout.build(S + A, cfg["dense_size"], cfg["num_atoms"] if args.model == "d4pg" else 1).numel
            try:
                ar_us = probe_allreduce_us(comm, bucket, device)
                extra["allreduce_probe_us"] = round(ar_us, 2)
                extra["allreduce_probe_bytes"] = bucket * 4
                if args.dp_selftest:      # a world of one: RCCL launches nothing, the captured form is known to run
                    extra["allreduce_probe_us_in_graph"] = round(probe_allreduce_us(comm, bucket, device, in_graph=True), 2)
            except Exception as exc:
                ar_us = None
                extra["allreduce_probe_error"] = repr(exc)[:200]
            auto = resolve_dp_mode("auto", args.batch, world, args.dtype, allreduce_us=ar_us)
            runners = {"replicated": run_replicated, "grad_allreduce": run_grad_allreduce}
            other = "grad_allreduce" if auto == "replicated" else "replicated"
            # Neither mode has ever met more than one GPU before a job like this one runs, so the first mode is
            # bounded as well: if its collectives never complete on this machine, every rank says so and leaves
            # with a non-zero status instead of hanging the job (nothing valid could be reported).
            def first_mode_hung():
                if rank == 0:
                    print(json.dumps({"metric": "learner updates/sec", "value": None, "n_gpus": world,
                                      "error": f"dp mode '{auto}' did not finish on {world} GPUs"}), flush=True)
                os._exit(3)
            first_dog = threading.Timer(float(os.environ.get("D4PG_BENCH_MODE_TIMEOUT_S", "420")), first_mode_hung)
            first_dog.daemon = True
            first_dog.start()
            core, dt = runners[auto]()
            first_dog.cancel()
            metrics_primary = core.read_metrics()
            dp_mode = auto
            extra["dp_mode_auto"] = auto
            extra[auto + "_value"] = round(args.steps / dt * world, 2)
            extra[auto + "_ms_per_step"] = round(1000.0 * dt / args.steps, 5)
            torch.cuda.empty_cache()
            # The second mode must not cost the first its result: if it has not finished after a generous
            # multiple of the first mode's time (a collective that never completes on this machine), every rank
            # reports what it has and leaves.

            def give_up():
                extra[other + "_error"] = "timed out"
                extra["dp_mode_selected"] = dp_mode + " (dp_mode: auto; the other mode did not finish)"
                if rank == 0:
                    emit(core, dt, dp_mode, metrics_primary)
                os._exit(4)          # the line above is valid, the job is not: a collective never completed
            watchdog = threading.Timer(max(180.0, 40.0 * dt * (repeats + 1) + 120.0), give_up)
            watchdog.daemon = True
            watchdog.start()
            try:
                core_o, dt_o = runners[other]()
                extra[other + "_value"] = round(args.steps / dt_o * world, 2)
                extra[other + "_ms_per_step"] = round(1000.0 * dt_o / args.steps, 5)
                # `value` stays what dp_mode: auto runs; the faster of the two timed modes is reported beside it
                faster, dt_f = (other, dt_o) if dt_o < dt else (auto, dt)
                extra["faster_dp_mode"] = faster
                extra["faster_dp_mode_value"] = round(args.steps / dt_f * world, 2)
            except Exception as exc:
                extra[other + "_error"] = repr(exc)[:200]
            finally:
                watchdog.cancel()
            extra["dp_mode_selected"] = dp_mode + " (dp_mode: auto)"

    if world > 1 and not args.no_extras and not args.baseline and not args.dp_selftest:
        import threading
        metrics_primary = metrics_primary if metrics_primary is not None else core.read_metrics()

        def extras_hung():          # the headline must not be lost to an extra that never returns
            extra["rollout_workers_error"] = "timed out"
            if rank == 0:
                emit(core, dt, dp_mode, metrics_primary)
            os._exit(0)

        def extras_terminated(signum, frame):      # the launcher ends every rank when one of them has died
            extra["rollout_workers_error"] = "terminated by signal %d while the rollout workers ran" % signum
            if rank == 0:
                emit(core, dt, dp_mode, metrics_primary)
            os._exit(0 if rank == 0 else 1)
        import signal
        old_term = signal.signal(signal.SIGTERM, extras_terminated)
        import faulthandler
        limit = float(os.environ.get("D4PG_BENCH_EXTRAS_TIMEOUT_S", "150"))
        dog = threading.Timer(limit, extras_hung)
        dog.daemon = True
        dog.start()
        faulthandler.dump_traceback_later(max(1.0, limit - 5.0), exit=False)     # a hang names its own stack on stderr
        try:
            if getattr(comm, "backend", "") == "nccl" and "allreduce_probe_bytes" in extra:
                # the same all-reduce timed as nodes of a graph (the way the gradient exchange runs it), for the record
                from d4pg_pytorch_amd.parallel.comm import probe_allreduce_us
                extra["allreduce_probe_us_in_graph"] = round(
                    probe_allreduce_us(comm, extra["allreduce_probe_bytes"] // 4, device, in_graph=True), 2)
        except Exception as exc:
            extra["allreduce_probe_in_graph_error"] = repr(exc)[:200]
        try:
            extra.update(rollout_workers_extra())
        except Exception as exc:
            extra["rollout_workers_error"] = repr(exc)[:300]
        finally:
            faulthandler.cancel_dump_traceback_later()
            dog.cancel()
            signal.signal(signal.SIGTERM, old_term)
    emit(core, dt, dp_mode, metrics_primary)
    comm.close()


if __name__ == "__main__":
    main()
