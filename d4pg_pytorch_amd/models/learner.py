"""Learner process: owns the networks, drives updates, publishes weights.

Process-level counterpart of the reference's ``LearnerD4PG`` / ``LearnerD3PG``
(``models/d4pg/d4pg.py:15-171``, ``models/d3pg/d3pg.py:14-128``): construct from
the config, ``run()`` until ``num_steps_train`` updates are done, log
``learner/policy_loss``, ``learner/value_loss`` and the update timing, publish
actor weights every ``weight_publish_interval`` updates, then raise the stop flag.

Two data paths:

* ``run_hbm``  — GPU topology.  The replay shard and its priority tree live in
  this process on the learner's GPU; explorer rings are drained in chunks
  through pinned staging buffers and async H2D copies; sampling, the update and
  the priority write-back are one HIP graph (``models/hip_core.py``).
* ``run_host`` — CPU topology.  A sampler process owns the replay and feeds
  ready batches through a queue, as in the reference.

Losses are accumulated on the device and read back every ``metrics_interval``
updates (the reference forces two device syncs per update with ``.item()``,
``models/d4pg/d4pg.py:150-151``).
"""
from __future__ import annotations

import os
import queue
import time
from typing import Dict, List, Optional

import numpy as np
import torch

from ..parallel.comm import Comm, LocalComm, attach_data_parallel, resolve_dp_mode
from ..parallel.ipc import ControlBlock, ShmRing, WeightSlab
from ..utils.checkpoint import load_learner_bundle, save_learner_bundle
from ..utils.logger import Logger
from .core import Batch, build_core


class RingIngestor:
    """Moves transition rows from explorer rings into an HBM replay shard.

    Zero-copy (the default on a GPU): the rings' shared-memory data areas are registered with the HIP runtime
    once, so a ``poll`` is, per ring, one ``ring_peek`` (no copy) and one asynchronous DMA from the ring's own
    memory into a device staging block, followed by one ``replay_insert`` launch for all rings.  The rows stay
    owned by their ring until the DMA's event has completed; only then does ``ring_advance`` hand the slots back
    to the explorer (polled, the learner never waits).

    Fallback (registration refused, or ``zero_copy=False``): every ring is drained with ``ring_pop`` into one
    pinned staging buffer (three rotate) and copied from there."""

    def __init__(self, rings: List[ShmRing], replay, device, chunk: int, n_buffers: int = 3, zero_copy: bool = True):
        self.rings, self.replay, self.device = rings, replay, device
        self.chunk = int(chunk)
        self.cap = max(self.chunk, min(8 * self.chunk, 16384))      # rows per poll over all rings
        self.staged = [torch.zeros(self.cap, replay.row_ld, device=device) for _ in range(n_buffers)]
        self._next = 0
        self.rows_ingested = 0
        self.zero_copy = bool(zero_copy) and bool(rings) and self._register()
        if self.zero_copy:
            self.ahead = [0] * len(rings)          # rows peeked and in flight, per ring
            self.inflight: List = []               # (event, [(ring index, rows), ...]) oldest first
        else:
            self.pinned = [torch.zeros(self.cap, replay.row_ld).pin_memory() for _ in range(n_buffers)]
            self.pinned_np = [p.numpy() for p in self.pinned]
            self.events = [None] * n_buffers

    def _register(self) -> bool:
        from ..ops import hip as H
        done = []
        for r in self.rings:
            if H.lib().d4pg_host_register(r.data_ptr, r.data_bytes) != 0:
                for q in done:
                    H.lib().d4pg_host_unregister(q.data_ptr)
                print("[learner] ring memory could not be registered for DMA; staging through pinned buffers", flush=True)
                return False
            done.append(r)
        return True

    def close(self) -> None:
        if self.zero_copy:
            from ..ops import hip as H
            self.release(block=True)
            for r in self.rings:
                H.lib().d4pg_host_unregister(r.data_ptr)
            self.zero_copy = False

    def release(self, block: bool = False) -> None:
        """Give the slots of completed DMAs back to their rings."""
        while self.inflight:
            ev, spans = self.inflight[0]
            if block:
                ev.synchronize()
            elif not ev.query():
                return
            for k, n in spans:
                self.rings[k].advance(n)
                self.ahead[k] -= n
            self.inflight.pop(0)

    def _poll_zero_copy(self, per_ring: int) -> int:
        from ..ops import hip as H
        self.release()
        i = self._next
        dst, row_bytes = self.staged[i], self.replay.row_ld * 4
        stream = H.stream_ptr()
        n, spans = 0, []
        for k, ring in enumerate(self.rings):
            taken = 0
            while taken < per_ring and n < self.cap:        # at most two spans: the readable part may wrap
                off, m = ring.peek(min(per_ring - taken, self.cap - n), ahead=self.ahead[k])
                if m == 0:
                    break
                H._check(H.lib().d4pg_memcpy_h2d_async(dst[n:].data_ptr(), ring.data_ptr + off * row_bytes,
                                                       m * row_bytes, stream), "d4pg_memcpy_h2d_async")
                self.ahead[k] += m
                spans.append((k, m))
                taken += m
                n += m
        if n == 0:
            return 0
        # staged[i] is rewritten n_buffers polls later, on the same stream, i.e. behind this insert
        self.replay.insert(dst[:n])
        ev = torch.cuda.Event()
        ev.record()
        self.inflight.append((ev, spans))
        self._next = (i + 1) % len(self.staged)
        self.rows_ingested += n
        return n

    def poll(self, max_chunks_per_ring: int = 4) -> int:
        per_ring = max_chunks_per_ring * self.chunk
        if self.zero_copy:
            return self._poll_zero_copy(per_ring)
        i = self._next
        if self.events[i] is not None:
            self.events[i].synchronize()          # the previous H2D out of this pinned buffer is done
        buf = self.pinned_np[i]
        n = 0
        for ring in self.rings:
            room = min(per_ring, self.cap - n)
            if room <= 0:
                break
            n += ring.pop(room, out=buf[n: n + room]).shape[0]
        if n == 0:
            return 0
        self.staged[i][:n].copy_(self.pinned[i][:n], non_blocking=True)
        self.replay.insert(self.staged[i][:n])
        ev = torch.cuda.Event()
        ev.record()
        self.events[i] = ev
        self._next = (i + 1) % len(self.pinned)
        self.rows_ingested += n
        return n


class WeightPublisher:
    """Publication of ``[online actor | target actor]`` that never makes the learner wait for the GPU.

    The reference pushes a CPU copy of the weights into a queue every 100 updates, which synchronises the
    device each time (``models/d4pg/d4pg.py:141-146``).  Here a publication is three enqueues:

    1. on the update stream, two device-to-device copies take a consistent snapshot into one of two staging
       slots (the slots are a HIP-IPC buffer when GPU agents follow the learner in place,
       ``parallel/hip_ipc.py``);
    2. a side stream, ordered behind that snapshot by an event, copies the slot into pinned host memory;
    3. :meth:`poll` (called from the learner's loop) sees the copy's event complete and only then writes the
       seq-locked slab for CPU agents and raises the version GPU agents watch.

    A slot is rewritten two publications later; the update stream waits (on the device) for the event of the
    copy that last read it, which has long completed by then."""

    def __init__(self, core, slab: Optional[WeightSlab], ctrl: ControlBlock, device, shared=None):
        self.core, self.slab, self.ctrl, self.device = core, slab, ctrl, torch.device(device)
        self.n = core.layout.actor.numel
        self.cuda = self.device.type == "cuda"
        self.shared = shared
        self.issued = self.completed = 0
        if self.cuda:
            if shared is not None:
                assert shared.numel == 4 * self.n
                self.stage = shared.tensor.view(2, 2 * self.n)
            else:
                self.stage = torch.zeros(2, 2 * self.n, device=self.device)
            self.pinned = [torch.zeros(2 * self.n).pin_memory() for _ in range(2)]
            self.side = torch.cuda.Stream(device=self.device)
            self.pending = [None, None]            # per slot: (event of the D2H copy, learner step)
            self.read_done = [None, None]          # per slot: event after which the slot may be rewritten
        else:
            self.host = torch.zeros(2 * self.n)

    def publish(self) -> None:
        n, core = self.n, self.core
        step = int(core.step)
        if not self.cuda:
            with torch.no_grad():
                self.host[:n].copy_(core.online.actor_flat().detach())
                self.host[n:].copy_(core.target.actor_flat().detach())
            if self.slab is not None:
                self.slab.publish(self.host.numpy(), step)
            return
        slot = self.issued % 2
        if self.pending[slot] is not None:         # two publications in flight: finish the older one first
            self._complete(slot, block=True)
        main = torch.cuda.current_stream(self.device)
        if self.read_done[slot] is not None:
            main.wait_event(self.read_done[slot])
        self.ctrl.store(ControlBlock.GPU_WEIGHT_ISSUED, self.issued + 1)     # before the slot is touched
        with torch.no_grad():
            self.stage[slot, :n].copy_(core.online.actor_flat().detach(), non_blocking=True)
            self.stage[slot, n:].copy_(core.target.actor_flat().detach(), non_blocking=True)
        snap = torch.cuda.Event()
        snap.record(main)
        self.side.wait_event(snap)
        with torch.cuda.stream(self.side):
            self.pinned[slot].copy_(self.stage[slot], non_blocking=True)
            done = torch.cuda.Event()
            done.record(self.side)
        self.pending[slot] = (done, step)
        self.read_done[slot] = done
        self.issued += 1

    def _complete(self, slot: int, block: bool) -> bool:
        done, step = self.pending[slot]
        if block:
            done.synchronize()
        elif not done.query():
            return False
        if self.slab is not None:
            self.slab.publish(self.pinned[slot].numpy(), step)
        self.pending[slot] = None
        self.completed += 1
        self.ctrl.store(ControlBlock.GPU_WEIGHT_STEP, step)
        self.ctrl.store(ControlBlock.GPU_WEIGHT_VERSION, self.completed)     # newest snapshot: slot (version - 1) % 2
        return True

    def poll(self, block: bool = False) -> int:
        """Hand finished snapshots to the agents, oldest first.  Returns how many completed."""
        if not self.cuda:
            return 0
        n = 0
        while self.completed < self.issued:
            if not self._complete(self.completed % 2, block):
                break
            n += 1
        return n


class Learner:
    """Policy and value network update routine (process level)."""

    expected_models = ("d4pg", "d3pg", "ddpg")

    def __init__(self, config: Dict, ctrl: ControlBlock, slab: Optional[WeightSlab], log_dir: str = "",
                 rank: int = 0, comm: Optional[Comm] = None, device=None):
        if config["model"] not in self.expected_models:
            raise ValueError(f"{type(self).__name__} cannot run model '{config['model']}'")
        self.config, self.ctrl, self.slab, self.rank = config, ctrl, slab, rank
        self.comm = comm or LocalComm()
        self.device = torch.device(device if device is not None else config["device"])
        self.log_dir = log_dir
        self.num_train_steps = int(config["num_steps_train"])
        # replicated data parallelism: every rank runs the update on the global batch
        probe_us = None
        if self.comm.world > 1 and config.get("dp_mode", "auto") == "auto" and self.device.type == "cuda":
            # what an all-reduce of the critic's gradient bucket costs on THIS machine's communicator
            from ..parallel.comm import probe_allreduce_us
            from .nets import NetLayout
            bucket = NetLayout.build(int(config["state_dim"]) + int(config["action_dim"]), int(config["dense_size"]),
                                     int(config["num_atoms"]) if config["model"] == "d4pg" else 1).numel
            probe_us = probe_allreduce_us(self.comm, bucket, self.device)
            if rank == 0:
                print(f"[learner] all-reduce of a {bucket * 4 >> 10} KiB gradient bucket over {self.comm.world} ranks: "
                      f"{probe_us:.1f} us", flush=True)
        dp_mode = resolve_dp_mode(config.get("dp_mode", "auto"), int(config["batch_size"]), self.comm.world,
                                  str(config.get("compute_dtype", "auto")), allreduce_us=probe_us)
        self.dp_mode, self.allreduce_probe_us = dp_mode, probe_us
        self.replicated = (self.comm.world > 1 and dp_mode == "replicated"
                           and self.device.type == "cuda" and config.get("fused_learner", "auto") != "torch")
        core_cfg = dict(config, batch_size=int(config["batch_size"]) * self.comm.world) if self.replicated else config
        self.core = build_core(core_cfg, self.device, seed=config["random_seed"])
        if config.get("resume"):
            self.core.load_state_bundle(load_learner_bundle(config["resume"]))
            print(f"[learner {rank}] resumed from {config['resume']} at step {self.core.step}", flush=True)
        if not self.replicated:
            attach_data_parallel(self.core, self.comm)
        self.logger = Logger(os.path.join(log_dir, "learner"), config.get("log_backend", "tb")) \
            if (log_dir and rank == 0) else None
        self.timing_tag = "learner/learner_update_timing" if config["model"] == "d4pg" else "learner/update_time"
        self.hb_slot = rank                      # heartbeat slot of this learner rank (engine.py: learners come first)
        self._shared = None
        if (rank == 0 and self.device.type == "cuda" and config.get("share_weights_ipc", True)
                and (str(config.get("exploiter_device", "cpu")).startswith("cuda") or config.get("rollout_devices"))):
            self._shared = self._export_weights()
        self.publisher = WeightPublisher(self.core, slab, ctrl, self.device, self._shared) if rank == 0 else None
        self.rollout = None
        self._last_metrics_t = time.time()
        self._last_metrics_step = self.core.step
        self._last_env_steps = 0

    # ----------------------------------------------------------------- weights
    def _export_weights(self):
        """A HIP-IPC buffer of two ``[online actor | target actor]`` slots for GPU agents on this machine; its
        handle is left next to the control block.  ``None`` when the runtime refuses (agents then use the slab)."""
        try:
            from ..parallel.hip_ipc import SharedDeviceBuffer, write_handle
            shared = SharedDeviceBuffer.create(4 * self.core.layout.actor.numel, self.device)
            write_handle(self.ctrl.name, shared.handle, shared.numel, self.device.index or 0)
            return shared
        except Exception as exc:      # pragma: no cover - depends on the host's IPC support
            print(f"[learner] device weight sharing unavailable ({exc}); agents read the host slab", flush=True)
            return None

    def publish_weights(self, wait: bool = False) -> None:
        """[online actor | target actor] -> every agent (seq-locked slab; HIP-IPC slots for GPU agents).
        Enqueue only unless ``wait``: the hand-over completes in :meth:`WeightPublisher.poll`."""
        if self.publisher is None:
            return
        self.publisher.publish()
        if wait:
            self.publisher.poll(block=True)

    def _beat(self) -> None:
        self.ctrl.heartbeat(self.hb_slot, int(time.time() * 1000))
        if self.publisher is not None:
            self.publisher.poll()

    # ----------------------------------------------------------------- logging
    def _log_metrics(self, extra: Optional[Dict[str, float]] = None) -> None:
        m = self.core.read_metrics()
        now = time.time()
        steps = max(1, self.core.step - self._last_metrics_step)
        dt = max(1e-9, now - self._last_metrics_t)
        env_steps = self.ctrl.env_steps
        if self.logger is not None and m["count"] > 0:
            s = self.core.step
            self.logger.scalar_summary("learner/policy_loss", m["policy_loss"], s)
            self.logger.scalar_summary("learner/value_loss", m["value_loss"], s)
            self.logger.scalar_summary(self.timing_tag, dt / steps, s)
            self.logger.scalar_summary("perf/updates_per_s", steps / dt, s)
            self.logger.scalar_summary("perf/env_steps_per_s", (env_steps - self._last_env_steps) / dt, s)
            for k, v in (extra or {}).items():
                self.logger.scalar_summary(k, v, s)
        self.last_metrics = m
        self._last_metrics_t, self._last_metrics_step, self._last_env_steps = now, self.core.step, env_steps

    def _checkpoint(self, final: bool = False) -> Optional[str]:
        if self.rank != 0 or not self.log_dir:
            return None
        name = "checkpoint_final.pt" if final else f"checkpoint_{self.core.step}.pt"
        return save_learner_bundle(os.path.join(self.log_dir, "learner", name), self.core.state_bundle(), self.config)

    def _after_update(self, extra_fn=None, prev_step: Optional[int] = None) -> None:
        """Periodic work after one call that advanced the learner from ``prev_step`` to
        ``core.step`` (several updates when a whole chunk was replayed as one graph): every
        interval whose multiple was reached or crossed fires once."""
        cfg, step = self.config, self.core.step
        prev = step - 1 if prev_step is None else prev_step

        def crossed(interval: int) -> bool:
            return interval > 0 and step // interval != prev // interval

        self.ctrl.store(ControlBlock.UPDATE_STEP, step)
        if crossed(int(cfg["weight_publish_interval"])):
            self.publish_weights()
        if crossed(int(cfg["metrics_interval"])):
            self._log_metrics(extra_fn() if extra_fn else None)
        self._beat()
        if crossed(1000) and self.rank == 0:
            print("Training step ", step - step % 1000, flush=True)
        if crossed(int(cfg.get("learner_checkpoint_interval") or 0)):
            self._checkpoint()

    def _throttle(self) -> bool:
        """Optional updates-per-env-step cap.  Returns True if the learner should wait."""
        ratio = self.config.get("updates_per_env_step")
        if not ratio:
            return False
        return self.core.step >= float(ratio) * max(1, self.ctrl.env_steps)

    def _finish(self) -> None:
        self.publish_weights(wait=True)
        self._log_metrics()
        self.ctrl.store(ControlBlock.UPDATE_STEP, self.core.step)
        self._checkpoint(final=True)
        self.ctrl.request_stop()
        if self.logger is not None:
            self.logger.close()
        print("Exit learner.", flush=True)

    # ----------------------------------------------------------------- hbm path
    def run_hbm(self, rings: List[ShmRing]) -> None:
        from ..replay.hbm_replay import HbmReplay
        cfg, core, ctrl = self.config, self.core, self.ctrl
        replay = HbmReplay(cfg, self.device, core.dev_state)
        dp = None
        if self.replicated:
            from ..parallel.replicated import attach_replicated
            # prioritised replay: priorities lag by one to two chunks, so keep the chunks short
            chunk = int(cfg["sample_ahead"]) if not replay.prioritized else min(8, int(cfg["sample_ahead"]))
            dp = attach_replicated(core, replay, self.comm, int(cfg["batch_size"]), chunk=max(1, chunk))
        else:
            core.attach_replay(replay)
        self.replay = replay
        ingest = RingIngestor(rings, replay, self.device, cfg["ingest_chunk"], zero_copy=bool(cfg.get("ring_dma", True)))
        min_rows = max(int(cfg["min_replay_size"]), int(cfg["batch_size"]))
        ingest_every = max(1, int(cfg.get("ingest_interval", 32)))     # updates between two ring drains
        self.rollout = self._make_rollout(replay)
        mailbox = self._make_mailbox(replay)
        rollout_every = max(1, int(cfg.get("gpu_rollout_interval", 32)))
        last_rollout = -rollout_every
        self.publish_weights(wait=True)
        ctrl.store(ControlBlock.LEARNER_ALIVE, 1)
        last_ingest = core.step
        if dp is not None:
            burst = dp.update_many
        else:
            burst = getattr(core, "update_from_replay_many", None) or (lambda n: core.update_from_replay() and 1)

        def extra():
            out = {"data_struct/replay_buffer": float(len(replay)),
                   "data_struct/replay_queue": float(sum(len(r) for r in rings)),
                   "data_struct/global_episode": float(ctrl.global_episode),
                   "data_struct/ring_dropped": float(sum(r.dropped for r in rings))}
            if self.rollout is not None:
                mean_ret, n_ep = self.rollout.episode_stats()       # the metrics read-back has synchronised already
                if n_ep:
                    ctrl.add(ControlBlock.GLOBAL_EPISODE, n_ep)
                    out["agent/reward_device_explorers"] = mean_ret
            return out

        def rollout_segment() -> None:
            n = self.rollout.collect_overlapped(core.online.flat.data_ptr(), core.layout.actor.base)
            ctrl.add(ControlBlock.ENV_STEPS, n)
            if dp is not None:
                dp.mark_inserted()

        # Several learners step in lockstep through their collectives: none may leave on its own.  The stop
        # flag is therefore agreed on (max over ranks of a control scalar) every `control_interval` updates.
        world = self.comm.world
        control_every = max(1, int(cfg.get("control_interval", 1024)))
        while core.step < self.num_train_steps and not (ctrl.stop and world == 1):
            if len(replay) < min_rows or self._throttle():
                moved = ingest.poll() + (mailbox.poll(replay) if mailbox is not None else 0)
                if moved and dp is not None:
                    dp.mark_inserted()
                if self.rollout is not None:
                    rollout_segment()
                    last_rollout = core.step
                    moved += 1
                if moved == 0:
                    time.sleep(0.0005)
                self._beat()
                if ctrl.stop:            # starved and told to stop: the job is being torn down
                    break
                continue
            prev = core.step
            if prev % ingest_every == 0 or prev // ingest_every != last_ingest // ingest_every:
                last_ingest = prev
                if ingest.poll() + (mailbox.poll(replay) if mailbox is not None else 0) and dp is not None:
                    dp.mark_inserted()
            if self.rollout is not None and prev - last_rollout >= rollout_every:
                last_rollout = prev
                rollout_segment()
            burst(self.num_train_steps - prev)       # one update, or a whole drawn-ahead chunk as one graph
            self._after_update(extra, prev)
            if world > 1 and prev // control_every != core.step // control_every:
                if self.comm.all_reduce_max_scalar(1.0 if ctrl.stop else 0.0) > 0.0:
                    break
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)
        ingest.close()
        if mailbox is not None:
            self.rows_from_workers = mailbox.rows_received
            print(f"[learner {self.rank}] {mailbox.rows_received} rows from {mailbox.n_workers} rollout worker(s)", flush=True)
            mailbox.close()
        if dp is not None:
            drift = dp.check_consistency()
            if drift != 0.0:
                raise RuntimeError(f"learner replicas diverged (checksum difference {drift})")
        if cfg.get("save_buffer_on_disk"):
            replay.dump(os.path.join(cfg["results_path"], "replay_buffer.pkl"))
        self._finish()

    def _make_mailbox(self, replay):
        """Staging ring in this GPU's memory for the rows of the rollout workers dealt to this learner rank
        (``rollout_devices``; round-robin over the ranks, ``actors/rollout_worker.py::worker_assignment``)."""
        cfg = self.config
        devs = cfg.get("rollout_devices") or []
        if not devs or self.device.type != "cuda":
            return None
        from ..actors.rollout_worker import workers_of_rank
        mine = workers_of_rank(len(devs), self.comm.world, self.rank)
        if mine == 0:
            return None
        if self.rank == 0 and self._shared is None:
            raise RuntimeError("rollout_devices needs the learner's weights shared through HIP IPC (share_weights_ipc)")
        from ..parallel.row_mailbox import RowMailbox
        rows = int(cfg["gpu_rollout_envs"]) * int(cfg.get("gpu_rollout_segment", 32))
        return RowMailbox.create(self.ctrl.name, mine, rows, replay.row_ld, self.device, learner_rank=self.rank)

    def _make_rollout(self, replay):
        """Explorers resident on this GPU (``gpu_rollout_envs`` > 0 and the environment exists as device
        kernels): they fill the replay next to, or instead of, the CPU explorer processes."""
        cfg = self.config
        n_envs = int(cfg.get("gpu_rollout_envs") or 0)
        if n_envs <= 0 or self.device.type != "cuda" or cfg.get("rollout_devices"):
            return None
        from ..actors.gpu_rollout import DeviceRollout
        from ..envs.device_envs import make_device_env
        seed = int(cfg["random_seed"]) + 7919 * (self.rank + 1)
        env = make_device_env(cfg, n_envs, self.device, seed)
        if env is None:
            raise ValueError(f"gpu_rollout_envs: no device-resident implementation of '{cfg['env']}'")
        return DeviceRollout(cfg, env, replay, self.device, segment=int(cfg.get("gpu_rollout_segment", 32)), seed=seed)

    # ---------------------------------------------------------------- host path
    def run_host(self, batch_queue, priority_queue=None) -> None:
        cfg, core, ctrl = self.config, self.core, self.ctrl
        self.publish_weights(wait=True)
        ctrl.store(ControlBlock.LEARNER_ALIVE, 1)
        per = bool(cfg["replay_memory_prioritized"])
        while core.step < self.num_train_steps and not ctrl.stop:
            if self._throttle():
                time.sleep(0.0005)
                continue
            try:
                arrays = batch_queue.get(timeout=0.05)
            except queue.Empty:
                self._beat()
                continue
            s, a, r, s2, d, g, w, idx = (torch.from_numpy(np.ascontiguousarray(x)) for x in arrays)
            batch = Batch(s.float(), a.float(), r.float(), s2.float(), d.float(), g.float(),
                          w.float() if per else None, idx.long()).to(self.device)
            prio = core.update(batch)
            if per and priority_queue is not None:
                try:
                    priority_queue.put((arrays[7], prio.detach().cpu().numpy()), timeout=1.0)
                except queue.Full:
                    pass
            self._after_update()
        self._finish()


class LearnerD4PG(Learner):
    """Distributional (51-atom categorical) critic learner (``models/d4pg/d4pg.py``)."""
    expected_models = ("d4pg",)


class LearnerD3PG(Learner):
    """Scalar-critic learner for D3PG and DDPG (``models/d3pg/d3pg.py``)."""
    expected_models = ("d3pg", "ddpg")


def learner_class(model: str):
    return LearnerD4PG if model == "d4pg" else LearnerD3PG
