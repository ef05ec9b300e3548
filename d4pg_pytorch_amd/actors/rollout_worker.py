"""Rollout worker: device-resident explorers on a GPU the learner does not use.

The reference's explorers are separate processes that feed the sampler (``models/d4pg/engine.py:147-151``,
``models/agent.py:51-123``).  A rollout worker is that role for environments that exist as device kernels: one
process per entry of ``rollout_devices``, owning a device environment of ``gpu_rollout_envs`` instances and a
:class:`DeviceRollout` (policy launch, OU noise, dynamics, n-step rows as one captured graph per segment) on ITS
GPU.  It

* reads the online actor out of the learner's HIP-IPC weight slots in place (``parallel/hip_ipc.py``; peer memory
  over xGMI when the worker's GPU is not the learner's) whenever the published version changes;
* delivers every segment's rows into the learner's HBM through :class:`RowMailbox` (a device-to-device copy into a
  staging ring the learner exported, then host atomics), where one ``replay_insert`` launch takes them;
* beats its heartbeat slot, counts environment steps and episodes in the control block, and ends on the stop flag.

On the learner's own GPU the same explorers cost the learner 14 - 50 % of its update rate; on another GPU they cost
it one insert launch per segment.
"""
from __future__ import annotations

import os
import sys
import time
import traceback
from typing import Dict

from ..parallel.ipc import ControlBlock


class _RowShape:
    """What :class:`DeviceRollout` needs from a replay it never inserts into."""

    def __init__(self, state_dim: int, action_dim: int):
        from ..replay.hbm_replay import row_ld
        self.row_ld = row_ld(state_dim, action_dim)

    def insert(self, rows) -> None:      # pragma: no cover - the worker delivers rows through the mailbox
        raise RuntimeError("a rollout worker has no replay of its own")


def worker_assignment(k: int, n_learners: int):
    """(learner rank, index in that rank's mailbox) of rollout worker ``k``: workers are dealt to the learner ranks
    round-robin, so that with data-parallel learners every rank's replay shard is fed."""
    n = max(1, int(n_learners))
    return k % n, k // n


def workers_of_rank(n_workers: int, n_learners: int, rank: int) -> int:
    return len([k for k in range(int(n_workers)) if worker_assignment(k, n_learners)[0] == rank])


def run_rollout_worker(config: Dict, ctrl: ControlBlock, k: int, device_index: int, hb_slot: int,
                       max_segments: int = 0, stats: Dict = None) -> int:
    """Body of worker ``k``; returns the environment steps it delivered (``stats``, if given, receives the number of
    policy refreshes and the last weight version packed).  The policy always comes from learner rank 0's weight slots
    (all ranks hold the same parameters); the rows go to the mailbox of the rank the worker is dealt to."""
    import torch
    from ..envs.device_envs import make_device_env
    from ..parallel.hip_ipc import SharedDeviceBuffer, read_handle
    from ..parallel.row_mailbox import RowMailbox
    from .gpu_rollout import DeviceRollout

    torch.cuda.set_device(device_index)
    device = torch.device("cuda", device_index)
    cfg = config
    n_envs, segment = int(cfg["gpu_rollout_envs"]), int(cfg.get("gpu_rollout_segment", 32))
    seed = int(cfg["random_seed"]) + 104729 * (k + 1)
    env = make_device_env(cfg, n_envs, device, seed)
    if env is None:
        raise ValueError(f"rollout_devices: no device-resident implementation of '{cfg['env']}'")
    rollout = DeviceRollout(cfg, env, _RowShape(cfg["state_dim"], cfg["action_dim"]), device, segment=segment, seed=seed)

    debug = bool(os.environ.get("D4PG_DEBUG_WORKER"))      # a timeline of packs, segments and pushes on stderr

    def note(msg: str) -> None:
        if debug:
            print(f"[rollout worker {k} {time.time() % 1000:8.3f}] {msg}", file=sys.stderr, flush=True)

    def beat() -> None:
        ctrl.heartbeat(hb_slot, int(time.time() * 1000))

    def stopping() -> bool:
        beat()
        return ctrl.stop

    beat()
    n_learners = int(cfg.get("num_learners") or 1) if str(cfg.get("device", "cuda")).startswith("cuda") else 1
    learner_rank, box = worker_assignment(k, n_learners)
    mailbox = RowMailbox.open(ctrl.name, device, should_stop=stopping, learner_rank=learner_rank)
    if mailbox is None:
        return 0
    if mailbox.rows_per_slot != n_envs * segment or box >= mailbox.n_workers:
        raise ValueError("rollout worker and learner disagree on the segment size or the number of workers")
    # the learner's weight slots: [online actor | target actor] x 2, written alternately (WeightPublisher)
    shared, n_actor, version = None, rollout.actor.layout.numel, 0
    delivered = segments = refreshes = 0
    try:
        while not ctrl.stop and (max_segments <= 0 or segments < max_segments):
            if shared is None:
                got = read_handle(ctrl.name)
                if got is not None:
                    handle, numel, dev_index = got
                    if dev_index != device_index and not torch.cuda.can_device_access_peer(device_index, dev_index):
                        raise RuntimeError(f"cuda:{device_index} has no peer access to the learner's cuda:{dev_index}")
                    shared = SharedDeviceBuffer.open(handle, numel, device)
            v = ctrl.load(ControlBlock.GPU_WEIGHT_VERSION)
            if shared is None or v < 1:
                beat()
                time.sleep(0.002)            # nothing to act with before the learner's first publication
                continue
            if v != version:
                note(f"pack version {v} (had {version}), segment {segments}")
                with torch.cuda.stream(rollout.stream):
                    for _ in range(3):
                        base = ((v - 1) % 2) * 2 * n_actor       # the online half of the slot of publication v
                        rollout.actor.pack_from(shared.ptr, base)
                        rollout.stream.synchronize()
                        # A slot is rewritten by the publication two after its own: pack again if that one was
                        # issued while this pack ran.  At most twice: a learner that publishes faster than a pack
                        # takes (two processes sharing one GPU) must not starve the explorers, and a policy torn
                        # between two consecutive versions is what the reference's exploiter reads all the time
                        # (models/d4pg/engine.py:134, unsynchronised shared memory).
                        if ctrl.load(ControlBlock.GPU_WEIGHT_ISSUED) < v + 2 or ctrl.stop:
                            break
                        v = max(v, ctrl.load(ControlBlock.GPU_WEIGHT_VERSION))
                version = v
                refreshes += 1
                note("packed")
            rollout.launch()
            rollout.done_event.synchronize()
            note("segment done")
            rollout.drain(insert=False)
            with torch.cuda.stream(rollout.stream):
                if not mailbox.push(box, rollout.rows, should_stop=stopping):
                    break
            note("pushed")
            n = n_envs * segment
            delivered += n
            segments += 1
            ctrl.add(ControlBlock.ENV_STEPS, n)
            if segments % 8 == 0:
                _, n_ep = rollout.episode_stats()
                if n_ep:
                    ctrl.add(ControlBlock.GLOBAL_EPISODE, n_ep)
            beat()
    finally:
        if stats is not None:
            stats.update(policy_refreshes=refreshes, last_weight_version=version, segments=segments)
        torch.cuda.synchronize(device)
        mailbox.close()
        if shared is not None:
            shared.close()
    return delivered


def rollout_worker(config: Dict, ctrl_name: str, k: int, device_index: int, hb_slot: int) -> None:
    """Process entry (spawned by the engine)."""
    ctrl = ControlBlock.attach(ctrl_name)
    try:
        stats: Dict = {}
        n = run_rollout_worker(config, ctrl, k, device_index, hb_slot, stats=stats)
        print(f"Exit rollout worker {k} on cuda:{device_index}: {n} environment steps delivered in {stats.get('segments')} "
              f"segments, {stats.get('policy_refreshes')} policy refreshes (last weight version "
              f"{stats.get('last_weight_version')} of {ctrl.load(ControlBlock.GPU_WEIGHT_VERSION)}).", flush=True)
    except Exception:
        traceback.print_exc()
        raise
