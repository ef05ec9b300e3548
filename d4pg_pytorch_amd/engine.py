"""Job orchestration: process topology, IPC wiring, supervision, teardown.

``load_engine(config).train()`` is the reference's entry
(``models/engine.py:5-10``, ``models/d4pg/engine.py:101-158``,
``models/d3pg/engine.py:94-151``).  The roles it spawns are kept:

* learner(s)            — own the networks (one per GPU when ``num_learners > 1``)
* sampler               — owns the replay **only in the host topology**
                          (``replay_device: host``); with ``replay_device: hbm``
                          that role is the ingest path + sampling kernel inside the learner
* agent 0 (exploiter)   — greedy target policy, evaluation reward, actor checkpoints
* agents 1..n-1         — explorers (online policy + OU noise), the data producers

What is new is everything between them: transitions travel through
shared-memory rings, weights through a seq-locked broadcast slab, counters and
the stop flag through an atomic control block (``parallel/ipc.py``), and the
parent is a *supervisor*: it watches exit codes and heartbeats, restarts a
crashed explorer, tears the job down if a learner or the sampler dies, and
always unlinks the shared segments.  (The reference only ``start()``s and
``join()``s; a crashed child leaves the others spinning forever.)
"""
from __future__ import annotations

import multiprocessing as mp
import os
import queue
import sys
import time
import traceback
from datetime import datetime
from typing import Dict, List, Optional

import numpy as np

from .config import beta_at, finalize_config
from .parallel.ipc import ControlBlock, ShmRing, WeightSlab, cleanup_names
from .utils.logger import Logger
from .utils.misc import empty_torch_queue


# --------------------------------------------------------------------------- workers
def _set_threads(n: int = 1) -> None:
    try:
        import torch
        torch.set_num_threads(n)
    except Exception:
        pass


def learner_worker(config: Dict, ctrl_name: str, slab_name: str, ring_names: List[str], log_dir: str,
                   rank: int, world: int, master_port: int, batch_queue=None, priority_queue=None) -> None:
    """Learner rank ``rank`` of ``world`` (``models/d4pg/engine.py:80-83``)."""
    import torch
    from .models.learner import learner_class
    from .parallel.comm import LocalComm, TorchDistComm
    ctrl = ControlBlock.attach(ctrl_name)
    try:
        device = torch.device(config["device"])
        if device.type == "cuda":
            index = int(config["learner_devices"][rank]) if world > 1 else (device.index or 0)
            torch.cuda.set_device(index)
            device = torch.device("cuda", index)
        else:
            _set_threads(max(1, (os.cpu_count() or 2) // 2))
        comm = LocalComm()
        if world > 1:
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(master_port), RANK=str(rank),
                              WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
            comm = TorchDistComm(device=device, rank=rank, world=world)
        slab = WeightSlab.attach(slab_name) if rank == 0 else None
        learner = learner_class(config["model"])(config, ctrl, slab, log_dir, rank, comm, device)
        if config["replay_device"] == "hbm":
            learner.run_hbm([ShmRing.attach(n) for n in ring_names])
        else:
            learner.run_host(batch_queue, priority_queue)
        comm.close()
    except Exception:
        traceback.print_exc()
        ctrl.store(ControlBlock.FAILED, 1)
        ctrl.request_stop()
        raise
    finally:
        for q in (batch_queue, priority_queue):
            if q is not None:
                empty_torch_queue(q)


def sampler_worker(config: Dict, ctrl_name: str, ring_names: List[str], batch_queue, priority_queue,
                   log_dir: str, hb_slot: Optional[int] = None) -> None:
    """Replay owner of the host topology (``models/d4pg/engine.py:23-77``)."""
    from .replay.host_replay import HostReplay
    _set_threads(1)
    ctrl = ControlBlock.attach(ctrl_name)
    try:
        rings = [ShmRing.attach(n) for n in ring_names]
        replay = HostReplay(config)
        logger = Logger(os.path.join(log_dir, "data_struct"), config.get("log_backend", "tb"))
        batch_size = int(config["batch_size"])
        min_rows = max(batch_size, int(config["min_replay_size"]))
        scratch = np.empty((int(config["ingest_chunk"]), replay.row_ld), dtype=np.float32)
        last_log = 0.0
        while not ctrl.stop:
            for r in rings:
                rows = r.pop(scratch.shape[0], out=scratch)
                if rows.shape[0]:
                    replay.insert_rows(rows)
            if priority_queue is not None:
                try:
                    while True:
                        idx, pr = priority_queue.get_nowait()
                        replay.update_priorities(idx, pr)
                except queue.Empty:
                    pass
            if len(replay) < min_rows:
                time.sleep(0.001)
                continue
            batch = replay.sample(batch_size, beta_at(config, ctrl.update_step))
            try:
                batch_queue.put(batch, timeout=0.1)
            except queue.Full:
                pass
            now = time.time()
            if now - last_log > 0.5:
                last_log, step = now, ctrl.update_step
                logger.scalar_summary("data_struct/global_episode", ctrl.global_episode, step)
                logger.scalar_summary("data_struct/replay_queue", sum(len(r) for r in rings), step)
                logger.scalar_summary("data_struct/batch_queue", batch_queue.qsize() if hasattr(batch_queue, "qsize") else 0, step)
                logger.scalar_summary("data_struct/replay_buffer", len(replay), step)
            ctrl.heartbeat(ctrl.n_workers - 1 if hb_slot is None else hb_slot, int(now * 1000))
        if config.get("save_buffer_on_disk"):
            replay.dump(os.path.join(config["results_path"], "replay_buffer.pkl"))
        logger.close()
        print("Stop sampler worker.", flush=True)
    except Exception:
        traceback.print_exc()
        ctrl.store(ControlBlock.FAILED, 1)
        ctrl.request_stop()
        raise
    finally:
        empty_torch_queue(batch_queue)
        if priority_queue is not None:
            empty_torch_queue(priority_queue)


def agent_worker(config: Dict, ctrl_name: str, slab_name: str, ring_name: Optional[str], n_agent: int,
                 agent_type: str, log_dir: str, hb_slot: Optional[int] = None) -> None:
    """Explorer / exploiter process (``models/d4pg/engine.py:86-94``)."""
    from .actors.agent import Agent, make_policy
    _set_threads(1)
    ctrl = ControlBlock.attach(ctrl_name)
    slab = WeightSlab.attach(slab_name)
    ring = ShmRing.attach(ring_name) if ring_name else None
    device = config["exploiter_device"] if agent_type == "exploitation" else config["agent_device"]
    if str(device).startswith("cuda"):
        import torch
        if not torch.cuda.is_available():
            device = "cpu"
    policy = make_policy(config, device)
    agent = Agent(config, policy, ctrl, n_agent=n_agent, agent_type=agent_type, log_dir=log_dir,
                  ring=ring, slab=slab, hb_slot=hb_slot)
    agent.run()


def rollout_worker_entry(config: Dict, ctrl_name: str, k: int, device_index: int, hb_slot: int) -> None:
    """Device-resident explorers on GPU ``device_index`` (``actors/rollout_worker.py``)."""
    from .actors.rollout_worker import rollout_worker
    _set_threads(1)
    rollout_worker(config, ctrl_name, k, device_index, hb_slot)


# --------------------------------------------------------------------------- engine
class Engine:
    def __init__(self, config: Dict):
        self.config = finalize_config(config)
        self.experiment_dir: Optional[str] = None
        self.exit_codes: Dict[str, int] = {}
        self.final_update_step = 0
        self.final_env_steps = 0
        self.final_episodes = 0
        self.restarts = 0
        self.hung_killed = 0
        self.hb_slots: Dict[str, int] = {}

    def _spawn(self, ctx, name: str, target, args) -> mp.Process:
        p = ctx.Process(target=target, args=args, name=name, daemon=False)
        p.start()
        return p

    def train(self) -> str:
        cfg = self.config
        ctx = mp.get_context("spawn")
        stamp = datetime.now().strftime("%Y-%m-%d_%H:%M:%S")
        exp_dir = os.path.join(cfg["results_path"], f"{cfg['env']}-{cfg['model']}-{stamp}")
        os.makedirs(exp_dir, exist_ok=True)
        self.experiment_dir = exp_dir
        world = int(cfg["num_learners"]) if str(cfg["device"]).startswith("cuda") else 1
        n_agents = int(cfg["num_agents"])
        n_explorers_per_rank = n_agents - 1
        total_agents = 1 + world * n_explorers_per_rank
        host_mode = cfg["replay_device"] == "host"
        if host_mode and world > 1:
            raise ValueError("data-parallel learners need replay_device: hbm")

        from .models.nets import NetLayout
        from .replay.hbm_replay import row_ld
        n_actor = NetLayout.build(cfg["state_dim"], cfg["dense_size"], cfg["action_dim"]).numel
        rollout_devices = list(cfg.get("rollout_devices") or []) if str(cfg["device"]).startswith("cuda") else []
        if rollout_devices:
            import torch
            n_gpus = torch.cuda.device_count()          # (counting devices does not initialise the GPU in this process)
            if world > 1 and n_explorers_per_rank == 0 and len(rollout_devices) < world:
                raise ValueError(f"rollout_devices {rollout_devices}: {world} learner ranks are dealt the workers round-robin "
                                 f"and there are no CPU explorers, so ranks {len(rollout_devices)}..{world - 1} would never "
                                 f"see a transition: list at least {world} devices (an index may repeat)")
            missing = [d for d in rollout_devices if d >= n_gpus]
            if missing:
                raise ValueError(f"rollout_devices {rollout_devices}: this machine has {n_gpus} GPU(s), no cuda:{missing[0]} "
                                 f"(e.g. --set rollout_devices={list(range(min(1, n_gpus - 1), n_gpus)) or [0]})")
        ctrl = ControlBlock.create(n_workers=world + total_agents + 2 + len(rollout_devices))
        slab = WeightSlab.create(2 * n_actor)
        rings: List[ShmRing] = []
        names = [ctrl.name, slab.name]
        procs: Dict[str, mp.Process] = {}
        specs: Dict[str, tuple] = {}
        batch_queue = priority_queue = None
        # one heartbeat slot per process: learner ranks first, then the agents, the sampler last
        self.hb_slots = {f"learner-{r}": r for r in range(world)}
        self.hb_slots.update({f"agent-{i}": world + i for i in range(total_agents)})
        self.hb_slots["sampler"] = world + total_agents
        self.hb_slots.update({f"rollout-{k}": world + total_agents + 1 + k for k in range(len(rollout_devices))})
        try:
            rl = row_ld(cfg["state_dim"], cfg["action_dim"])
            for _ in range(world * n_explorers_per_rank):
                r = ShmRing.create(int(cfg["ring_capacity"]), rl)
                if cfg.get("check_rings"):       # race detection: slot sequence numbers, verified on consumption
                    r.set_check(True)
                rings.append(r)
                names.append(r.name)
            ring_names = [r.name for r in rings]
            port = 29600 + (os.getpid() % 3000)
            if host_mode:
                batch_queue = ctx.Queue(maxsize=int(cfg["batch_queue_size"]))
                priority_queue = ctx.Queue(maxsize=int(cfg["replay_queue_size"])) \
                    if cfg["replay_memory_prioritized"] else None
                specs["sampler"] = (sampler_worker, (cfg, ctrl.name, ring_names, batch_queue, priority_queue, exp_dir,
                                                     self.hb_slots["sampler"]))
            for rank in range(world):
                mine = ring_names[rank * n_explorers_per_rank:(rank + 1) * n_explorers_per_rank]
                specs[f"learner-{rank}"] = (learner_worker, (cfg, ctrl.name, slab.name, mine, exp_dir, rank, world,
                                                             port, batch_queue, priority_queue))
            specs["agent-0"] = (agent_worker, (cfg, ctrl.name, slab.name, None, 0, "exploitation", exp_dir,
                                               self.hb_slots["agent-0"]))
            for i in range(world * n_explorers_per_rank):
                specs[f"agent-{i + 1}"] = (agent_worker, (cfg, ctrl.name, slab.name, ring_names[i], i + 1,
                                                          "exploration", exp_dir, self.hb_slots[f"agent-{i + 1}"]))
            for k, dev_index in enumerate(rollout_devices):
                specs[f"rollout-{k}"] = (rollout_worker_entry, (cfg, ctrl.name, k, int(dev_index), self.hb_slots[f"rollout-{k}"]))
            # learners first: agents block on nothing, but fresh weights should exist early
            for name, (fn, args) in specs.items():
                procs[name] = self._spawn(ctx, name, fn, args)
            self._supervise(ctx, procs, specs, ctrl)
        finally:
            ctrl.request_stop()
            deadline = time.time() + 30.0
            for name, p in procs.items():
                p.join(timeout=max(0.1, deadline - time.time()))
                if p.is_alive():
                    print(f"[engine] terminating unresponsive process {name}", flush=True)
                    p.terminate()
                    p.join(timeout=5.0)
                    if p.is_alive():
                        p.kill()
                self.exit_codes[name] = p.exitcode if p.exitcode is not None else -999
            self.final_update_step = ctrl.update_step
            self.final_env_steps = ctrl.env_steps
            self.final_episodes = ctrl.global_episode
            failed = ctrl.load(ControlBlock.FAILED)
            self.ring_seq_errors = sum(r.seq_errors for r in rings)
            if cfg.get("check_rings"):
                pushed = sum(r.pushed_total for r in rings)
                print(f"[engine] ring check: {pushed} rows through {len(rings)} rings, "
                      f"{self.ring_seq_errors} sequence errors", flush=True)
                if self.ring_seq_errors:
                    failed = failed or 1
            for q in (batch_queue, priority_queue):
                if q is not None:
                    empty_torch_queue(q)
            for r in rings:
                r.close()
            slab.close()
            # device weight handle, row staging rings of the rollout workers (one per learner rank that is dealt workers)
            from .parallel.row_mailbox import mailbox_path
            for path in [os.path.join("/dev/shm", ctrl.name.strip("/") + ".hipipc")] + \
                    [mailbox_path(ctrl.name, r) for r in range(world)]:
                try:
                    os.unlink(path)
                except OSError:
                    pass
            ctrl.close()
            cleanup_names(names)
        if failed:
            raise RuntimeError(f"training failed; exit codes: {self.exit_codes}")
        return exp_dir

    def _supervise(self, ctx, procs: Dict[str, mp.Process], specs: Dict[str, tuple], ctrl: ControlBlock) -> None:
        cfg = self.config
        hb_timeout_ms = float(cfg["heartbeat_timeout_s"]) * 1000.0
        # A replaceable process (agent, rollout worker) may be replaced `max_restarts_per_process` times; the job
        # gives up on one that exceeds it, or when replacements come faster than `max_restarts_per_hour` overall
        # (one flaky explorer in a day-long run is not a reason to end it).
        per_proc = int(cfg.get("max_restarts_per_process", 3))
        per_hour = int(cfg.get("max_restarts_per_hour", 20))
        counts: Dict[str, int] = {}
        recent: List[float] = []

        def may_restart(name: str) -> bool:
            now = time.time()
            recent[:] = [t for t in recent if now - t < 3600.0]
            if counts.get(name, 0) >= per_proc or len(recent) >= per_hour:
                return False
            counts[name] = counts.get(name, 0) + 1
            recent.append(now)
            self.restarts += 1
            return True

        while True:
            time.sleep(0.05)
            if ctrl.stop:
                return
            for name, p in list(procs.items()):
                if p.is_alive():
                    continue
                code = p.exitcode
                critical = name.startswith("learner") or name == "sampler"
                if critical:
                    if code != 0:
                        print(f"[engine] {name} died with exit code {code}: stopping the job", flush=True)
                        ctrl.store(ControlBlock.FAILED, 1)
                    ctrl.request_stop()
                    return
                if code != 0 and not ctrl.stop:
                    if may_restart(name):
                        print(f"[engine] {name} crashed (exit {code}); restart {counts[name]}/{per_proc}", flush=True)
                        fn, args = specs[name]
                        procs[name] = self._spawn(ctx, name, fn, args)
                    else:
                        print(f"[engine] {name} keeps crashing: stopping the job", flush=True)
                        ctrl.store(ControlBlock.FAILED, 1)
                        ctrl.request_stop()
                        return
            # heartbeats: a process that has beaten once and then falls silent is hung (blocked in a collective, a
            # dead GPU queue, an environment that never returns).  Hung explorers are replaced like crashed ones;
            # a hung learner rank or sampler ends the job (its peers sit in the same collectives).
            now_ms = time.time() * 1000.0
            for name, p in list(procs.items()):
                slot = self.hb_slots.get(name)
                if slot is None or not p.is_alive():
                    continue
                hb = ctrl.last_heartbeat(slot)
                if not hb or now_ms - hb <= hb_timeout_ms:
                    continue
                if name.startswith("learner") or name == "sampler":
                    print(f"[engine] {name} heartbeat lost: stopping the job", flush=True)
                    ctrl.store(ControlBlock.FAILED, 1)
                    ctrl.request_stop()
                    return
                print(f"[engine] {name} heartbeat lost: killing it", flush=True)
                p.kill()
                p.join(timeout=5.0)
                ctrl.heartbeat(slot, 0)
                self.hung_killed += 1
                if may_restart(name):
                    fn, args = specs[name]
                    procs[name] = self._spawn(ctx, name, fn, args)
                else:
                    print(f"[engine] {name} keeps hanging: stopping the job", flush=True)
                    ctrl.store(ControlBlock.FAILED, 1)
                    ctrl.request_stop()
                    return


def load_engine(config: Dict) -> Engine:
    """Engine for ``config['model']`` in {ddpg, d3pg, d4pg}; anything else raises
    (the reference returns ``None`` silently, ``models/engine.py:7-10``)."""
    print(f"Loading {config.get('model')} for {config.get('env')}.", flush=True)
    cfg = finalize_config(config)
    return Engine(cfg)
