"""Rollout workers: device-resident explorers in their OWN process (``actors/rollout_worker.py``), weights read out of
the learner's HIP-IPC slots, rows delivered into the learner's HBM through ``parallel/row_mailbox.py`` (GPU only).
On a one-GPU box worker and learner share ``cuda:0``; the two-GPU case runs where two devices exist."""
import os
import time

import pytest
import torch

from d4pg_pytorch_amd.config import default_config

pytestmark = pytest.mark.gpu


def _cfg(**kw):
    base = dict(env="Pendulum-v0", model="d4pg", batch_size=64, dense_size=128, device="cuda", v_min=-20.0, v_max=0.0,
                n_step_returns=5, replay_mem_size=50_000, gpu_rollout_envs=64, gpu_rollout_segment=16,
                rollout_devices=[0], num_agents=1, random_seed=5)
    base.update(kw)
    return default_config(**base)


def _worker_child(_, cfg, ctrl_name, device_index, segments):
    from d4pg_pytorch_amd.actors.rollout_worker import run_rollout_worker
    from d4pg_pytorch_amd.parallel.ipc import ControlBlock
    ctrl = ControlBlock.attach(ctrl_name)
    run_rollout_worker(cfg, ctrl, 0, device_index, hb_slot=3, max_segments=segments)


def _deliver(worker_device: int, segments: int = 5):
    """A learner-side process (this one) and one worker process; returns (replay rows, expected rows)."""
    import torch.multiprocessing as mp
    from d4pg_pytorch_amd.actors.gpu_rollout import DeviceRollout
    from d4pg_pytorch_amd.actors.rollout_worker import _RowShape
    from d4pg_pytorch_amd.envs.device_envs import make_device_env
    from d4pg_pytorch_amd.models.core import build_core
    from d4pg_pytorch_amd.models.learner import WeightPublisher
    from d4pg_pytorch_amd.parallel.hip_ipc import SharedDeviceBuffer, write_handle
    from d4pg_pytorch_amd.parallel.ipc import ControlBlock
    from d4pg_pytorch_amd.parallel.row_mailbox import RowMailbox
    from d4pg_pytorch_amd.replay.hbm_replay import HbmReplay
    cfg = _cfg(rollout_devices=[worker_device])
    E, T = cfg["gpu_rollout_envs"], cfg["gpu_rollout_segment"]
    core = build_core(cfg, "cuda:0")
    replay = HbmReplay(cfg, "cuda:0", core.dev_state)
    ctrl = ControlBlock.create(8)
    shared = SharedDeviceBuffer.create(4 * core.layout.actor.numel, "cuda:0")
    write_handle(ctrl.name, shared.handle, shared.numel, 0)
    pub = WeightPublisher(core, None, ctrl, "cuda:0", shared)
    mailbox = RowMailbox.create(ctrl.name, 1, E * T, replay.row_ld, "cuda:0")
    try:
        pub.publish()
        pub.poll(block=True)
        assert ctrl.load(ControlBlock.GPU_WEIGHT_VERSION) == 1 and ctrl.load(ControlBlock.GPU_WEIGHT_ISSUED) == 1
        child = mp.start_processes(_worker_child, args=(cfg, ctrl.name, worker_device, segments), nprocs=1, join=False,
                                   start_method="spawn")
        deadline = time.time() + 120.0
        while mailbox.rows_received < segments * E * T and time.time() < deadline:
            mailbox.poll(replay)
            time.sleep(0.002)
        torch.cuda.synchronize()
        mailbox.release(block=True)
        child.join(timeout=60)
        assert mailbox.rows_received == segments * E * T, mailbox.rows_received
        assert ctrl.env_steps == segments * E * T and ctrl.last_heartbeat(3) > 0
        got = replay.rows[: segments * E * T].clone()
        # the same explorers run here: same seeds, same policy -> the same rows, in the same order
        seed = int(cfg["random_seed"]) + 104729
        ro = DeviceRollout(cfg, make_device_env(cfg, E, "cuda:0", seed), _RowShape(cfg["state_dim"], cfg["action_dim"]),
                           "cuda:0", segment=T, seed=seed)
        ro.refresh_weights(core.online.flat.data_ptr(), core.layout.actor.base)
        want = []
        for _ in range(segments):
            ro.launch()
            ro.drain(insert=False)
            torch.cuda.synchronize()
            want.append(ro.rows.clone())
        return got, torch.cat(want)
    finally:
        mailbox.close()
        shared.close()
        try:
            os.unlink(os.path.join("/dev/shm", ctrl.name.strip("/") + ".hipipc"))
        except OSError:
            pass
        ctrl.close()


def test_worker_rows_arrive_complete_and_in_order_same_gpu():
    got, want = _deliver(0)
    assert torch.isfinite(got).all()
    assert torch.equal(got, want)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs")
def test_worker_on_another_gpu_delivers_over_xgmi():
    if not torch.cuda.can_device_access_peer(1, 0):
        pytest.skip("no peer access between GPU 1 and GPU 0")
    got, want = _deliver(1)
    assert torch.equal(got, want)


def test_engine_trains_the_pendulum_from_rollout_workers(tmp_path):
    """No explorer in the learner's process and no CPU explorer: a rollout worker process (here on the learner's GPU)
    produces every row; the GPU exploiter follows through the same IPC slots.

    The configuration of tests/test_learning_gpu.py::test_d4pg_fp32_learns_the_pendulum_within_20k_updates with
    tau = 0.005 and 30 000 updates (with the reference's tau = 0.001 the target policy leaves its constant-torque phase
    after about 18 000 updates, bench/exp/d4pg_learn_sweep.py).  D4PG on this task ends in a saturated constant-torque
    policy in about one run in ten whatever feeds the replay (deterministic in-process explorers too: bench/exp/r3_ratio.py,
    2 of 18 runs), and what a free-running worker process has delivered by a given update depends on timing, so a run
    that ends there is repeated once with another seed: both collapsing is a one-in-a-hundred event, a broken data path
    fails both."""
    from d4pg_pytorch_amd.engine import load_engine
    from d4pg_pytorch_amd.models.nets import ActorPolicy
    from d4pg_pytorch_amd.utils.checkpoint import load_learner_bundle
    from d4pg_pytorch_amd.utils.learning import GreedyEvaluator
    from d4pg_pytorch_amd.utils.logger import read_events
    returns = []
    for attempt, seed in enumerate((103, 203)):
        cfg = dict(env="Pendulum-v0", model="d4pg", num_agents=1, batch_size=256, num_steps_train=30_000, tau=0.005,
                   replay_mem_size=1_000_000, n_step_returns=5, device="cuda", dense_size=400, v_min=-20, v_max=0,
                   metrics_interval=1000, weight_publish_interval=50, num_episode_save=1000, gpu_rollout_envs=64,
                   gpu_rollout_segment=32, rollout_devices=[0], updates_per_env_step=0.125, random_seed=seed,
                   envs_per_agent=4, results_path=str(tmp_path / f"results{attempt}"))
        engine = load_engine(cfg)
        exp = engine.train()
        assert engine.final_update_step == 30_000
        assert all(c == 0 for c in engine.exit_codes.values()), engine.exit_codes
        assert set(engine.exit_codes) == {"learner-0", "agent-0", "rollout-0"}
        assert engine.final_env_steps >= 30_000 * 8 - 64 * 32
        rewards = [e[3] for e in read_events(os.path.join(exp, "agent-0")) if e[2] == "agent/reward"]
        assert len(rewards) >= 4 and min(rewards[:4]) < -1000, rewards        # the exploiter followed through the IPC slots
        # What was learned, measured the way tests/test_learning_gpu.py measures it: the checkpointed TARGET policy,
        # played greedily on 64 fresh device pendulums (the exploiter's own few episodes race a three-second run; a
        # random policy scores about -1200).
        bundle = load_learner_bundle(os.path.join(exp, "learner", "checkpoint_final.pt"))
        pol = ActorPolicy(3, 1, 400, "cuda")
        pol.load_state_dict(bundle["target_actor"])
        returns.append(GreedyEvaluator(engine.config, 64, "cuda", 7)(pol.flat.data_ptr(), 0))
        if returns[-1] > -400.0:
            break
    assert returns[-1] > -400.0, returns


def _crash_once_rollout_entry(config, ctrl_name, k, device_index, hb_slot):
    """Worker that delivers two segments and dies on its first start; its replacement carries on."""
    from d4pg_pytorch_amd.actors.rollout_worker import rollout_worker, run_rollout_worker
    from d4pg_pytorch_amd.parallel.ipc import ControlBlock
    marker = os.path.join(config["results_path"], "rollout_crashed_once")
    if not os.path.exists(marker):
        os.makedirs(os.path.dirname(marker), exist_ok=True)
        open(marker, "w").close()
        run_rollout_worker(config, ControlBlock.attach(ctrl_name), k, device_index, hb_slot, max_segments=2)
        os._exit(13)
    rollout_worker(config, ctrl_name, k, device_index, hb_slot)


def test_killed_rollout_worker_is_replaced(tmp_path, monkeypatch):
    from d4pg_pytorch_amd import engine as eng
    monkeypatch.setattr(eng, "rollout_worker_entry", _crash_once_rollout_entry)
    cfg = dict(env="Pendulum-v0", model="d4pg", num_agents=1, batch_size=64, num_steps_train=400,
               replay_mem_size=100_000, device="cuda", dense_size=128, v_min=-20, v_max=0, gpu_rollout_envs=64,
               gpu_rollout_segment=16, rollout_devices=[0], updates_per_env_step=0.05, min_replay_size=4096,
               results_path=str(tmp_path / "results"))
    e = eng.load_engine(cfg)
    e.train()
    assert e.restarts == 1 and e.final_update_step == 400
    assert e.exit_codes["rollout-0"] == 0 and e.exit_codes["learner-0"] == 0
    assert e.final_env_steps >= 400 * 20


def test_two_learner_ranks_are_each_fed_by_their_own_worker(tmp_path, monkeypatch, capfd):
    """Data-parallel learners AND rollout workers: the workers are dealt to the learner ranks round-robin, every rank
    owns a row mailbox in its GPU's memory, so every rank's replay shard is fed.  (Both ranks share the one GPU here,
    so their collectives go through gloo: RCCL refuses two ranks on one device.)"""
    import re
    from d4pg_pytorch_amd.engine import load_engine
    monkeypatch.setenv("D4PG_DIST_BACKEND", "gloo")
    cfg = dict(env="Pendulum-v0", model="d4pg", num_agents=1, batch_size=128, num_steps_train=1500,
               replay_mem_size=100_000, device="cuda", dense_size=128, v_min=-20, v_max=0, gpu_rollout_envs=64,
               gpu_rollout_segment=16, num_learners=2, learner_devices=[0, 0], rollout_devices=[0, 0],
               results_path=str(tmp_path / "results"))
    e = load_engine(cfg)
    e.train()
    assert all(c == 0 for c in e.exit_codes.values()), e.exit_codes
    assert set(e.exit_codes) == {"learner-0", "learner-1", "agent-0", "rollout-0", "rollout-1"}
    assert e.final_update_step == 1500
    out = capfd.readouterr().out
    got = {int(m.group(1)): int(m.group(2)) for m in re.finditer(r"\[learner (\d)\] (\d+) rows from 1 rollout worker", out)}
    assert set(got) == {0, 1} and min(got.values()) >= 64 * 16 * 4, (got, out[-2000:])
    # a job whose ranks could not all be fed is refused before anything is spawned
    with pytest.raises(ValueError, match="round-robin"):
        load_engine(dict(cfg, rollout_devices=[0])).train()
