"""End-to-end multi-process runs on the CPU: the reference's six smoke scenarios
(tests/test_pendulum.py, test_lunar.py, test_bipedal.py: 3 envs x {ddpg, d4pg}
with tests/config_test.yml values) — here with assertions."""
import glob
import os

import numpy as np
import pytest
import torch

from d4pg_pytorch_amd.config import default_config
from d4pg_pytorch_amd.engine import Engine, load_engine
from d4pg_pytorch_amd.utils.checkpoint import load_actor, load_learner_bundle
from d4pg_pytorch_amd.utils.logger import read_events

TEST_VALUES = dict(action_low=-1, action_high=1, num_agents=2, random_seed=2019, batch_size=256,
                   num_steps_train=100, max_ep_length=1000, replay_mem_size=1_000, n_step_returns=1,
                   device="cpu", agent_device="cpu", dense_size=100, v_min=0, v_max=10,
                   metrics_interval=20, weight_publish_interval=25)


def _run(tmp_path, env, model, **kw):
    cfg = dict(TEST_VALUES, env=env, model=model, results_path=str(tmp_path / "results"))
    cfg.update(kw)
    engine = load_engine(cfg)
    exp = engine.train()
    return engine, exp


def _check_run(engine, exp, model, n_agents=2, steps=100):
    assert engine.final_update_step == steps                      # the learner did exactly num_steps_train updates
    assert all(code == 0 for code in engine.exit_codes.values()), engine.exit_codes   # every process exited cleanly
    assert engine.final_env_steps > 0 and engine.final_episodes >= 0
    base = os.path.basename(exp)
    assert base.startswith(f"{engine.config['env']}-{model}-")     # {env}-{model}-{timestamp} (engine.py:108)
    ev = read_events(os.path.join(exp, "learner"))
    tags = {e[2] for e in ev}
    timing = "learner/learner_update_timing" if model == "d4pg" else "learner/update_time"
    assert {"learner/policy_loss", "learner/value_loss", timing} <= tags
    assert all(np.isfinite(e[3]) for e in ev)
    for i in range(n_agents):
        assert os.path.isdir(os.path.join(exp, f"agent-{i}"))
    bundle = load_learner_bundle(os.path.join(exp, "learner", "checkpoint_final.pt"))
    assert bundle["update_step"] == steps and bundle["model"] == model
    for net in ("actor", "critic", "target_actor", "target_critic"):
        assert all(torch.isfinite(v).all() for v in bundle[net].values())
    return bundle


@pytest.mark.parametrize("model", ["ddpg", "d4pg"])
def test_pendulum(tmp_path, model):
    engine, exp = _run(tmp_path, "Pendulum-v0", model, state_dim=3, action_dim=1)
    bundle = _check_run(engine, exp, model)
    # host topology on the CPU: the sampler role exists and logs its data-structure scalars
    assert "sampler" in engine.exit_codes
    tags = {e[2] for e in read_events(os.path.join(exp, "data_struct"))}
    assert {"data_struct/replay_buffer", "data_struct/batch_queue", "data_struct/global_episode"} <= tags
    # weights were trained (differ from the seeded init) and targets lag the online nets
    assert not torch.equal(bundle["actor"]["linear1.weight"], bundle["target_actor"]["linear1.weight"])


@pytest.mark.parametrize("model", ["ddpg", "d4pg"])
def test_lunar_lander_shape(tmp_path, model):
    engine, exp = _run(tmp_path, "LunarLanderContinuous-v2", model, state_dim=8, action_dim=2)
    _check_run(engine, exp, model)


@pytest.mark.parametrize("model", ["ddpg", "d4pg"])
def test_bipedal_walker_shape(tmp_path, model):
    engine, exp = _run(tmp_path, "BipedalWalker-v2", model, state_dim=24, action_dim=4, max_ep_length=200)
    _check_run(engine, exp, model)


def test_d4pg_prioritised_nstep_checkpoints_and_resume(tmp_path):
    """PER + n-step 5 + 3 agents on Pendulum; the exploiter writes actor checkpoints in the
    reference's location; a second job resumes from the learner bundle."""
    engine, exp = _run(tmp_path, "Pendulum-v0", "d4pg", state_dim=3, action_dim=1, num_agents=3,
                       replay_memory_prioritized=1, n_step_returns=5, num_steps_train=150,
                       v_min=-20, v_max=0, num_episode_save=1, batch_size=64, envs_per_agent=2,
                       updates_per_env_step=0.25, save_buffer_on_disk=1)
    bundle = _check_run(engine, exp, "d4pg", n_agents=3, steps=150)
    ckpts = glob.glob(os.path.join(exp, "agent_0", "local_episode_*_reward_*.pt"))
    assert ckpts, "agent 0 wrote no checkpoint"
    sd = load_actor(ckpts[0])
    assert sd["linear1.weight"].shape == (100, 3) and sd["linear3.weight"].shape == (1, 100)
    rewards = [e[3] for e in read_events(os.path.join(exp, "agent-0")) if e[2] == "agent/reward"]
    assert rewards and all(-2500 < r <= 0 for r in rewards)        # raw (un-normalised) pendulum returns
    assert os.path.exists(os.path.join(str(tmp_path / "results"), "replay_buffer.pkl"))
    # ---- resume ----
    engine2, exp2 = _run(tmp_path, "Pendulum-v0", "d4pg", state_dim=3, action_dim=1, num_agents=2,
                         replay_memory_prioritized=1, n_step_returns=5, num_steps_train=200, v_min=-20, v_max=0,
                         batch_size=64, resume=os.path.join(exp, "learner", "checkpoint_final.pt"))
    assert engine2.final_update_step == 200
    b2 = load_learner_bundle(os.path.join(exp2, "learner", "checkpoint_final.pt"))
    assert b2["update_step"] == 200
    assert not torch.equal(b2["actor"]["linear2.weight"], bundle["actor"]["linear2.weight"])


def _crash_once_agent_worker(*args):
    """Explorer that dies on its first start (marker file) and runs normally afterwards."""
    from d4pg_pytorch_amd import engine as eng
    marker = os.path.join(args[0]["results_path"], "crashed_once")
    if args[4] == 1 and not os.path.exists(marker):
        os.makedirs(os.path.dirname(marker), exist_ok=True)
        open(marker, "w").close()
        os._exit(13)
    eng.agent_worker(*args)


def test_supervisor_restarts_crashed_explorer(tmp_path, monkeypatch):
    from d4pg_pytorch_amd import engine as eng
    monkeypatch.setattr(eng, "agent_worker", _crash_once_agent_worker)
    cfg = dict(TEST_VALUES, env="Pendulum-v0", model="ddpg", results_path=str(tmp_path / "results"),
               num_steps_train=60, batch_size=64)
    e = Engine(default_config(**cfg))
    # Engine looks the worker up at call time through the module attribute
    exp = e.train()
    assert e.restarts == 1 and e.final_update_step == 60
    assert os.path.exists(os.path.join(str(tmp_path / "results"), "crashed_once"))


def _failing_learner_worker(config, ctrl_name, *rest, **kw):
    raise RuntimeError("injected learner failure")


def test_learner_failure_tears_the_job_down(tmp_path, monkeypatch):
    from d4pg_pytorch_amd import engine as eng
    monkeypatch.setattr(eng, "learner_worker", _failing_learner_worker)
    cfg = dict(TEST_VALUES, env="Pendulum-v0", model="d4pg", results_path=str(tmp_path / "results"))
    e = Engine(default_config(**cfg))
    with pytest.raises(RuntimeError, match="training failed"):
        e.train()
    assert e.exit_codes["learner-0"] != 0
    # no process is left behind and the shared segments are gone
    assert not [f for f in os.listdir("/dev/shm") if f.startswith(f"d4pg_") and str(os.getpid()) in f]


def _hang_once_agent_worker(*args):
    """Explorer that beats once and then never returns from its 'environment' on its first start."""
    import time as _t
    from d4pg_pytorch_amd import engine as eng
    from d4pg_pytorch_amd.parallel.ipc import ControlBlock
    marker = os.path.join(args[0]["results_path"], "hung_once")
    if args[4] == 1 and not os.path.exists(marker):
        os.makedirs(os.path.dirname(marker), exist_ok=True)
        open(marker, "w").close()
        ctrl = ControlBlock.attach(args[1])
        ctrl.heartbeat(args[7], int(_t.time() * 1000))
        _t.sleep(3600)
    eng.agent_worker(*args)


def test_supervisor_replaces_a_hung_explorer(tmp_path, monkeypatch):
    """A process that stops beating without dying is killed and started again; the job completes."""
    from d4pg_pytorch_amd import engine as eng
    monkeypatch.setattr(eng, "agent_worker", _hang_once_agent_worker)
    cfg = dict(TEST_VALUES, env="Pendulum-v0", model="ddpg", results_path=str(tmp_path / "results"),
               num_steps_train=60, batch_size=64, heartbeat_timeout_s=1.5)
    e = Engine(default_config(**cfg))
    e.train()
    assert e.hung_killed == 1 and e.restarts == 1 and e.final_update_step == 60
    assert all(code == 0 for name, code in e.exit_codes.items()), e.exit_codes


def _hanging_learner_worker(config, ctrl_name, slab_name, ring_names, log_dir, rank, *rest, **kw):
    """Learner rank that comes up, beats, and then blocks for good (a peer lost inside a collective)."""
    import time as _t
    from d4pg_pytorch_amd.parallel.ipc import ControlBlock
    ctrl = ControlBlock.attach(ctrl_name)
    ctrl.store(ControlBlock.LEARNER_ALIVE, 1)
    ctrl.heartbeat(rank, int(_t.time() * 1000))
    _t.sleep(3600)


def test_hung_learner_tears_the_job_down(tmp_path, monkeypatch):
    from d4pg_pytorch_amd import engine as eng
    monkeypatch.setattr(eng, "learner_worker", _hanging_learner_worker)
    cfg = dict(TEST_VALUES, env="Pendulum-v0", model="d4pg", results_path=str(tmp_path / "results"),
               heartbeat_timeout_s=1.5)
    e = Engine(default_config(**cfg))
    with pytest.raises(RuntimeError, match="training failed"):
        e.train()
    assert e.exit_codes["learner-0"] != 0                       # it had to be terminated
    assert not [f for f in os.listdir("/dev/shm") if f.startswith("d4pg_") and str(os.getpid()) in f]


def test_supervisor_watches_every_learner_rank():
    """Any rank of a multi-learner job falling silent ends the job, not only rank 0."""
    import multiprocessing as mp
    import time as _t
    from d4pg_pytorch_amd.parallel.ipc import ControlBlock
    e = Engine(default_config(**dict(TEST_VALUES, env="Pendulum-v0", model="d4pg", heartbeat_timeout_s=0.5)))
    ctx = mp.get_context("spawn")
    with ControlBlock.create(n_workers=8) as ctrl:
        procs = {f"learner-{r}": ctx.Process(target=_t.sleep, args=(30,)) for r in range(2)}
        for p in procs.values():
            p.start()
        try:
            e.hb_slots = {"learner-0": 0, "learner-1": 1}
            ctrl.heartbeat(1, int(_t.time() * 1000) - 10_000)            # rank 1 last beat 10 s ago
            stop_beating = _t.time() + 20

            def beat():
                while _t.time() < stop_beating and not ctrl.stop:
                    ctrl.heartbeat(0, int(_t.time() * 1000))             # rank 0 is healthy
                    _t.sleep(0.05)
            import threading
            th = threading.Thread(target=beat)
            th.start()
            t0 = _t.time()
            e._supervise(ctx, procs, {}, ctrl)
            th.join()
            assert ctrl.stop and ctrl.load(ControlBlock.FAILED) == 1 and _t.time() - t0 < 5
        finally:
            for p in procs.values():
                p.kill()
                p.join()
