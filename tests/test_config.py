import glob
import os
import warnings

import pytest
import yaml

from d4pg_pytorch_amd.config import (ConfigError, DEFAULTS, apply_overrides, beta_at, default_config, delta_z,
                                     finalize_config, read_config)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OURS = sorted(glob.glob(os.path.join(ROOT, "configs", "*.yml")))
REF = sorted(glob.glob("/root/reference/configs/*.yml"))


def test_thirty_shipped_configs_load():
    assert len(OURS) == 30
    for f in OURS:
        with warnings.catch_warnings():
            warnings.simplefilter("error")          # shipped files must be clean
            cfg = read_config(f)
        assert set(DEFAULTS) <= set(cfg)
        assert cfg["model"] in ("d4pg", "d3pg", "ddpg")
        assert cfg["num_agents"] >= 2 and cfg["state_dim"] > 0 and cfg["action_dim"] > 0


@pytest.mark.skipif(not REF, reason="reference tree not mounted")
def test_reference_configs_load_unchanged_and_are_repaired():
    assert len(REF) == 30
    for f in REF:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            cfg = read_config(f)
        raw = yaml.safe_load(open(f))
        for k, v in raw.items():
            if k in ("state_dim", "num_agents"):
                continue
            assert cfg[k] == v or float(cfg[k]) == float(v), (f, k)
    with pytest.warns(UserWarning):
        hopper = read_config("/root/reference/configs/hopper_d4pg.yml")
    assert hopper["state_dim"] == 11                  # file says 1
    with pytest.warns(UserWarning):
        walker = read_config("/root/reference/configs/walker2d_d4pg.yml")
    assert walker["num_agents"] == 2                  # file says 1: no explorer, no data


@pytest.mark.skipif(not REF, reason="reference tree not mounted")
def test_shipped_values_equal_reference_values():
    allowed = {("hopper_d4pg.yml", "state_dim"), ("walker2d_d4pg.yml", "num_agents")}
    for f in REF:
        a = yaml.safe_load(open(f))
        b = yaml.safe_load(open(os.path.join(ROOT, "configs", os.path.basename(f))))
        assert set(a) == set(b), os.path.basename(f)
        for k, v in a.items():
            if (os.path.basename(f), k) in allowed:
                continue
            assert (b[k] == v) or float(b[k]) == float(v), (f, k)


def test_mi355x_configs_load():
    files = sorted(glob.glob(os.path.join(ROOT, "configs", "mi355x", "*.yml")))
    assert len(files) >= 6
    for f in files:
        cfg = read_config(f)
        assert cfg["replay_device"] in ("hbm", "host")
    assert read_config(os.path.join(ROOT, "configs", "mi355x", "pendulum_d4pg_cpu.yml"))["replay_device"] == "host"
    assert read_config(os.path.join(ROOT, "configs", "mi355x", "walker2d_d4pg_dp8.yml"))["num_learners"] == 8


def test_unknown_model_raises_instead_of_returning_none():
    with pytest.raises(ConfigError):
        default_config(model="sac")
    from d4pg_pytorch_amd.engine import load_engine
    with pytest.raises(ConfigError):
        load_engine({"env": "Pendulum-v0", "model": "TEST"})


def test_overrides_coercion_and_auto_values():
    raw = {"env": "Pendulum-v0", "model": "DDPG", "num_steps_train": "100_000", "device": "cpu"}
    apply_overrides(raw, ["batch_size=64", "tau=0.01", "use_hip_graph=false", "replay_memory_prioritized=1"])
    cfg = finalize_config(raw)
    assert cfg["model"] == "ddpg" and cfg["num_steps_train"] == 100000 and cfg["batch_size"] == 64
    assert cfg["tau"] == 0.01 and cfg["use_hip_graph"] is False and cfg["replay_memory_prioritized"] == 1
    assert (cfg["state_dim"], cfg["action_dim"]) == (3, 1)
    assert cfg["replay_device"] == "host" and cfg["min_replay_size"] == 64
    assert default_config(device="cuda")["replay_device"] == "hbm"
    with pytest.raises(ConfigError):
        apply_overrides({}, ["novalue"])
    with pytest.raises(ConfigError):
        default_config(env="NoSuchEnv-v0")
    with pytest.raises(ConfigError):
        default_config(v_min=5.0, v_max=1.0)
    with pytest.raises(ConfigError):
        default_config(critic_loss="hinge")


def test_beta_schedule_and_delta_z():
    cfg = default_config(num_steps_train=1000)
    assert beta_at(cfg, 0) == pytest.approx(0.4)
    assert beta_at(cfg, 500) == pytest.approx(0.7)
    assert beta_at(cfg, 5000) == pytest.approx(1.0)
    assert delta_z(default_config(v_min=-10, v_max=10, num_atoms=51)) == pytest.approx(0.4)


def test_reference_test_config_shape_loads(tmp_path):
    # the reference's tests mutate tests/config_test.yml in code (tests/test_pendulum.py:12-15)
    p = tmp_path / "config_test.yml"
    p.write_text("env: TEST\nstate_dim: 0\naction_dim: 0\nmodel: TEST\nnum_agents: 2\ndevice: cpu\n"
                 "dense_size: 100\nnum_steps_train: 100\nreplay_mem_size: 1_000\nv_min: 0\nv_max: 10\n")
    raw = read_config(str(p), finalize=False)
    raw.update(env="Pendulum-v0", model="d4pg", state_dim=3, action_dim=1)
    cfg = finalize_config(raw)
    assert cfg["dense_size"] == 100 and cfg["replay_mem_size"] == 1000


def test_rollout_worker_config_loads_and_is_checked_against_the_machine():
    """configs/mi355x/walker2d_d4pg_rollout_workers_8gpu.yml: learner on GPU 0, a rollout worker on each of GPUs 1 - 7; the
    engine refuses devices the machine does not have before it spawns anything."""
    import os

    import pytest
    import torch
    from d4pg_pytorch_amd.config import read_config
    from d4pg_pytorch_amd.engine import load_engine
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = read_config(os.path.join(root, "configs", "mi355x", "walker2d_d4pg_rollout_workers_8gpu.yml"))
    assert cfg["rollout_devices"] == [1, 2, 3, 4, 5, 6, 7] and cfg["gpu_rollout_envs"] == 1024
    assert read_config(os.path.join(root, "configs", "mi355x", "walker2d_d4pg_rollout_workers_8gpu.yml"),
                       overrides=["rollout_devices=[2,3]"])["rollout_devices"] == [2, 3]
    if torch.cuda.device_count() < 8:
        with pytest.raises(ValueError, match="rollout_devices"):
            load_engine(dict(cfg, num_steps_train=10, results_path="/tmp/d4pg_never_written")).train()


def test_rollout_workers_are_dealt_to_the_learner_ranks_round_robin():
    from d4pg_pytorch_amd.actors.rollout_worker import worker_assignment, workers_of_rank
    from d4pg_pytorch_amd.parallel.row_mailbox import mailbox_path
    assert [worker_assignment(k, 1) for k in range(3)] == [(0, 0), (0, 1), (0, 2)]
    assert [worker_assignment(k, 2) for k in range(5)] == [(0, 0), (1, 0), (0, 1), (1, 1), (0, 2)]
    assert [workers_of_rank(5, 2, r) for r in range(2)] == [3, 2] and workers_of_rank(1, 2, 1) == 0
    assert mailbox_path("/x", 0).endswith("x.rows.hipipc") and mailbox_path("/x", 3).endswith("x.rows3.hipipc")
