"""Seeded learning thresholds on the GPU: the fused HIP cores must actually solve the pendulum swing-up
(in-process loop with device-resident explorers, ``utils/learning.py``; a random policy scores about -1200,
a good one -150 .. -250).  The multi-seed comparison against the eager PyTorch core is
``bench/learn_parity.py`` (table in ``profiles/learning/README.md``)."""
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("core", ["fp32", "bf16"])
def test_d3pg_learns_the_pendulum(core):
    from d4pg_pytorch_amd.utils.learning import train_pendulum_on_device
    r = train_pendulum_on_device("d3pg", core, updates=10_000, seed=100, eval_every=1000)
    assert r["initial_return"] < -1000
    assert r["final_return"] > -400 and r["last5_mean_return"] > -450, r["curve"]


def test_d4pg_fp32_learns_the_pendulum_within_20k_updates():
    from d4pg_pytorch_amd.models.hip_core_f32 import HipCoreF32  # noqa: F401  (the default core)
    from d4pg_pytorch_amd.utils.learning import train_pendulum_on_device
    r = train_pendulum_on_device("d4pg", "fp32", updates=20_000, seed=103, eval_every=1000)
    assert r["core_class"] == "HipCoreF32" and r["initial_return"] < -1000
    assert r["best_return"] > -400 and r["final_return"] > -400, r["curve"]


def test_same_seed_same_run():
    """Everything in the loop is seeded and on the device (Philox streams for sampling, noise and resets):
    two runs make the same curve."""
    from d4pg_pytorch_amd.utils.learning import train_pendulum_on_device
    a = train_pendulum_on_device("d3pg", "fp32", updates=1500, seed=5, eval_every=500)
    b = train_pendulum_on_device("d3pg", "fp32", updates=1500, seed=5, eval_every=500)
    assert a["curve"] == b["curve"]


def test_d3pg_learns_to_balance_the_device_cartpole():
    """A task with terminal states (the pole falls): random actions last ~25 steps, a trained policy hundreds."""
    from d4pg_pytorch_amd.utils.learning import train_pendulum_on_device
    r = train_pendulum_on_device("d3pg", "fp32", updates=6000, seed=100, eval_every=1000, env="InvertedPendulum-v2",
                                 max_ep_length=500)
    assert r["initial_return"] < 120
    assert r["best_return"] > 250, r["curve"]


def test_d4pg_learns_to_land_the_lunar_lander():
    """The reference's second benchmark task (``configs/lunar_lander_continuous_d4pg.yml``, README curve) on the
    native lander with the reference's hyper-parameters: a fresh policy crashes (about -100 .. -300), a trained
    one lands (+100 bonus; >= 200 is the conventional solved level of the task)."""
    from d4pg_pytorch_amd.utils.learning import train_pendulum_on_device
    r = train_pendulum_on_device("d4pg", "fp32", updates=12_000, seed=200, eval_every=2000, envs=16,
                                 env="LunarLanderContinuous-v2", max_ep_length=1000)
    assert r["initial_return"] < 0
    assert r["best_return"] > 200 and r["final_return"] > 150, r["curve"]


def test_d4pg_learns_the_device_bipedal_walker_and_half_cheetah():
    """The reference's third benchmark task (``configs/bipedal_d4pg.yml``, README curve) and a MuJoCo-style one on the
    device-resident articulated-body simulator (``csrc/planar_env.hip``), the reference's hyper-parameters: a fresh
    walker falls over (-100), a trained one walks most of the track (reaching its end is worth about 300); a
    fresh cheetah does not move (about 0), a trained one runs."""
    from d4pg_pytorch_amd.utils.learning import train_pendulum_on_device
    r = train_pendulum_on_device("d4pg", "fp32", updates=60_000, seed=300, eval_every=10_000, envs=256,
                                 env="BipedalWalker-v2", max_ep_length=1600, v_min=-100.0, v_max=300.0)
    assert r["initial_return"] < -90
    assert r["best_return"] > 200, r["curve"]
    r = train_pendulum_on_device("d4pg", "fp32", updates=40_000, seed=300, eval_every=10_000, envs=256,
                                 env="HalfCheetah-v2", max_ep_length=1000, v_min=-50.0, v_max=600.0)
    assert abs(r["initial_return"]) < 300
    assert r["best_return"] > 2000, r["curve"]
