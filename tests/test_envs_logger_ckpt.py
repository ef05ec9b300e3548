import os
import sys
import types

import numpy as np
import pytest
import torch

from d4pg_pytorch_amd.config import default_config
from d4pg_pytorch_amd.envs import CartPoleContinuousVec, PendulumVec, SyntheticVec, create_env_wrapper
from d4pg_pytorch_amd.utils import checkpoint as ck
from d4pg_pytorch_amd.utils.logger import Logger, crc32c, read_events
from d4pg_pytorch_amd.utils.misc import frames_to_gif, make_gif


# ------------------------------------------------------------------ envs
def test_pendulum_dynamics_match_closed_form_step():
    env = PendulumVec(2, seed=0)
    obs = env.reset()
    assert obs.shape == (2, 3) and np.allclose(obs[:, 0] ** 2 + obs[:, 1] ** 2, 1.0, atol=1e-6)
    env.theta[:] = [np.pi, 0.3]
    env.theta_dot[:] = [0.0, -1.0]
    nxt, rew, done = env.step(np.array([[5.0], [-0.5]]))      # first action is clipped to +2
    th, thd, u = np.array([np.pi, 0.3]), np.array([0.0, -1.0]), np.array([2.0, -0.5])
    ang = ((th + np.pi) % (2 * np.pi)) - np.pi
    cost = ang ** 2 + 0.1 * thd ** 2 + 0.001 * u ** 2
    new_thd = thd + (-3 * 10.0 / 2 * np.sin(th + np.pi) + 3.0 * u) * 0.05
    new_th = th + new_thd * 0.05
    np.testing.assert_allclose(rew, -cost, rtol=1e-6)
    np.testing.assert_allclose(nxt, np.stack([np.cos(new_th), np.sin(new_th), new_thd], 1), rtol=1e-5, atol=1e-6)
    assert not done.any()
    assert env.normalise_reward(-100.0) == -1.0                # reward / 100 (env/pendulum.py:13-14)
    for _ in range(199):
        _, _, done = env.step(np.zeros((2, 1)))
    assert done.all()                                          # 200-step time limit reported as done
    o = env.reset_env(1)
    assert env.t[1] == 0 and env.t[0] == 200 and o.shape == (3,)
    assert env.render(0).shape == (128, 128, 3)
    a = env.get_random_action()
    assert a.shape == (2, 1) and np.all(np.abs(a) <= 2.0)


def test_factory_resolution_and_synthetic_shapes():
    import warnings
    cfg = default_config(env="Pendulum-v0", model="d4pg", device="cpu")
    assert isinstance(create_env_wrapper(cfg, 3), PendulumVec)
    cfg = default_config(env="InvertedPendulum-v2", model="d3pg", device="cpu")
    env = create_env_wrapper(cfg, 2)
    assert isinstance(env, CartPoleContinuousVec)
    obs = env.reset()
    for _ in range(400):
        obs, r, d = env.step(np.ones((2, 1)))       # constant push -> the pole falls
        if d.any():
            break
    assert d.any() and np.all(r == 1.0)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        env = create_env_wrapper(default_config(env="Humanoid-v2", model="d4pg", device="cpu", state_dim=111, action_dim=8), 4, seed=0)
    assert isinstance(env, SyntheticVec) and any("synthetic" in str(x.message) for x in w)
    obs = env.reset()
    nxt, r, d = env.step(env.get_random_action())
    assert obs.shape == (4, 111) and nxt.shape == (4, 111) and r.shape == (4,) and d.dtype == bool
    assert np.all(np.abs(nxt) <= 1.0)
    env2 = create_env_wrapper(default_config(env="LunarLanderContinuous-v2", model="d4pg", device="cpu"), 1)
    assert env2.normalise_reward(50.0) == 0.5                  # env/lunar_lander_continous.py:12-13


# ------------------------------------------------------------------ logger
def test_crc32c_known_vectors():
    assert crc32c(b"") == 0
    assert crc32c(b"123456789") == 0xE3069283
    assert crc32c(b"\x00" * 32) == 0x8A9136AA


def test_event_file_roundtrip(tmp_path):
    lg = Logger(str(tmp_path / "learner"), backend="both")
    for s in range(50):
        lg.scalar_summary("learner/value_loss", 0.5 / (s + 1), s)
    lg.scalar_summary("agent/reward", -123.25, 7)
    lg.close()
    files = os.listdir(tmp_path / "learner")
    assert any("tfevents" in f for f in files) and "scalars.jsonl" in files
    ev = read_events(str(tmp_path / "learner"))
    assert len(ev) == 51
    assert ev[0][1:] == (0, "learner/value_loss", 0.5)
    assert ev[-1][1:] == (7, "agent/reward", -123.25)
    assert all(e[0] > 1e9 for e in ev)
    # the event file opens with the version record TensorBoard expects
    raw = open(os.path.join(tmp_path, "learner", [f for f in files if "tfevents" in f][0]), "rb").read()
    assert b"brain.Event:2" in raw[:64]
    # a flipped bit is detected
    bad = bytearray(raw)
    bad[-10] ^= 0xFF
    p = tmp_path / "bad.tfevents"
    p.write_bytes(bytes(bad))
    with pytest.raises(ValueError):
        read_events(str(p))


# ------------------------------------------------------------------ gif
def test_gif_helpers(tmp_path):
    from PIL import Image
    frames = [np.full((16, 16, 3), i * 40, np.uint8) for i in range(5)]
    out = frames_to_gif(frames, str(tmp_path / "a.gif"))
    assert Image.open(out).n_frames == 5
    d = tmp_path / "png"
    d.mkdir()
    for i in (10, 2, 1):
        Image.fromarray(frames[i % 5]).save(d / f"{i}.png")
    out2 = make_gif(str(d), str(tmp_path / "b.gif"))          # ordered by integer stem: 1, 2, 10
    assert Image.open(out2).n_frames == 3


# ------------------------------------------------------------------ checkpoints
def test_actor_checkpoint_format_and_path(tmp_path):
    from d4pg_pytorch_amd.models.nets import ActorPolicy, init_net_
    pol = ActorPolicy(3, 1, 32)
    init_net_(pol.view, 3e-3, torch.Generator().manual_seed(0))
    path = ck.actor_checkpoint_path(str(tmp_path), 0, 12, -153.5)
    assert path.endswith(os.path.join("agent_0", "local_episode_12_reward_-153.500000.pt"))
    ck.save_actor(path, pol.state_dict(), {"env": "Pendulum-v0"})
    raw = torch.load(path, weights_only=True)                  # loadable without executing anything
    assert raw["format"] == ck.ACTOR_FORMAT and list(raw["state_dict"])[0] == "linear1.weight"
    sd = ck.load_actor(path)
    other = ActorPolicy(3, 1, 32)
    other.load_state_dict(sd)
    torch.testing.assert_close(other.flat, pol.flat)
    # a bare state_dict file works too
    torch.save(pol.state_dict(), tmp_path / "bare.pt")
    assert "linear3.bias" in ck.load_actor(str(tmp_path / "bare.pt"))


def test_legacy_whole_module_pickle_import(tmp_path):
    """A file written like the reference's ``torch.save(self.actor, path)``
    (models/agent.py:161) is read without importing the reference's modules."""
    pkg = types.ModuleType("models")
    sub = types.ModuleType("models.d4pg")
    mod = types.ModuleType("models.d4pg.networks")

    class PolicyNetwork(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.device = "cpu"
            self.linear1 = torch.nn.Linear(3, 16)
            self.linear2 = torch.nn.Linear(16, 16)
            self.linear3 = torch.nn.Linear(16, 1)

    PolicyNetwork.__module__ = "models.d4pg.networks"
    PolicyNetwork.__qualname__ = "PolicyNetwork"
    mod.PolicyNetwork = PolicyNetwork
    sys.modules.update({"models": pkg, "models.d4pg": sub, "models.d4pg.networks": mod})
    try:
        net = PolicyNetwork()
        path = tmp_path / "legacy.pt"
        torch.save(net, path)
    finally:
        for k in ("models", "models.d4pg", "models.d4pg.networks"):
            sys.modules.pop(k, None)
    sd = ck.import_legacy_actor(str(path))
    assert set(sd) == {"linear1.weight", "linear1.bias", "linear2.weight", "linear2.bias", "linear3.weight", "linear3.bias"}
    torch.testing.assert_close(sd["linear2.weight"], net.linear2.weight.detach())
    with pytest.raises(Exception):                              # the legacy reader is opt-in
        ck.load_actor(str(path))
    assert "linear1.weight" in ck.load_actor(str(path), allow_legacy=True)


def test_legacy_importer_refuses_arbitrary_classes(tmp_path):
    """Pickles that name a callable — directly, through a dotted name under an innocent module, or under the
    torch namespace — are refused before anything runs."""
    import pickle
    marker = tmp_path / "ran"

    class Evil:
        def __reduce__(self):
            return (os.system, (f"touch {marker}",))

    class EvilHub:
        def __reduce__(self):
            return (torch.hub.load, ("nowhere/nothing", "model"))

    cases = {"os_system.pt": pickle.dumps(Evil()), "torch_hub.pt": pickle.dumps(EvilHub()),
             # protocol 4 resolves dotted names: ('torch.serialization', 'os.system') is os.system
             "dotted.pt": b"\x80\x04\x8c\x13torch.serialization\x8c\tos.system\x93\x8c\x04true\x85R.",
             "getattr.pt": b"\x80\x04\x8c\x08builtins\x8c\x07getattr\x93."}
    for name, blob in cases.items():
        p = tmp_path / name
        p.write_bytes(blob)
        with pytest.raises(Exception):
            ck.import_legacy_actor(str(p))
        with pytest.raises(Exception):
            ck.load_actor(str(p), allow_legacy=True)
    # the same payloads inside a torch zip archive
    for name, obj in (("zip_os.pt", Evil()), ("zip_hub.pt", EvilHub())):
        p = tmp_path / name
        torch.save(obj, p)
        with pytest.raises(Exception):
            ck.import_legacy_actor(str(p))
    assert not marker.exists()
