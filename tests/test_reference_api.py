"""The reference's class-level API surface: segment trees, replay buffer classes and factory,
single-environment wrappers (CPU)."""
import os
import pickle
import warnings

import numpy as np
import pytest

from d4pg_pytorch_amd.replay import (D3PGReplayBuffer, MinSegmentTree, PrioritizedReplayBuffer, ReplayBuffer,
                                     SumSegmentTree, create_replay_buffer)


def test_sum_and_min_trees_match_brute_force():
    rng = np.random.default_rng(0)
    cap = 64
    st, mt = SumSegmentTree(cap), MinSegmentTree(cap)
    ref = np.zeros(cap)
    for _ in range(40):
        i, v = int(rng.integers(cap)), float(rng.random()) + 0.01
        st[i] = v
        mt[i] = v
        ref[i] = v
    idx = rng.integers(0, cap, 16)
    val = rng.random(16) + 0.01
    st[idx] = val
    mt[idx] = val
    for i, v in zip(idx, val):     # duplicates: last wins
        ref[i] = v
    assert st.sum() == pytest.approx(ref.sum())
    assert st.sum(5, 40) == pytest.approx(ref[5:40].sum())
    assert st.sum(5, -3) == pytest.approx(ref[5:cap - 3].sum())
    refmin = np.where(ref > 0, ref, np.inf)
    assert mt.min() == pytest.approx(refmin.min())
    assert mt.min(10, 30) == pytest.approx(refmin[10:30].min())
    assert st[int(idx[0])] == pytest.approx(ref[idx[0]])
    # prefix-sum search == cumulative-sum search
    cs = np.cumsum(ref)
    masses = rng.random(200) * ref.sum() * 0.999
    want = np.searchsorted(cs, masses, side="right")
    got = st.find_prefixsum_idx(masses)
    np.testing.assert_array_equal(got, want)
    assert st.find_prefixsum_idx(float(masses[0])) == want[0]
    with pytest.raises(AssertionError):
        SumSegmentTree(48)


def test_remove_items_shifts_leaves():
    st = SumSegmentTree(16)
    st[np.arange(10)] = np.arange(1, 11, dtype=float)
    st.remove_items(3)
    np.testing.assert_allclose(st[np.arange(16)], list(range(4, 11)) + [0] * 9)
    assert st.sum() == pytest.approx(sum(range(4, 11)))
    mt = MinSegmentTree(8)
    mt[np.arange(4)] = [5.0, 1.0, 7.0, 3.0]
    mt.remove_items(2)
    assert mt.min() == 3.0


def _fill(buf, n, S=3, A=1, start=0):
    for i in range(start, start + n):
        buf.add(np.full(S, i, np.float32), np.full(A, -i, np.float32), float(i), np.full(S, i + 1, np.float32),
                i % 7 == 0, 0.95)


def test_uniform_buffer_interface_and_capacity():
    buf = ReplayBuffer(50, seed=1)
    assert len(buf) == 0
    _fill(buf, 80)
    assert len(buf) == 50                      # capacity is enforced
    obs, act, rew, nxt, done, gamma, w, idx = buf.sample(32)
    assert obs.shape == (32, 3) and act.shape == (32, 1) and rew.shape == (32,) and w.shape == (32,)
    assert rew.min() >= 30                     # the 30 oldest rows were overwritten
    np.testing.assert_allclose(obs[:, 0], rew)
    np.testing.assert_allclose(nxt[:, 0], rew + 1)
    np.testing.assert_allclose(act[:, 0], -rew)
    np.testing.assert_allclose(w, 1.0)
    np.testing.assert_allclose(gamma, 0.95)
    buf.remove(10)
    assert len(buf) == 40
    assert buf.sample(64)[2].min() >= 40


def test_prioritized_buffer_follows_priorities_and_weights(tmp_path):
    cfg = {"replay_mem_size": 64, "replay_memory_prioritized": 1, "priority_alpha": 1.0, "random_seed": 3,
           "results_path": str(tmp_path)}
    buf = create_replay_buffer(cfg)
    assert isinstance(buf, PrioritizedReplayBuffer)
    _fill(buf, 64)
    pr = np.full(64, 0.01)
    pr[10] = 5.0
    buf.update_priorities(np.arange(64), pr)
    counts = np.zeros(64)
    for _ in range(40):
        *_, w, idx = buf.sample(32, beta=1.0)
        np.add.at(counts, idx, 1)
        assert w.max() <= 1.0 + 1e-6 and w.min() > 0
        if (idx == 10).any():                  # w_i = (p_min / p_i) ** beta
            assert w[idx == 10][0] == pytest.approx(0.01 / 5.0, rel=1e-4)
    assert counts[10] / counts.sum() == pytest.approx(5.0 / (5.0 + 63 * 0.01), abs=0.05)
    path = buf.dump(str(tmp_path))
    with open(path, "rb") as fh:
        assert pickle.load(fh)["rows"].shape[0] == 64
    assert isinstance(create_replay_buffer({"replay_mem_size": 8, "replay_memory_prioritized": 0}), ReplayBuffer)


def test_d3pg_buffer_interface():
    buf = D3PGReplayBuffer(4, 2, max_size=16, seed=0)
    for i in range(20):
        buf.add(np.full(4, i), np.full(2, i), np.full(4, i + 1), float(i), i == 19)
    assert len(buf) == 16
    s, a, r, s2, nd = buf.sample(8)
    assert s.shape == (8, 4) and a.shape == (8, 2) and r.shape == (8, 1) and nd.shape == (8, 1)
    np.testing.assert_allclose(s2[:, 0], r[:, 0] + 1)
    assert set(np.unique(nd)) <= {0.0, 1.0}


def test_single_env_wrappers():
    from d4pg_pytorch_amd.envs.wrappers import BipedalWalker, EnvWrapper, LunarLanderContinous, PendulumWrapper
    env = PendulumWrapper(seed=0)
    obs = env.reset()
    assert obs.shape == (3,)
    o2, r, d = env.step(env.get_random_action())
    assert o2.shape == (3,) and isinstance(r, float) and isinstance(d, bool)
    assert env.normalise_reward(-100.0) == pytest.approx(-1.0)      # reward / 100
    assert env.get_action_space().high[0] == pytest.approx(2.0)
    assert env.render().ndim == 3
    env.close()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")       # no Box2D here: the shape-only stand-in is used
        ll, bw = LunarLanderContinous(seed=1), BipedalWalker(seed=1)
    assert ll.reset().shape == (8,) and bw.reset().shape == (24,)
    assert ll.normalise_reward(50.0) == pytest.approx(0.5) and bw.normalise_reward(50.0) == pytest.approx(50.0)
    with pytest.raises(ValueError):
        EnvWrapper("Hopper-v2")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        assert EnvWrapper("Hopper-v2", {"state_dim": 11, "action_dim": 3}).reset().shape == (11,)


def test_nn_module_front_ends_round_trip_with_the_flat_buffers(tmp_path):
    """PolicyNetwork / ValueNetwork as ``nn.Module``s (the reference's classes): same attribute names and
    methods; weights move to and from the learner's flat buffer and the checkpoints on disk."""
    import torch
    from d4pg_pytorch_amd.config import default_config
    from d4pg_pytorch_amd.models import PolicyNetwork, ValueNetwork
    from d4pg_pytorch_amd.models.core import build_core
    from d4pg_pytorch_amd.utils.checkpoint import save_actor
    cfg = default_config(env="LunarLanderContinuous-v2", model="d4pg", device="cpu", dense_size=48, batch_size=16,
                         v_min=-5.0, v_max=5.0, fused_learner="torch")
    core = build_core(cfg, "cpu")
    S, A, Hd = 8, 2, 48
    pol = PolicyNetwork(S, A, Hd, device="cpu").load_view(core.online.actor)
    val = ValueNetwork(S, A, Hd, -5.0, 5.0, 51, device="cpu").load_view(core.online.critic)
    assert isinstance(pol, torch.nn.Module) and [n for n, _ in pol.named_children()] == ["linear1", "linear2", "linear3"]
    s, a = torch.randn(6, S), torch.rand(6, A) * 2 - 1
    torch.testing.assert_close(pol(s), core.online.actor.forward(s, "tanh"))
    torch.testing.assert_close(val(s, a), core.online.critic.forward(torch.cat([s, a], 1), "none"))
    probs = val.get_probs(s, a)
    torch.testing.assert_close(probs.sum(1), torch.ones(6))
    assert val.z_atoms.shape == (51,) and val.z_atoms[0] == -5.0 and val.z_atoms[-1] == 5.0
    torch.testing.assert_close(val.expected_q(s, a), (probs * torch.linspace(-5, 5, 51)).sum(1))
    act = pol.get_action(np.zeros(S, np.float32))
    assert act.shape == (1, A) and act.abs().max() <= 1
    # module -> flat buffer
    with torch.no_grad():
        pol.linear2.bias.add_(0.25)
    pol.store_view(core.online.actor)
    torch.testing.assert_close(core.online.actor.biases[1], pol.linear2.bias.detach())
    # checkpoint on disk -> module
    path = save_actor(str(tmp_path / "agent_0" / "ckpt.pt"), core.online.actor.state_dict(), {"env": cfg["env"]})
    loaded = PolicyNetwork.from_checkpoint(path)
    torch.testing.assert_close(loaded(s), pol(s))
    # scalar critic (D3PG / DDPG), with the reference's unused batch-norm member for key compatibility
    q = ValueNetwork(S, A, Hd, device="cpu", batch_norm=True)
    assert q(s, a).shape == (6, 1) and q.expected_q(s, a).shape == (6,)
    assert {"bn1.weight", "linear3.bias"} <= set(q.state_dict())
    with pytest.raises(ValueError):
        q.get_probs(s, a)
    with pytest.raises(ValueError):
        ValueNetwork(S, A, Hd, num_atoms=51)
    fresh = PolicyNetwork(S, A, Hd, init_w=3e-3, device="cpu")
    assert fresh.linear3.weight.abs().max() <= 3e-3 and fresh.linear3.bias.abs().max() <= 3e-3
