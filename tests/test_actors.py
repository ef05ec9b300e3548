"""OU noise, n-step assembly and CPU actor inference (C++ host runtime) against
closed forms / the torch oracles."""
import numpy as np
import pytest
import torch

from d4pg_pytorch_amd.actors.native import CpuActor, NStepAssembler, OUNoise
from d4pg_pytorch_amd.models.nets import ActorPolicy
from d4pg_pytorch_amd.ops import ref
from d4pg_pytorch_amd.replay.hbm_replay import row_ld


def test_ou_noise_statistics_reset_and_clipping():
    E, A = 64, 3
    ou = OUNoise(E, A, -1.0, 1.0, seed=1)
    assert np.all(ou.state == 0.0)
    xs = []
    for t in range(4000):
        ou.get_action(np.zeros((E, A), np.float32), t)
        xs.append(ou.state)
    x = np.stack(xs[500:])
    # stationary AR(1): var = sigma^2 / (1 - (1-theta)^2), lag-1 correlation = 1 - theta
    var_want = 0.3 ** 2 / (1 - 0.85 ** 2)
    assert abs(x.mean()) < 0.02
    assert x.var() == pytest.approx(var_want, rel=0.05)
    x0, x1 = x[:-1].ravel(), x[1:].ravel()
    assert np.corrcoef(x0, x1)[0, 1] == pytest.approx(0.85, abs=0.01)
    ou.reset(5)
    st = ou.state
    assert np.all(st[5] == 0.0) and np.any(st[6] != 0.0)
    ou.reset()
    assert np.all(ou.state == 0.0)
    out = OUNoise(4, 2, -0.1, 0.1, max_sigma=5.0, min_sigma=5.0, seed=2).get_action(np.zeros((4, 2)), 0)
    assert out.min() >= -0.1 and out.max() <= 0.1
    # different seeds -> different streams; same seed -> identical
    a = OUNoise(2, 2, -9, 9, seed=3).get_action(np.zeros((2, 2)), 0)
    b = OUNoise(2, 2, -9, 9, seed=3).get_action(np.zeros((2, 2)), 0)
    c = OUNoise(2, 2, -9, 9, seed=4).get_action(np.zeros((2, 2)), 0)
    assert np.array_equal(a, b) and not np.array_equal(a, c)


def test_ou_sigma_schedule():
    ou = OUNoise(1, 1, -100, 100, theta=0.0, max_sigma=1.0, min_sigma=0.0, decay_period=10, seed=5)
    ou.get_action(np.zeros((1, 1)), 10)        # t >= decay_period -> sigma = min_sigma = 0
    assert ou.state[0, 0] == 0.0


def _run_episode_through_assembler(n, gamma, rewards, terminal, S=3, A=2):
    T = len(rewards)
    asm = NStepAssembler(1, n, S, A, gamma)
    rows = []
    for t in range(T):
        s = np.full((1, S), t, np.float32)
        a = np.full((1, A), -t, np.float32)
        ns = np.full((1, S), t + 1, np.float32)
        last = t == T - 1
        rows.append(asm.push(s, a, [rewards[t]], ns, [last and terminal], [last and not terminal]))
    return np.concatenate(rows), asm


@pytest.mark.parametrize("n,T,terminal", [(5, 12, True), (5, 12, False), (5, 3, True), (1, 6, True), (3, 3, False)])
def test_nstep_assembler_matches_reference_semantics(n, T, terminal):
    gamma, S, A = 0.9, 3, 2
    rng = np.random.default_rng(0)
    rewards = rng.standard_normal(T).astype(np.float32)
    rows, asm = _run_episode_through_assembler(n, gamma, rewards, terminal)
    assert rows.shape == (T, row_ld(S, A))        # every step yields exactly one transition
    done = torch.zeros(T, 1)
    trunc = torch.zeros(T, 1)
    (done if terminal else trunc)[T - 1] = 1.0
    R, gk, nxt, d = ref.nstep_assemble(torch.tensor(rewards).reshape(T, 1), done, n, gamma, trunc)
    order = np.argsort(rows[:, 0])                # state column carries the step index
    rows = rows[order]
    np.testing.assert_allclose(rows[:, S + A], R[:, 0].numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(rows[:, S + A + 2], gk[:, 0].numpy(), rtol=1e-6)
    np.testing.assert_array_equal(rows[:, S + A + 1], d[:, 0].numpy())
    np.testing.assert_array_equal(rows[:, S + A + 3], (nxt[:, 0] + 1).numpy().astype(np.float32))  # bootstrap state
    np.testing.assert_array_equal(rows[:, S], -np.arange(T, dtype=np.float32))                      # action of step t
    # a full window only emits once n steps exist (models/agent.py:93)
    asm2 = NStepAssembler(1, 4, S, A, gamma)
    for t in range(3):
        out = asm2.push(np.zeros((1, S)), np.zeros((1, A)), [1.0], np.zeros((1, S)), [False], [False])
        assert out.shape[0] == 0
    assert asm2.push(np.zeros((1, S)), np.zeros((1, A)), [1.0], np.zeros((1, S)), [False], [False]).shape[0] == 1


def test_nstep_vectorised_envs_are_independent():
    asm = NStepAssembler(3, 2, 1, 1, 0.5)
    s = np.array([[0.0], [10.0], [20.0]], np.float32)
    out = asm.push(s, s, [1.0, 1.0, 1.0], s + 1, [False, True, False], None)
    assert out.shape[0] == 1 and out[0, 0] == 10.0 and out[0, 3] == 1.0 and out[0, 4] == 0.5
    out = asm.push(s + 1, s, [2.0, 2.0, 2.0], s + 2, [False, False, False], None, active=[1, 0, 1])
    assert sorted(out[:, 0].tolist()) == [0.0, 20.0]
    np.testing.assert_allclose(out[:, 2], [1.0 + 0.5 * 2.0] * 2)


def test_cpu_actor_matches_torch_mlp_and_roundtrips_state_dict():
    S, A, Hd = 17, 6, 400
    pol = ActorPolicy(S, A, Hd)
    from d4pg_pytorch_amd.models.nets import init_net_
    init_net_(pol.view, 0.5, torch.Generator().manual_seed(0))
    cpu = CpuActor(S, A, Hd)
    cpu.load_flat(pol.flat.numpy())
    x = np.random.default_rng(0).standard_normal((9, S)).astype(np.float32)
    np.testing.assert_allclose(cpu.act(x), pol.act(x).numpy(), rtol=1e-4, atol=1e-5)
    assert cpu.act(x[0]).shape == (1, A)
    sd = cpu.state_dict()
    other = CpuActor(S, A, Hd, action_scale=2.0)
    other.load_state_dict(sd)
    np.testing.assert_allclose(other.act(x), 2.0 * cpu.act(x), rtol=1e-6)


def test_step_counter_slots_alternate_and_rewind_to_slot_zero():
    """The Philox position of a device environment (envs/device_pendulum.py::StepCounter): unpaired, every launch reads
    and a follow-up launch increments ONE slot; paired, launches alternate between two slots and the environment's own
    kernel writes the successor, so a captured sequence has to end on the slot it started from."""
    from types import SimpleNamespace

    from d4pg_pytorch_amd.envs.device_pendulum import StepCounter
    c = StepCounter("cpu")
    base = c.slots.data_ptr()
    a = SimpleNamespace(counter=None, counter_next=None)
    c.fill(a)
    assert (a.counter, a.counter_next, c.cur) == (base, None, 0)            # unpaired: in place
    c.paired = True
    seen = []
    for _ in range(4):
        a = SimpleNamespace(counter=None, counter_next=None)
        c.fill(a)
        seen.append((a.counter - base, a.counter_next - base))
    assert seen == [(0, 8), (8, 0), (0, 8), (8, 0)] and c.cur == 0
    assert c.ptr == base
    c.rewind()                                                               # on slot 0 already: no launch, nothing to load
    c.fill(SimpleNamespace(counter=None, counter_next=None))
    assert c.cur == 1 and c.ptr == base + 8
