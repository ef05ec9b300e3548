"""Replicated data parallelism (parallel/replicated.py) on one GPU: world 1 against the plain
replay path, two thread-ranks with an in-process all-gather against a single learner that is
fed the same global batches, and a world-1 RCCL group for the collective wiring (GPU only)."""
import os
import threading

import numpy as np
import pytest
import torch

from d4pg_pytorch_amd.config import default_config

pytestmark = pytest.mark.gpu

GOLD = 0x9E3779B97F4A7C15


def _cfg(B, per=0, dtype="bf16", graph=True, model="d4pg"):
    return default_config(env="BipedalWalker-v2", model=model, batch_size=B, device="cuda", v_min=-10.0, v_max=10.0,
                          replay_mem_size=40_000, replay_memory_prioritized=per, num_steps_train=1000,
                          compute_dtype=dtype, use_hip_graph=graph, random_seed=7)


def _rows(rp, n, seed):
    g = torch.Generator().manual_seed(seed)
    rows = torch.randn(n, rp.row_ld, generator=g)
    S, A = rp.S, rp.A
    rows[:, S + A + 1] = (torch.rand(n, generator=g) < 0.05).float()
    rows[:, S + A + 2] = 0.95
    return rows.cuda()


@pytest.mark.parametrize("dtype", ["bf16", "fp32"])
def test_world_one_reproduces_the_plain_replay_path(dtype):
    """plain: one sampling kernel inside every update.  ahead: the core's own draw-ahead path
    (bf16 + uniform replay: 8 batches per sampling launch, rows read in place by the phase
    kernels).  replicated: the data-parallel driver over a world of one.  Same RNG stream
    positions -> same rows -> bit-identical parameters."""
    from d4pg_pytorch_amd.models.core import build_core
    from d4pg_pytorch_amd.parallel.comm import LocalComm
    from d4pg_pytorch_amd.parallel.replicated import attach_replicated
    from d4pg_pytorch_amd.replay.hbm_replay import HbmReplay
    outs = {}
    for mode in ("plain", "ahead", "many", "replicated"):
        cfg = dict(_cfg(256, dtype=dtype), sample_ahead=1 if mode == "plain" else 4)
        core = build_core(cfg, "cuda")
        rp = HbmReplay(cfg, "cuda", core.dev_state)
        rp.insert(_rows(rp, 20_000, 1))
        if mode == "replicated":
            dp = attach_replicated(core, rp, LocalComm(), chunk=4)
            for _ in range(12):
                dp.update()
            assert dp.check_consistency() == 0.0
        else:
            core.attach_replay(rp)
            st = core.read_dev_state()
            core._write_dev_state(step=0, seed=st.seed + GOLD, keep_replay=True)     # the stream rank 0 samples from
            if mode == "many":        # whole chunks of updates as one graph replay, plus a remainder
                done = 0
                while done < 12:
                    done += core.update_from_replay_many(12 - done)
                assert done == 12 and core.step == 12
            else:
                for _ in range(12):
                    core.update_from_replay()
            if dtype == "bf16":
                assert (core._ahead is not None) == (mode != "plain")
        torch.cuda.synchronize()
        m = core.read_metrics()
        assert m["count"] == 12 and np.isfinite(m["value_loss"])
        assert core.read_dev_state().step_done == 12
        outs[mode] = (core.online.flat.clone(), core.target.flat.clone())
    for mode in ("ahead", "many", "replicated"):
        assert torch.equal(outs["plain"][0], outs[mode][0]) and torch.equal(outs["plain"][1], outs[mode][1]), mode


@pytest.mark.parametrize("per", [0, 1])
def test_two_thread_ranks_equal_one_learner_on_the_gathered_batches(per):
    from d4pg_pytorch_amd.models.core import build_core
    from d4pg_pytorch_amd.parallel.fake_comm import ThreadComm
    from d4pg_pytorch_amd.parallel.replicated import attach_replicated
    from d4pg_pytorch_amd.replay.hbm_replay import HbmReplay
    world, Bl, steps = 2, 128, 11                # chunks of 4 batches: crosses two chunk boundaries
    cfg_g = _cfg(Bl * world, per=per, graph=False)          # graph capture is process-global: threads run eagerly
    comms = ThreadComm.group(world)
    cores, dps, seen, errs = [None] * world, [None] * world, [], []

    def run(r):
        try:
            torch.cuda.set_device(0)
            core = build_core(cfg_g, "cuda", seed=100 + r)            # different inits: rank 0 is broadcast
            rp = HbmReplay(cfg_g, "cuda", core.dev_state)
            rp.insert(_rows(rp, 10_000, 10 + r))                      # different shards
            dp = attach_replicated(core, rp, comms[r], Bl, chunk=4)
            cores[r], dps[r] = core, dp
            for _ in range(steps):
                batch = dp.current_batch()            # (a collective on the first call: every rank makes it)
                if r == 0:
                    seen.append(batch)
                dp.update()
            assert dp.check_consistency() == 0.0
        except Exception as e:    # pragma: no cover
            errs.append(e)
            comms[r].shared.barrier.abort()

    ts = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    [t.start() for t in ts]
    [t.join(180) for t in ts]
    assert not errs, errs
    assert torch.equal(cores[0].online.flat, cores[1].online.flat)
    assert torch.equal(cores[0].target.flat, cores[1].target.flat)
    # the two halves of a gathered batch come from the two shards
    S, A = cores[0].S, cores[0].A
    assert not torch.equal(seen[0][:Bl, :S], seen[0][Bl:, :S])
    # a single learner fed the same global batches makes the same updates
    single = build_core(cfg_g, "cuda", seed=100)
    row_ld = dps[0].replay.row_ld
    for b in seen:
        single.update_from_packed(b, row_ld)
    torch.cuda.synchronize()
    assert torch.equal(single.online.flat, cores[0].online.flat)
    assert torch.equal(single.target.flat, cores[1].target.flat)
    if per:
        # each rank wrote back only its own half: priorities of sampled rows moved off the initial value
        for r in range(world):
            leaf = dps[r].replay.leaf[:10_000]
            assert (leaf != leaf.max()).sum().item() >= Bl // 2


def test_rccl_world_one_all_gather_wiring():
    from d4pg_pytorch_amd.models.core import build_core
    from d4pg_pytorch_amd.parallel.comm import TorchDistComm
    from d4pg_pytorch_amd.parallel.replicated import attach_replicated
    from d4pg_pytorch_amd.replay.hbm_replay import HbmReplay
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(30100 + os.getpid() % 100), RANK="0", WORLD_SIZE="1")
    comm = TorchDistComm(device=torch.device("cuda", 0), rank=0, world=1)
    try:
        cfg = _cfg(256, per=1)
        core = build_core(cfg, "cuda")
        rp = HbmReplay(cfg, "cuda", core.dev_state)
        rp.insert(_rows(rp, 20_000, 3))
        dp = attach_replicated(core, rp, comm)
        out = torch.zeros(8, device="cuda")
        comm.all_gather_into_(out, torch.arange(8.0, device="cuda"))
        assert torch.equal(out, torch.arange(8.0, device="cuda"))
        for _ in range(20):
            dp.update()
        torch.cuda.synchronize()
        m = core.read_metrics()
        assert m["count"] == 20 and np.isfinite(m["value_loss"]) and np.isfinite(m["policy_loss"])
        assert dp.check_consistency() == 0.0
    finally:
        comm.close()


def test_critic_forward_made_ahead_is_tagged_and_dropped_when_weights_change(monkeypatch):
    """The actor-phase launch of update u also makes the online critic's forward pass for the batch of
    update u + 1 (bf16 pipeline, drawn-ahead batches).  The hand-off carries a tag in the device state
    (update ordinal + address of the rows); a host-side change of the weights clears it, so the next
    critic phase makes the pass itself.  With and without the hand-off the parameters are bit-identical."""
    from d4pg_pytorch_amd.models.core import build_core
    from d4pg_pytorch_amd.replay.hbm_replay import HbmReplay
    outs = {}
    for mode in ("ahead", "inline"):
        monkeypatch.setenv("D4PG_CRITIC_AHEAD", "1" if mode == "ahead" else "0")
        cfg = dict(_cfg(256), sample_ahead=4)
        core = build_core(cfg, "cuda")
        rp = HbmReplay(cfg, "cuda", core.dev_state)
        rp.insert(_rows(rp, 20_000, 3))
        core.attach_replay(rp)
        for _ in range(5):                      # crosses a chunk boundary (4 batches per chunk)
            core.update_from_replay()
        torch.cuda.synchronize()
        st = core.read_dev_state()
        assert st.step_done == 5
        if mode == "ahead":
            assert core.direct_ahead.get("gathered") == 2
            assert st.pad_[0] == 6 and st.pad_[1] != 0          # tiles for update 6 are waiting
        else:
            assert st.pad_[0] == 0
        with torch.no_grad():                   # the host changes the weights: the tiles are stale
            core.online.flat.mul_(0.97)
        core.refresh_shadows()
        assert core.read_dev_state().pad_[0] == 0
        done = 0
        while done < 7:
            done += core.update_from_replay_many(7 - done)
        torch.cuda.synchronize()
        assert core.read_dev_state().step_done == 12
        outs[mode] = (core.online.flat.clone(), core.target.flat.clone(), core.adam_v.clone())
    for a, b in zip(outs["ahead"], outs["inline"]):
        assert torch.equal(a, b)


def test_prioritised_batch_drawn_by_the_previous_update_equals_the_sampling_kernel(monkeypatch):
    """Prioritised replay, bf16 pipeline: the batch of update u + 1 is drawn in extra workgroups of the
    last launch of update u (after its priority write-back).  Same RNG stream positions and the same tree
    state as a sampling kernel at the start of update u + 1, hence bit-identical parameters and priorities -
    also across a reload of the state in between."""
    from d4pg_pytorch_amd.models.core import build_core
    from d4pg_pytorch_amd.replay.hbm_replay import HbmReplay
    outs = {}
    for mode in ("role", "kernel"):
        monkeypatch.setenv("D4PG_PER_SAMPLE_ROLE", "1" if mode == "role" else "0")
        cfg = _cfg(256, per=1)
        core = build_core(cfg, "cuda")
        rp = HbmReplay(cfg, "cuda", core.dev_state)
        rp.insert(_rows(rp, 20_000, 5))
        core.attach_replay(rp)
        assert core._per_sampling_rides_along() == (mode == "role")
        for _ in range(6):
            core.update_from_replay()
        bundle = core.state_bundle()
        core.load_state_bundle(bundle)             # voids the batch waiting in the input buffers
        rp.insert(_rows(rp, 3_000, 6))             # new rows at the running max priority
        for _ in range(5):
            core.update_from_replay()
        torch.cuda.synchronize()
        assert core.read_dev_state().step_done == 11
        outs[mode] = (core.online.flat.clone(), core.target.flat.clone(), rp.leaf.clone(), core.idx.clone())
    for a, b in zip(outs["role"][:3], outs["kernel"][:3]):
        assert torch.equal(a, b)


def test_odd_update_counts_replay_power_of_two_runs_and_late_captures_drop_the_ahead_tiles():
    """``update_many`` with counts that are not multiples of the chunk, starting in the middle of a chunk,
    and a graph captured after real updates (whose warm-up run overwrites the critic tiles made ahead):
    bit-identical to one update per call on the plain sampling path."""
    from d4pg_pytorch_amd.models.core import build_core
    from d4pg_pytorch_amd.replay.hbm_replay import HbmReplay
    outs = {}
    for mode in ("plain", "runs"):
        cfg = dict(_cfg(256), sample_ahead=1 if mode == "plain" else 8)
        core = build_core(cfg, "cuda")
        rp = HbmReplay(cfg, "cuda", core.dev_state)
        rp.insert(_rows(rp, 20_000, 9))
        core.attach_replay(rp)
        if mode == "plain":
            for _ in range(37):
                core.update_from_replay()
        else:
            for _ in range(3):                      # single-update graph first, tiles for update 4 waiting
                core.update_from_replay()
            made = []
            for want in (13, 1, 20):                # captures the run graphs at step 3; 13 = 4 + 1 + 8, ...
                done = 0
                while done < want:
                    k = core.update_from_replay_many(want - done)
                    made.append(k)
                    done += k
                assert done == want
            assert max(made) == 8 and sum(made) == 34 and len(made) < 12
        torch.cuda.synchronize()
        assert core.read_dev_state().step_done == 37
        outs[mode] = (core.online.flat.clone(), core.target.flat.clone())
    assert torch.equal(outs["plain"][0], outs["runs"][0]) and torch.equal(outs["plain"][1], outs["runs"][1])


def _replicated_per_pendulum_run(tau: float, updates: int, seed: int = 21, per: int = 1):
    """Two thread-ranks, replicated mode: each owns a replay shard fed by its own device explorers.  Returns (the two
    cores, their priority leaves, the greedy return of rank 0's target policy)."""
    from d4pg_pytorch_amd.actors.gpu_rollout import DeviceRollout
    from d4pg_pytorch_amd.envs.device_pendulum import DevicePendulum
    from d4pg_pytorch_amd.models.core import build_core
    from d4pg_pytorch_amd.parallel.fake_comm import ThreadComm
    from d4pg_pytorch_amd.parallel.replicated import attach_replicated
    from d4pg_pytorch_amd.replay.hbm_replay import HbmReplay
    from d4pg_pytorch_amd.utils.learning import GreedyEvaluator, pendulum_config
    world, Bl = 2, 128
    cfg = pendulum_config("d4pg", "fp32", seed, batch_size=Bl * world, replay_memory_prioritized=per, use_hip_graph=False,
                          replay_mem_size=200_000, tau=tau)
    comms = ThreadComm.group(world)
    cores, leaves, errs = [None] * world, [None] * world, []

    def run(r):
        try:
            torch.cuda.set_device(0)
            core = build_core(cfg, "cuda", seed=seed + r)
            rp = HbmReplay(cfg, "cuda", core.dev_state)
            ro = DeviceRollout(cfg, DevicePendulum(16, "cuda", seed=50 + r), rp, "cuda", segment=32, seed=60 + r,
                               use_graph=False)
            base = core.layout.actor.base
            for _ in range(8):
                ro.refresh_weights(core.online.flat.data_ptr(), base)
                ro.collect()
            dp = attach_replicated(core, rp, comms[r], Bl, chunk=4)
            for _ in range(updates // 32):
                ro.refresh_weights(core.online.flat.data_ptr(), base)      # 16 steps per update over the two ranks
                ro.collect()
                dp.mark_inserted()
                for _ in range(32):
                    dp.update()
            assert dp.check_consistency() == 0.0
            torch.cuda.synchronize()
            cores[r], leaves[r] = core, rp.leaf[: rp.size].clone()
        except Exception as e:    # pragma: no cover
            errs.append(e)
            comms[r].shared.barrier.abort()

    ts = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    [t.start() for t in ts]
    [t.join(600) for t in ts]
    assert not errs, errs
    ret = GreedyEvaluator(cfg, 64, "cuda", 99)(cores[0].target.flat.data_ptr(), cores[0].layout.actor.base)
    return cores, leaves, ret


def test_replicated_per_world_two_learns_the_pendulum():
    """Replicated data parallelism WITH prioritised replay over a world of two, as a learning run: each rank owns a
    shard of the replay fed by its own device explorers, draws its half of every global batch by priority, and writes
    back the priorities of its own rows only, one to two chunks after the draw.  Both ranks must hold bit-identical
    parameters at the end and the greedy target policy must swing the pendulum up (a random policy scores about -1200)."""
    cores, leaves, ret = _replicated_per_pendulum_run(tau=0.001, updates=20_000)
    assert torch.equal(cores[0].online.flat, cores[1].online.flat)
    assert torch.equal(cores[0].target.flat, cores[1].target.flat)
    for r in range(2):      # priorities were written back on both shards: a spread, not the insert value everywhere
        assert leaves[r].unique().numel() > 1000
    assert ret > -400.0, ret
