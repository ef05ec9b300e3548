"""HBM-resident replay + fan-out-64 priority tree (GPU only)."""
import numpy as np
import pytest
import torch

from d4pg_pytorch_amd.config import default_config

pytestmark = pytest.mark.gpu


def _mk(capacity, per, B=256, env="BipedalWalker-v2", **kw):
    from d4pg_pytorch_amd.models.hip_core import HipCore
    from d4pg_pytorch_amd.replay.hbm_replay import HbmReplay
    cfg = default_config(env=env, model="d4pg", batch_size=B, dense_size=64, device="cuda",
                         replay_mem_size=capacity, replay_memory_prioritized=int(per), use_hip_graph=False,
                         num_steps_train=1000, **kw)
    core = HipCore(cfg, "cuda")
    rp = HbmReplay(cfg, "cuda", core.dev_state)
    core.attach_replay(rp)
    return cfg, core, rp


def _rows(rp, n, start=0):
    """Row i carries its own global index in every field so gathers can be verified."""
    r = torch.zeros(n, rp.row_ld)
    ids = torch.arange(start, start + n, dtype=torch.float32)
    S, A = rp.S, rp.A
    r[:, :S] = ids[:, None]
    r[:, S:S + A] = -ids[:, None]
    r[:, S + A] = ids * 0.5
    r[:, S + A + 1] = (ids % 2)
    r[:, S + A + 2] = 0.9
    r[:, S + A + 3: S + A + 3 + S] = ids[:, None] + 0.25
    return r.cuda()


def test_tree_level_sizes():
    from d4pg_pytorch_amd.replay.hbm_replay import tree_level_sizes
    assert tree_level_sizes(50) == [64]
    assert tree_level_sizes(64) == [64]
    assert tree_level_sizes(1000) == [1024, 64]
    assert tree_level_sizes(1_000_000) == [1_000_000, 15_680, 256, 64]
    sizes = tree_level_sizes(10_000_000)
    assert sizes[-1] == 64 and all(s % 64 == 0 for s in sizes)
    for lo, hi in zip(sizes[1:], sizes[:-1]):
        assert lo * 64 >= hi


def test_uniform_insert_wraparound_and_gather():
    cfg, core, rp = _mk(capacity=1000, per=False)
    rp.insert(_rows(rp, 700))
    assert len(rp) == 700 and rp.head == 700
    rp.insert(_rows(rp, 600, start=700))          # wraps: slots 700..999 then 0..299
    assert len(rp) == 1000 and rp.head == 300
    torch.cuda.synchronize()
    st = core.read_dev_state()
    assert st.replay_size == 1000 and st.replay_head == 300 and st.inserted_total == 1300
    ids = rp.rows[:, 0].cpu()
    assert torch.equal(ids[:300], torch.arange(1000, 1300, dtype=torch.float32))
    assert torch.equal(ids[300:700], torch.arange(300, 700, dtype=torch.float32))
    assert torch.equal(ids[700:], torch.arange(700, 1000, dtype=torch.float32))
    # sample + gather
    rp.enqueue_sample(core, do_tick=False)
    torch.cuda.synchronize()
    S, A = rp.S, rp.A
    idx = core.idx.cpu()
    gid = rp.rows[idx.cuda(), 0].cpu()
    assert idx.min() >= 0 and idx.max() < 1000
    torch.testing.assert_close(core.sa[:, :S].cpu(), gid[:, None].expand(-1, S))
    torch.testing.assert_close(core.sa[:, S:S + A].cpu(), -gid[:, None].expand(-1, A))
    torch.testing.assert_close(core.s_pi[:, :S].cpu(), gid[:, None].expand(-1, S))
    torch.testing.assert_close(core.s2a2[:, :S].cpu(), gid[:, None].expand(-1, S) + 0.25)
    torch.testing.assert_close(core.reward.cpu(), gid * 0.5)
    torch.testing.assert_close(core.done.cpu(), gid % 2)
    torch.testing.assert_close(core.gamma.cpu(), torch.full((256,), 0.9))
    torch.testing.assert_close(core.weights.cpu(), torch.ones(256))


def test_uniform_sampling_is_uniform_and_changes_per_step():
    cfg, core, rp = _mk(capacity=512, per=False, B=256)
    rp.insert(_rows(rp, 100))
    counts = torch.zeros(100)
    seen = []
    for step in range(200):
        core._write_dev_state(step=step, keep_replay=True)
        rp.enqueue_sample(core, do_tick=False)
        idx = core.idx.cpu()
        assert idx.max() < 100
        counts += torch.bincount(idx, minlength=100).float()
        seen.append(idx[:8].tolist())
    assert len({tuple(s) for s in seen}) > 190, "sampling does not vary with the step counter"
    expected = counts.sum() / 100
    chi2 = ((counts - expected) ** 2 / expected).sum().item()
    assert chi2 < 180, chi2          # 99 dof: mean 99, 99.99th percentile ~ 160


def test_per_tree_matches_brute_force_and_is_weights():
    cfg, core, rp = _mk(capacity=5000, per=True, B=256)
    n = 3000
    rp.insert(_rows(rp, n))
    torch.cuda.synchronize()
    # fresh rows enter with max priority 1 -> leaf = 1^alpha
    assert abs(rp.total_priority() - n) < 1e-6
    assert abs(rp.min_priority() - 1.0) < 1e-6
    g = torch.Generator().manual_seed(0)
    idx = torch.randperm(n, generator=g)[:1500]
    pr = torch.rand(1500, generator=g) * 5 + 0.01
    rp.update_priorities(idx, pr)
    torch.cuda.synchronize()
    leaf = torch.ones(n, dtype=torch.float64)
    leaf[idx] = pr.double() ** cfg["priority_alpha"]
    torch.testing.assert_close(rp.leaf[:n].cpu().double(), leaf, rtol=1e-6, atol=1e-7)
    assert abs(rp.total_priority() - leaf.sum().item()) < 1e-3
    assert abs(rp.min_priority() - leaf.min().item()) < 1e-6
    assert abs(core.read_dev_state().max_priority - max(1.0, pr.max().item())) < 1e-6
    # internal levels equal a rebuild from the leaves
    sums = [None if s is None else s.clone() for s in rp.sums]
    rp.rebuild_tree()
    for l in range(1, rp.L + 1):
        torch.testing.assert_close(sums[l], rp.sums[l], rtol=1e-12, atol=1e-9)
    # proportional sampling frequency + importance weights
    counts = torch.zeros(n, dtype=torch.float64)
    T = 400
    for step in range(T):
        core._write_dev_state(step=step, keep_replay=True)
        rp.enqueue_sample(core, do_tick=False)
        ii = core.idx.cpu()
        counts += torch.bincount(ii, minlength=n).double()
        if step == 0:
            beta = cfg["priority_beta_start"]
            want_w = (leaf.min() / leaf[ii]) ** beta
            torch.testing.assert_close(core.weights.cpu().double(), want_w, rtol=1e-4, atol=1e-6)
    p = leaf / leaf.sum()
    expected = p * T * 256
    # pool into 30 bins of increasing priority for a stable chi-square
    order = torch.argsort(leaf)
    bins = torch.chunk(order, 30)
    chi2 = sum(((counts[b].sum() - expected[b].sum()) ** 2 / expected[b].sum()).item() for b in bins)
    assert chi2 < 80, chi2           # stratified sampling has lower variance than multinomial


@pytest.mark.parametrize("capacity,n_items", [(300_000, 4096), (300_000, 8192), (70_000, 2048), (1_000_000, 4096)])
def test_tree_update_shared_by_several_workgroups(capacity, n_items):
    """Large priority write-backs are dealt to up to 8 workgroups by subtree ownership; the tree they leave
    equals a rebuild from the leaves, duplicates and repeated updates included."""
    cfg, core, rp = _mk(capacity=capacity, per=True, B=256)
    assert rp.L >= 2 and rp.tree_parts(n_items) == min(8, n_items // 512) > 1
    n = min(capacity, 250_000)
    for off in range(0, n, 50_000):
        rp.insert(_rows(rp, min(50_000, n - off)))
    g = torch.Generator().manual_seed(1)
    leaf = torch.ones(n, dtype=torch.float64)
    for rnd in range(4):
        idx = torch.randint(0, n, (n_items,), generator=g)
        if rnd == 1:
            idx[: n_items // 2] = idx[n_items // 2:]                 # every item twice, with the same priority below
        if rnd == 2:
            idx = torch.randint(0, 5000, (n_items,), generator=g)    # crowded: few subtrees, many duplicates
        pr = torch.rand(n_items, generator=g) * 3 + 0.05
        uniq, inv = torch.unique(idx, return_inverse=True)
        pr = pr[: uniq.numel()][inv]                                 # duplicates carry equal priorities
        rp.update_priorities(idx, pr)
        leaf[idx] = pr.double() ** cfg["priority_alpha"]
        torch.cuda.synchronize()
        torch.testing.assert_close(rp.leaf[:n].cpu().double(), leaf, rtol=1e-6, atol=1e-7)
        got_s = [None if x is None else x.clone() for x in rp.sums]
        got_m = [None if x is None else x.clone() for x in rp.mins]
        rp.rebuild_tree()
        for l in range(1, rp.L + 1):
            torch.testing.assert_close(got_s[l], rp.sums[l], rtol=1e-12, atol=1e-9)
            assert torch.equal(got_m[l], rp.mins[l])
        assert abs(rp.total_priority() - leaf.sum().item()) < 1e-2 and int(rp._ticket.item()) == 0
    assert abs(core.read_dev_state().max_priority - max(1.0, 3.05)) < 0.1


def test_per_beta_anneals_with_step():
    cfg, core, rp = _mk(capacity=1000, per=True, B=64)
    rp.insert(_rows(rp, 500))
    rp.update_priorities(torch.arange(500), torch.linspace(0.1, 4.0, 500))
    core._write_dev_state(step=1000, keep_replay=True)   # == num_steps_train -> beta_end
    rp.enqueue_sample(core, do_tick=False)
    leaf = rp.leaf[:500].cpu().double()
    ii = core.idx.cpu()
    want = (leaf.min() / leaf[ii]) ** cfg["priority_beta_end"]
    torch.testing.assert_close(core.weights.cpu().double(), want, rtol=1e-4, atol=1e-6)


def test_per_overwrite_resets_priority_and_small_capacity():
    cfg, core, rp = _mk(capacity=60, per=True, B=32)      # single-node tree (L == 0)
    assert rp.L == 0
    rp.insert(_rows(rp, 60))
    rp.update_priorities(torch.arange(60), torch.full((60,), 0.01))
    rp.insert(_rows(rp, 10, start=60))                    # overwrites slots 0..9 with max priority
    torch.cuda.synchronize()
    leaf = rp.leaf[:60].cpu()
    assert torch.all(leaf[:10] == 1.0) and torch.all(leaf[10:] < 0.1)
    counts = torch.zeros(60)
    for step in range(50):
        core._write_dev_state(step=step, keep_replay=True)
        rp.enqueue_sample(core, do_tick=False)
        counts += torch.bincount(core.idx.cpu(), minlength=60).float()
    assert counts[:10].sum() > 0.5 * counts.sum()


@pytest.mark.parametrize("per", [False, True])
@pytest.mark.parametrize("graph", [False, True])
def test_update_from_replay_end_to_end(per, graph):
    from d4pg_pytorch_amd.models.hip_core import HipCore
    from d4pg_pytorch_amd.replay.hbm_replay import HbmReplay
    cfg = default_config(env="BipedalWalker-v2", model="d4pg", batch_size=256, device="cuda",
                         replay_mem_size=100_000, replay_memory_prioritized=int(per), use_hip_graph=graph,
                         num_steps_train=1000)
    core = HipCore(cfg, "cuda")
    rp = HbmReplay(cfg, "cuda", core.dev_state)
    core.attach_replay(rp)
    g = torch.Generator().manual_seed(1)
    rows = torch.randn(20_000, rp.row_ld, generator=g)
    S, A = rp.S, rp.A
    rows[:, S + A + 1] = (torch.rand(20_000, generator=g) < 0.05).float()
    rows[:, S + A + 2] = 0.99 ** 5
    rp.insert(rows.cuda())
    p0 = core.online.flat.clone()
    for _ in range(20):
        prio = core.update_from_replay()
    torch.cuda.synchronize()
    assert core.read_dev_state().step_done == 20
    assert torch.isfinite(core.online.flat).all() and not torch.equal(p0, core.online.flat)
    m = core.read_metrics()
    assert m["count"] == 20 and np.isfinite(m["value_loss"]) and np.isfinite(m["policy_loss"])
    assert torch.all(prio > 0)
    if per:
        # sampled rows carry their new priority, everything else still 1
        leaf = rp.leaf[:20_000].cpu()
        assert (leaf != 1.0).sum() > 256
        sums = [None if s is None else s.clone() for s in rp.sums]
        rp.rebuild_tree()
        for l in range(1, rp.L + 1):
            torch.testing.assert_close(sums[l], rp.sums[l], rtol=1e-12, atol=1e-9)


@pytest.mark.parametrize("zero_copy", [True, False])
def test_ring_ingestor_moves_rows_ring_to_hbm(zero_copy):
    """Explorer rings -> HBM replay: with ``zero_copy`` the DMA reads the rings' own (registered) shared memory and
    slots return to the producer only after the copy's event; rows arrive complete and in order either way."""
    import numpy as np
    from d4pg_pytorch_amd.models.core import build_core
    from d4pg_pytorch_amd.models.learner import RingIngestor
    from d4pg_pytorch_amd.parallel.ipc import ShmRing
    from d4pg_pytorch_amd.replay.hbm_replay import HbmReplay
    cfg = default_config(env="BipedalWalker-v2", model="d4pg", batch_size=64, dense_size=64, device="cuda",
                         replay_mem_size=8192, ingest_chunk=64)
    core = build_core(cfg, "cuda")
    rp = HbmReplay(cfg, "cuda", core.dev_state)
    rings = [ShmRing.create(100, rp.row_ld) for _ in range(2)]       # small: the readable span wraps often
    for r in rings:
        r.set_check(True)
    ing = RingIngestor(rings, rp, torch.device("cuda"), 64, zero_copy=zero_copy)
    if zero_copy and not ing.zero_copy:
        pytest.skip("the HIP runtime on this host refuses hipHostRegister (bench/exp/host_register_probe.py): "
                    "the pinned-staging path is what runs here")
    assert ing.zero_copy == zero_copy
    rng = np.random.default_rng(0)
    sent = [[], []]
    got_total = 0
    for it in range(60):
        for k, r in enumerate(rings):
            n = int(rng.integers(1, 70))
            rows = rng.standard_normal((n, rp.row_ld)).astype(np.float32)
            rows[:, 0] = k                                           # which ring
            rows[:, 1] = np.arange(len(sent[k]), len(sent[k]) + n)   # running index inside the ring
            ok = r.push(rows, timeout_ms=0)
            sent[k].extend(rows[:ok])
        got_total += ing.poll()
        if it % 7 == 0:
            torch.cuda.synchronize()
    while sum(len(r) for r in rings):
        got_total += ing.poll()
        torch.cuda.synchronize()
    ing.close() if zero_copy else None
    torch.cuda.synchronize()
    assert got_total == len(sent[0]) + len(sent[1]) == rp.size and got_total > 1500
    assert all(len(r) == 0 and r.seq_errors == 0 for r in rings)
    rows = rp.rows[: rp.size].cpu().numpy()
    for k in range(2):
        mine = rows[rows[:, 0] == k]
        np.testing.assert_array_equal(mine[:, 1], np.arange(len(sent[k])))       # nothing lost, duplicated or reordered
        np.testing.assert_array_equal(mine, np.stack(sent[k]))
    for r in rings:
        r.close()
