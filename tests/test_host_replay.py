import numpy as np
import pytest

from d4pg_pytorch_amd.config import default_config
from d4pg_pytorch_amd.replay.hbm_replay import pack_rows, row_ld
from d4pg_pytorch_amd.replay.host_replay import HostReplay, SumMinTree


def _cfg(**kw):
    return default_config(env="BipedalWalker-v2", model="d4pg", device="cpu", **kw)


def _rows(n, S=24, A=4, start=0):
    ids = np.arange(start, start + n, dtype=np.float32)
    return pack_rows(np.tile(ids[:, None], (1, S)), np.tile(-ids[:, None], (1, A)), ids * 0.5,
                     np.tile(ids[:, None] + 0.25, (1, S)), ids % 2, np.full(n, 0.9))


def test_pack_rows_layout():
    r = _rows(3)
    assert r.shape == (3, row_ld(24, 4)) and r.dtype == np.float32
    assert r[2, 0] == 2 and r[2, 24] == -2 and r[2, 28] == 1.0 and r[2, 29] == 0 and r[2, 30] == np.float32(0.9)
    assert r[2, 31] == 2.25 and np.all(r[:, 55:] == 0)


def test_sum_min_tree_against_brute_force():
    rng = np.random.default_rng(0)
    t = SumMinTree(1000)
    vals = np.zeros(1000)
    for _ in range(20):
        idx = rng.integers(0, 1000, 64)
        v = rng.random(64) * 3 + 0.01
        _, last = np.unique(idx[::-1], return_index=True)
        keep = idx.size - 1 - last
        t.set(idx[keep], v[keep])
        vals[idx[keep]] = v[keep]
    assert t.total == pytest.approx(vals.sum(), rel=1e-12)
    assert t.minimum == pytest.approx(vals[vals > 0].min())
    mass = rng.random(500) * vals.sum()
    got = t.find_prefix(mass)
    want = np.searchsorted(np.cumsum(vals), mass, side="right")
    np.testing.assert_array_equal(got, want)


def test_uniform_ring_overwrites_oldest_and_samples_valid_rows():
    rp = HostReplay(_cfg(replay_mem_size=100))
    rp.insert_rows(_rows(70))
    rp.insert_rows(_rows(60, start=70))
    assert len(rp) == 100 and rp.head == 30
    ids = rp.rows[:, 0]
    np.testing.assert_array_equal(ids[:30], np.arange(100, 130))
    np.testing.assert_array_equal(ids[30:], np.arange(30, 100))
    s, a, r, s2, d, g, w, idx = rp.sample(256)
    assert s.shape == (256, 24) and a.shape == (256, 4) and idx.dtype == np.int64
    np.testing.assert_array_equal(a[:, 0], -s[:, 0])
    np.testing.assert_allclose(s2[:, 0], s[:, 0] + 0.25)
    np.testing.assert_allclose(r, s[:, 0] * 0.5)
    assert np.all(w == 1.0) and np.all(g == np.float32(0.9))
    rp.add(np.zeros(24), np.zeros(4), 1.0, np.ones(24), True, 0.5)   # reference-style single add
    assert rp.rows[30, 28] == 1.0 and rp.rows[30, 29] == 1.0


def test_prioritised_sampling_is_proportional_with_correct_weights():
    cfg = _cfg(replay_mem_size=2000, replay_memory_prioritized=1)
    rp = HostReplay(cfg, seed=1)
    n = 1500
    rp.insert_rows(_rows(n))
    assert rp.tree.total == pytest.approx(n)              # new rows enter at max priority 1
    rng = np.random.default_rng(2)
    idx = rng.permutation(n)[:800]
    pr = rng.random(800) * 4 + 0.05
    rp.update_priorities(idx, pr)
    leaf = np.ones(n)
    leaf[idx] = pr ** cfg["priority_alpha"]
    assert rp.tree.total == pytest.approx(leaf.sum(), rel=1e-9)
    assert rp.max_priority == pytest.approx(max(1.0, pr.max()))
    counts = np.zeros(n)
    for _ in range(300):
        *_, w, ii = rp.sample(256, beta=0.5)
        counts += np.bincount(ii, minlength=n)
        np.testing.assert_allclose(w, (leaf.min() / leaf[ii]) ** 0.5, rtol=1e-5)
    expected = leaf / leaf.sum() * 300 * 256
    order = np.argsort(leaf)
    chi2 = sum((counts[b].sum() - expected[b].sum()) ** 2 / expected[b].sum() for b in np.array_split(order, 30))
    assert chi2 < 80, chi2
    # overwritten rows re-enter at the running max priority
    rp.insert_rows(_rows(600, start=n))
    assert rp.tree.sum[rp.tree.cap + 50] == pytest.approx(rp.max_priority ** cfg["priority_alpha"])
    # NaN / non-positive priorities never poison the tree
    rp.update_priorities([0, 1], [float("nan"), -3.0])
    assert np.isfinite(rp.tree.total) and rp.tree.minimum > 0


def test_dump_writes_valid_rows(tmp_path):
    import pickle
    rp = HostReplay(_cfg(replay_mem_size=50))
    rp.insert_rows(_rows(20))
    p = tmp_path / "replay_buffer.pkl"
    rp.dump(str(p))
    blob = pickle.load(open(p, "rb"))
    assert blob["rows"].shape == (20, row_ld(24, 4)) and blob["state_dim"] == 24
