"""Shared-memory ring, seq-locked weight slab, atomic control block (C++ host runtime)."""
import multiprocessing as mp
import time

import numpy as np
import pytest

from d4pg_pytorch_amd.parallel.ipc import ControlBlock, ShmRing, SlabReader, WeightSlab


def test_ring_fifo_wraparound_and_drop_accounting():
    with ShmRing.create(8, 5) as ring:
        assert ring.capacity == 8 and ring.row_floats == 5 and len(ring) == 0
        a = np.arange(30, dtype=np.float32).reshape(6, 5)
        assert ring.push(a, timeout_ms=0) == 6 and len(ring) == 6
        out = ring.pop(4)
        np.testing.assert_array_equal(out, a[:4])
        b = np.arange(100, 130, dtype=np.float32).reshape(6, 5)
        assert ring.push(b, timeout_ms=0) == 6          # wraps around the end
        assert len(ring) == 8
        assert ring.push(np.zeros((3, 5), np.float32), timeout_ms=0) == 0
        assert ring.dropped == 3                        # refused rows are counted, not silently lost
        got = ring.pop(100)
        np.testing.assert_array_equal(got, np.concatenate([a[4:], b]))
        assert len(ring) == 0 and ring.pushed_total == 12
        # zero-copy peek / advance
        ring.push(a, timeout_ms=0)
        off, n = ring.peek(100)
        view = ring.data_view()
        assert n >= 1
        np.testing.assert_array_equal(view[off:off + n], a[:n])
        ring.advance(n)
        assert len(ring) == 6 - n


def _producer(name, n_rows, width):
    ring = ShmRing.attach(name)
    for start in range(0, n_rows, 7):
        k = min(7, n_rows - start)
        rows = np.tile(np.arange(start, start + k, dtype=np.float32)[:, None], (1, width))
        assert ring.push(rows, timeout_ms=20_000) == k   # lossless back-pressure
    ring.mark_closed()
    ring.close()


def test_ring_cross_process_lossless_backpressure():
    ctx = mp.get_context("spawn")
    with ShmRing.create(64, 6) as ring:
        p = ctx.Process(target=_producer, args=(ring.name, 5000, 6))
        p.start()
        got = []
        deadline = time.time() + 60
        while (len(got) < 5000) and time.time() < deadline:
            rows = ring.pop(33)
            if rows.shape[0]:
                got.extend(rows[:, 0].tolist())
                assert np.all(rows == rows[:, :1])       # no torn rows
            else:
                time.sleep(0.0005)
        p.join(30)
        assert p.exitcode == 0
        assert got == list(range(5000))
        assert ring.dropped == 0 and ring.closed


def test_ring_check_mode_counts_slots_that_do_not_hold_the_expected_row():
    """Race detection: in check mode every slot carries the running index of its row.  A clean
    producer / consumer pair (copying pop and zero-copy peek + advance, across the wrap-around) never
    trips it; a consumer that is made to skip ahead does."""
    ctx = mp.get_context("spawn")
    with ShmRing.create(64, 6) as ring:
        ring.set_check(True)
        p = ctx.Process(target=_producer, args=(ring.name, 3000, 6))
        p.start()
        seen = 0
        deadline = time.time() + 60
        while seen < 3000 and time.time() < deadline:
            if seen % 2:
                seen += ring.pop(17).shape[0]
            else:
                _, n = ring.peek(9)
                ring.advance(n)
                seen += n
        p.join(30)
        assert p.exitcode == 0 and seen == 3000
        assert ring.seq_errors == 0 and ring.dropped == 0
        # a consumer releasing rows that were never produced reads slots with other rows' indices
        ring.advance(5)
        assert ring.seq_errors == 5
    with ShmRing.create(8, 2) as plain:             # off by default: nothing is stamped or verified
        plain.push(np.zeros((3, 2), np.float32))
        plain.advance(3)
        plain.advance(2)
        assert plain.seq_errors == 0


def test_attach_unknown_segment_fails_loudly():
    with pytest.raises(OSError):
        ShmRing.attach("/d4pg_no_such_ring")
    with pytest.raises(OSError):
        WeightSlab.attach("/d4pg_no_such_slab")


def _slab_writer(name, versions):
    slab = WeightSlab.attach(name)
    n = slab.n_floats
    for v in range(1, versions + 1):
        slab.publish(np.full(n, float(v), np.float32), v)
    slab.close()


def test_slab_broadcast_every_reader_sees_consistent_snapshots():
    ctx = mp.get_context("spawn")
    with WeightSlab.create(50_000) as slab:
        r1, r2 = SlabReader(slab), SlabReader(slab)
        assert not r1.poll()                              # nothing published yet
        slab.publish(np.full(50_000, 7.0, np.float32), 3)
        assert r1.poll() and r2.poll()                    # broadcast: BOTH readers get it
        assert np.all(r1.buf == 7.0) and np.all(r2.buf == 7.0) and slab.update_step == 3
        assert not r1.poll()                              # unchanged -> no copy
        p = ctx.Process(target=_slab_writer, args=(slab.name, 300))
        p.start()
        seen = []
        deadline = time.time() + 60
        while time.time() < deadline and (not seen or seen[-1] < 300.0):
            if r1.poll():
                assert np.all(r1.buf == r1.buf[0]), "torn read"   # seq-lock: never a mix of two versions
                seen.append(float(r1.buf[0]))
        p.join(30)
        assert p.exitcode == 0 and seen[-1] == 300.0 and seen == sorted(seen)


def _bump(name, n):
    c = ControlBlock.attach(name)
    for _ in range(n):
        c.add(ControlBlock.GLOBAL_EPISODE, 1)
    c.close()


def test_control_block_atomic_counters_across_processes():
    ctx = mp.get_context("spawn")
    with ControlBlock.create(n_workers=8) as ctrl:
        assert not ctrl.stop and ctrl.update_step == 0 and ctrl.n_workers == 8
        ps = [ctx.Process(target=_bump, args=(ctrl.name, 20_000)) for _ in range(3)]
        for p in ps:
            p.start()
        for p in ps:
            p.join(60)
        assert ctrl.global_episode == 60_000              # the reference's `value += 1` loses updates here
        ctrl.store(ControlBlock.UPDATE_STEP, 41)
        assert ctrl.add(ControlBlock.UPDATE_STEP) == 42 and ctrl.max(ControlBlock.UPDATE_STEP, 7) == 42
        ctrl.heartbeat(3, 123456)
        assert ctrl.last_heartbeat(3) == 123456
        ctrl.request_stop()
        assert ctrl.stop
