"""Experiment: what do cross-stream events around a graph replay cost per update?"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bench import synthetic_rows
from d4pg_pytorch_amd.config import default_config
from d4pg_pytorch_amd.models.core import build_core
from d4pg_pytorch_amd.replay.hbm_replay import HbmReplay
from d4pg_pytorch_amd.ops import hip as H

cfg = default_config(env="Walker2d-v2", model="d4pg", batch_size=256, device="cuda", v_min=0.0, v_max=5000.0,
                     compute_dtype="bf16", replay_mem_size=200_000)
core = build_core(cfg, "cuda")
rp = HbmReplay(cfg, "cuda", core.dev_state)
core.attach_replay(rp)
rp.insert(synthetic_rows(100_000, rp.S, rp.A, rp.row_ld, 0.95, "cuda", 1))
buf = torch.zeros(256, rp.packed_width(), device="cuda")
rp.enqueue_sample_rows(buf, 256, 0)
core.register_packed("g0", buf, rp.row_ld, None)
side = torch.cuda.Stream()
main = torch.cuda.current_stream()
sargs = rp.sample_rows_args(buf.clone(), 256)
ev_a, ev_b = torch.cuda.Event(), torch.cuda.Event()

def timeit(name, fn, n=3000):
    for _ in range(300): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    print(f"{name:60s} {(time.perf_counter()-t0)/n*1e6:8.2f} us/step", flush=True)

timeit("plain replay-source graph", core.update_from_replay)
timeit("packed-source graph (unpack inside)", lambda: core.update_from_source("g0"))
def f_rec():
    core.update_from_source("g0"); ev_a.record(main)
timeit("+ event record on main", f_rec)
def f_side_kernel():
    H.launch("d4pg_replay_sample_rows", sargs, side)
    core.update_from_source("g0")
timeit("+ independent kernel on side stream (no sync)", f_side_kernel)
def f_wait_done():
    side.wait_event(ev_a)
    H.launch("d4pg_replay_sample_rows", sargs, side)
    ev_b.record(side)
    main.wait_event(ev_b)
    core.update_from_source("g0")
    ev_a.record(main)
timeit("+ side waits main, main waits side (serialised)", f_wait_done)
ev = [torch.cuda.Event(), torch.cuda.Event()]
rd = [torch.cuda.Event(), torch.cuda.Event()]
state = {"k": 0, "used": [False, False]}
def f_pipelined():
    k = state["k"]
    if state["used"][1 - k]:
        side.wait_event(ev[1 - k])
    H.launch("d4pg_replay_sample_rows", sargs, side)
    rd[1 - k].record(side)
    if state["used"][k] or True:
        main.wait_event(rd[k]) if state["used"][k] else None
    core.update_from_source("g0")
    ev[k].record(main)
    state["used"][k] = True
    state["k"] = 1 - k
timeit("pipelined double-buffer pattern (as ReplicatedDP)", f_pipelined)
def f_host_only():
    k = state["k"]
    core.update_from_source("g0")
    state["k"] = 1 - k
timeit("packed-source graph again", f_host_only)
