#!/usr/bin/env python
"""Per-kernel latency micro-benchmarks (graph-captured chains of ONE kernel type).

Each chain is N dependent launches of the same kernel captured into a HIP graph
and replayed; the reported number is wall time / N, i.e. the marginal cost of
one more launch of that kernel inside the learner's update graph (launch
boundary + kernel body).  Run on the GPU box:

    python bench/micro.py [--n 100] [--reps 30]
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from d4pg_pytorch_amd.ops import hip as H  # noqa: E402


def timed_graph(enqueue, n, reps):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        enqueue()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            enqueue()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (reps * n) * 1e6


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=100)
    ap.add_argument("--reps", type=int, default=30)
    args = ap.parse_args()
    H.lib()
    dev = "cuda"
    res = {}
    st = torch.zeros(C.sizeof(H.DevStateMirror), dtype=torch.uint8, device=dev)

    def tick():
        H._check(H.lib().d4pg_tick(st.data_ptr(), 0.9, 0.999, 0.4, 1.0, 1000, H.stream_ptr()), "tick")
    res["tick_1block"] = timed_graph(tick, args.n, args.reps)

    for B in (256, 4096):
        x = torch.randn(B, 400, device=dev)
        y = torch.zeros(B, 400, device=dev)
        w = torch.randn(400, 400, device=dev) * 0.05
        b = torch.zeros(400, device=dev)
        flip = [0]

        def fwd400():
            a, c = (x, y) if flip[0] == 0 else (y, x)
            flip[0] ^= 1
            H.gemm_group([H.gemm_desc(H.GEMM_NT, a, 400, w, 400, c, 400, B, 400, 400, bias=b, epi=H.EPI_RELU)])
        res[f"gemm_nt_{B}x400x400"] = timed_graph(fwd400, args.n, args.reps)

        xs = torch.randn(B, 12, device=dev)

        def fwd10():
            H.gemm_group([H.gemm_desc(H.GEMM_NT, xs, 12, w, 400, y, 400, B, 400, 10, bias=b, epi=H.EPI_RELU)])
        res[f"gemm_nt_{B}x400x10"] = timed_graph(fwd10, args.n, args.reps)

        dy = torch.randn(B, 400, device=dev)
        gw = torch.zeros(400, 400, device=dev)
        gb = torch.zeros(400, device=dev)
        dx = torch.zeros(B, 400, device=dev)

        def bwd():
            H.gemm_group([H.gemm_desc(H.GEMM_TN, dy, 400, x, 400, gw, 400, 400, 400, B, colsum=gb),
                          H.gemm_desc(H.GEMM_NN, dy, 400, w, 400, dx, 400, B, 400, 400,
                                      epi=H.EPI_MUL_RELU_MASK, aux=x, ldaux=400)])
        res[f"gemm_bwd_dw_dx_{B}x400x400"] = timed_graph(bwd, args.n, args.reps)

        def three():
            H.gemm_group([H.gemm_desc(H.GEMM_NT, x, 400, w, 400, y, 400, B, 400, 400, bias=b, epi=H.EPI_RELU),
                          H.gemm_desc(H.GEMM_NT, x, 400, w, 400, dx, 400, B, 400, 400, bias=b, epi=H.EPI_RELU),
                          H.gemm_desc(H.GEMM_NT, x, 400, w, 400, gw, 400, min(B, 400), 400, 400, bias=b, epi=H.EPI_RELU)])
        res[f"gemm_nt_group3_{B}x400x400"] = timed_graph(three, args.n, args.reps)

    n = 192_452 // 4 * 4
    p, g_, m, v, t = (torch.randn(n, device=dev) for _ in range(5))
    v.abs_()
    tick()

    def adam():
        a = H.AdamPolyakArgs()
        a.p, a.g, a.m, a.v, a.t, a.st, a.metrics, a.n = (p.data_ptr(), g_.data_ptr(), m.data_ptr(), v.data_ptr(),
                                                         t.data_ptr(), st.data_ptr(), None, n)
        a.lr, a.beta1, a.beta2, a.eps, a.tau, a.gscale, a.finish_step = 5e-4, 0.9, 0.999, 1e-8, 1e-3, 1.0, 0
        a.per_loss = a.weights = a.q_val = None
        H.launch("d4pg_adam_polyak", a)
    res["adam_polyak_192k"] = timed_graph(adam, args.n, args.reps)

    B, N, ld = 256, 51, 52
    lt, lo = torch.randn(B, ld, device=dev), torch.randn(B, ld, device=dev)
    r, d = torch.randn(B, device=dev), torch.zeros(B, device=dev)
    gm, wt = torch.full((B,), 0.95, device=dev), torch.ones(B, device=dev)
    dl, per, pr, met = torch.zeros(B, ld, device=dev), torch.zeros(B, device=dev), torch.zeros(B, device=dev), torch.zeros(4, device=dev)

    def closs():
        a = H.D4pgLossArgs()
        a.logits_t, a.logits, a.reward, a.done, a.gamma = lt.data_ptr(), lo.data_ptr(), r.data_ptr(), d.data_ptr(), gm.data_ptr()
        a.weights, a.dlogits, a.per_loss, a.priority = wt.data_ptr(), dl.data_ptr(), per.data_ptr(), pr.data_ptr()
        a.proj_out, a.metrics = None, met.data_ptr()
        a.B, a.N, a.ld, a.use_ce = B, N, ld, 0
        a.v_min, a.v_max, a.fixed_gamma, a.inv_B = -10.0, 10.0, -1.0, 1.0 / B
        H.launch("d4pg_critic_loss", a)
    res["d4pg_critic_loss_256x51"] = timed_graph(closs, args.n, args.reps)

    def aloss():
        a = H.D4pgActorLossArgs()
        a.logits, a.dlogits, a.q_out, a.metrics = lo.data_ptr(), dl.data_ptr(), per.data_ptr(), met.data_ptr()
        a.B, a.N, a.ld, a.v_min, a.v_max, a.inv_B = B, N, ld, -10.0, 10.0, 1.0 / B
        H.launch("d4pg_actor_loss", a)
    res["d4pg_actor_loss_256x51"] = timed_graph(aloss, args.n, args.reps)

    # torch (rocBLAS/hipBLASLt) reference for the same layer, eager and graph
    x = torch.randn(256, 400, device=dev)
    w = torch.randn(400, 400, device=dev) * 0.05
    bb = torch.zeros(400, device=dev)
    out = torch.zeros(256, 400, device=dev)

    def torch_linear():
        torch.relu(torch.nn.functional.linear(x, w, bb), out=out) if False else out.copy_(torch.relu(torch.nn.functional.linear(x, w, bb)))
    res["torch_linear_relu_256x400x400_graph(3 kernels)"] = timed_graph(torch_linear, args.n, args.reps)
    for k in res:
        res[k] = round(res[k], 3)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
