"""Numerics of every HIP kernel against the plain-PyTorch fp32 oracles (GPU only)."""
import math

import pytest
import torch

from d4pg_pytorch_amd.config import default_config
from d4pg_pytorch_amd.ops import ref

pytestmark = pytest.mark.gpu


def _H():
    from d4pg_pytorch_amd.ops import hip as H
    H.lib()
    return H


def pad4(n):
    return (n + 3) // 4 * 4


def _rand(*shape, scale=1.0):
    return (torch.rand(*shape, device="cuda") * 2 - 1) * scale


# --------------------------------------------------------------------------- GEMM
@pytest.mark.parametrize("M,N,K", [(256, 400, 400), (256, 400, 10), (256, 51, 400), (37, 70, 23),
                                   (256, 1, 400), (4096, 400, 119), (5, 3, 3)])
@pytest.mark.parametrize("epi", ["none", "relu", "tanh"])
def test_gemm_nt_forward(M, N, K, epi):
    H = _H()
    torch.manual_seed(M * 7 + N * 3 + K)
    lda, ldb, ldc = pad4(K) + 4, pad4(K), pad4(N)
    x = torch.zeros(M, lda, device="cuda")
    x[:, :K] = _rand(M, K)
    x[:, K:] = 7.0          # garbage in the pad columns must not leak into the product
    w = torch.zeros(N, ldb, device="cuda")
    w[:, :K] = _rand(N, K, scale=0.2)
    b = _rand(N)
    out = torch.full((M, ldc), -5.0, device="cuda")
    e = {"none": H.EPI_NONE, "relu": H.EPI_RELU, "tanh": H.EPI_TANH}[epi]
    H.gemm_group([H.gemm_desc(H.GEMM_NT, x, lda, w, ldb, out, ldc, M, N, K, bias=b, epi=e)])
    want = x[:, :K].double() @ w[:, :K].double().t() + b.double()
    want = {"none": want, "relu": want.relu(), "tanh": want.tanh()}[epi]
    torch.testing.assert_close(out[:, :N].double(), want, rtol=2e-5, atol=2e-5)
    assert torch.all(out[:, N:] == -5.0), "pad columns of C were written"


@pytest.mark.parametrize("M,N,K", [(256, 400, 51), (256, 400, 400), (256, 2, 400), (33, 65, 7), (4096, 400, 400)])
def test_gemm_nn_dx_with_relu_mask(M, N, K):
    H = _H()
    torch.manual_seed(1)
    lda, ldb, ldc = pad4(K), pad4(N) + 8, pad4(N)
    dy = torch.zeros(M, lda, device="cuda")
    dy[:, :K] = _rand(M, K)
    w = _rand(K, ldb, scale=0.3)
    hmask = _rand(M, ldc)
    out = torch.zeros(M, ldc, device="cuda")
    H.gemm_group([H.gemm_desc(H.GEMM_NN, dy, lda, w, ldb, out, ldc, M, N, K,
                              epi=H.EPI_MUL_RELU_MASK, aux=hmask, ldaux=ldc)])
    want = (dy[:, :K].double() @ w[:, :N].double()) * (hmask[:, :N] > 0)
    torch.testing.assert_close(out[:, :N].double(), want, rtol=2e-5, atol=2e-5)


def test_gemm_nn_tanh_grad_and_column_offset():
    H = _H()
    M, K, S, A = 256, 400, 17, 6
    ldw = pad4(S + A)
    dy = _rand(M, K)
    w = _rand(K, ldw, scale=0.3)
    pi = torch.tanh(_rand(M, ldw))
    out = torch.zeros(M, pad4(A), device="cuda")
    H.gemm_group([H.gemm_desc(H.GEMM_NN, dy, K, w.view(-1)[S:], ldw, out, pad4(A), M, A, K,
                              epi=H.EPI_MUL_TANH_GRAD, aux=pi.view(-1)[S:], ldaux=ldw)])
    want = (dy.double() @ w[:, S:S + A].double()) * (1 - pi[:, S:S + A].double() ** 2)
    torch.testing.assert_close(out[:, :A].double(), want, rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("M,N,K", [(400, 400, 256), (51, 400, 256), (400, 23, 256), (1, 400, 256),
                                   (6, 400, 300), (400, 400, 4096), (35, 9, 5)])
def test_gemm_tn_dw_and_bias_grad(M, N, K):
    H = _H()
    torch.manual_seed(2)
    lda, ldb, ldc = pad4(M) + 4, pad4(N), pad4(N)
    dy = _rand(K, lda)
    x = _rand(K, ldb)
    gw = torch.full((M, ldc), 9.0, device="cuda")
    gb = torch.zeros(pad4(M), device="cuda")
    H.gemm_group([H.gemm_desc(H.GEMM_TN, dy, lda, x, ldb, gw, ldc, M, N, K, colsum=gb)])
    want = dy[:, :M].double().t() @ x[:, :N].double()
    tol = 3e-5 * math.sqrt(K / 256)
    torch.testing.assert_close(gw[:, :N].double(), want, rtol=tol, atol=tol)
    torch.testing.assert_close(gb[:M].double(), dy[:, :M].double().sum(0), rtol=tol, atol=tol)


def test_gemm_group_of_mixed_problems():
    H = _H()
    B, Hd, S, A = 256, 400, 8, 2
    x1, x2 = _rand(B, pad4(S)), _rand(B, pad4(S + A))
    w1, w2 = _rand(Hd, pad4(S)), _rand(Hd, pad4(S + A))
    b1, b2 = _rand(Hd), _rand(Hd)
    o1, o2 = torch.zeros(B, Hd, device="cuda"), torch.zeros(B, Hd, device="cuda")
    dy, xx = _rand(B, Hd), _rand(B, Hd)
    gw = torch.zeros(Hd, Hd, device="cuda")
    H.gemm_group([
        H.gemm_desc(H.GEMM_NT, x1, pad4(S), w1, pad4(S), o1, Hd, B, Hd, S, bias=b1, epi=H.EPI_RELU),
        H.gemm_desc(H.GEMM_TN, dy, Hd, xx, Hd, gw, Hd, Hd, Hd, B),
        H.gemm_desc(H.GEMM_NT, x2, pad4(S + A), w2, pad4(S + A), o2, Hd, B, Hd, S + A, bias=b2, epi=H.EPI_RELU),
    ])
    torch.testing.assert_close(o1, torch.relu(x1[:, :S] @ w1[:, :S].t() + b1), rtol=2e-5, atol=2e-5)
    torch.testing.assert_close(o2, torch.relu(x2[:, :S + A] @ w2[:, :S + A].t() + b2), rtol=2e-5, atol=2e-5)
    torch.testing.assert_close(gw.double(), dy.double().t() @ xx.double(), rtol=3e-5, atol=3e-5)


# --------------------------------------------------------------------------- losses
def _critic_loss_inputs(B, N):
    torch.manual_seed(3)
    lt, lo = _rand(B, N, scale=3.0), _rand(B, N, scale=3.0)
    r = _rand(B, scale=5.0)
    d = (torch.rand(B, device="cuda") < 0.2).float()
    g = torch.full((B,), 0.99 ** 5, device="cuda")
    g[::3] = 0.99 ** 2
    w = torch.rand(B, device="cuda") + 0.1
    return lt, lo, r, d, g, w


@pytest.mark.parametrize("N,vmin,vmax", [(51, -10.0, 10.0), (51, 0.0, 1000.0), (101, -3.0, 40.0), (7, -1.0, 1.0)])
@pytest.mark.parametrize("kind", ["bce", "ce"])
def test_d4pg_critic_loss_kernel(N, vmin, vmax, kind):
    H = _H()
    B = 256
    lt, lo, r, d, g, w = _critic_loss_inputs(B, N)
    # exact-atom hits and terminal rows at the support edges
    r[0], d[0] = vmax + 5.0, 1.0
    r[1], d[1] = vmin, 1.0
    r[2], d[2] = vmin + (vmax - vmin) / (N - 1) * 3, 1.0
    ld = pad4(N)
    ltp, lop = torch.zeros(B, ld, device="cuda"), torch.zeros(B, ld, device="cuda")
    ltp[:, :N], lop[:, :N] = lt, lo
    dl, proj = torch.zeros(B, ld, device="cuda"), torch.zeros(B, ld, device="cuda")
    per, prio = torch.zeros(B, device="cuda"), torch.zeros(B, device="cuda")
    met = torch.zeros(4, device="cuda")
    a = H.D4pgLossArgs()
    a.logits_t, a.logits, a.reward, a.done, a.gamma = ltp.data_ptr(), lop.data_ptr(), r.data_ptr(), d.data_ptr(), g.data_ptr()
    a.weights, a.dlogits, a.per_loss, a.priority = w.data_ptr(), dl.data_ptr(), per.data_ptr(), prio.data_ptr()
    a.proj_out, a.metrics = proj.data_ptr(), met.data_ptr()
    a.B, a.N, a.ld, a.use_ce = B, N, ld, int(kind == "ce")
    a.v_min, a.v_max, a.fixed_gamma, a.inv_B = vmin, vmax, -1.0, 1.0 / B
    H.launch("d4pg_critic_loss", a)
    torch.cuda.synchronize()

    m_ref = ref.l2_project(torch.softmax(lt, 1), r, d, g, vmin, vmax)
    torch.testing.assert_close(proj[:, :N], m_ref, rtol=1e-4, atol=2e-6)
    m_loop = ref.l2_project_loop(torch.softmax(lt, 1), r, d, g, vmin, vmax).cuda()
    torch.testing.assert_close(proj[:, :N], m_loop, rtol=1e-4, atol=5e-6)
    torch.testing.assert_close(proj[:, :N].sum(1), torch.ones(B, device="cuda"), rtol=1e-5, atol=1e-5)

    lo_req = lo.clone().requires_grad_(True)
    loss, per_ref, prio_ref = ref.d4pg_critic_loss(lo_req, m_ref, w, kind)
    loss.backward()
    torch.testing.assert_close(per, per_ref.detach(), rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(prio, prio_ref, rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(dl[:, :N], lo_req.grad, rtol=2e-4, atol=1e-8)
    torch.testing.assert_close(dl[:, :N], ref.d4pg_dlogits(lo, m_ref, w, kind), rtol=2e-4, atol=1e-8)


def test_d4pg_critic_loss_fixed_gamma_matches_reference_scalar():
    H = _H()
    B, N = 64, 51
    lt, lo, r, d, g, w = _critic_loss_inputs(B, N)
    ld = pad4(N)
    ltp, lop = torch.zeros(B, ld, device="cuda"), torch.zeros(B, ld, device="cuda")
    ltp[:, :N], lop[:, :N] = lt, lo
    dl, proj = torch.zeros(B, ld, device="cuda"), torch.zeros(B, ld, device="cuda")
    per, prio, met = torch.zeros(B, device="cuda"), torch.zeros(B, device="cuda"), torch.zeros(4, device="cuda")
    a = H.D4pgLossArgs()
    a.logits_t, a.logits, a.reward, a.done, a.gamma = ltp.data_ptr(), lop.data_ptr(), r.data_ptr(), d.data_ptr(), g.data_ptr()
    a.weights, a.dlogits, a.per_loss, a.priority = None, dl.data_ptr(), per.data_ptr(), prio.data_ptr()
    a.proj_out, a.metrics = proj.data_ptr(), met.data_ptr()
    a.B, a.N, a.ld, a.use_ce = B, N, ld, 0
    a.v_min, a.v_max, a.fixed_gamma, a.inv_B = -10.0, 10.0, 0.99 ** 5, 1.0 / B
    H.launch("d4pg_critic_loss", a)
    m_ref = ref.l2_project(torch.softmax(lt, 1), r, d, 0.99 ** 5, -10.0, 10.0)
    torch.testing.assert_close(proj[:, :N], m_ref, rtol=1e-4, atol=2e-6)


def test_d4pg_actor_loss_kernel():
    H = _H()
    B, N, vmin, vmax = 256, 51, -100.0, 300.0
    lg = _rand(B, N, scale=2.0)
    ld = pad4(N)
    lp, dl = torch.zeros(B, ld, device="cuda"), torch.zeros(B, ld, device="cuda")
    lp[:, :N] = lg
    q, met = torch.zeros(B, device="cuda"), torch.zeros(4, device="cuda")
    a = H.D4pgActorLossArgs()
    a.logits, a.dlogits, a.q_out, a.metrics = lp.data_ptr(), dl.data_ptr(), q.data_ptr(), met.data_ptr()
    a.B, a.N, a.ld, a.v_min, a.v_max, a.inv_B = B, N, ld, vmin, vmax, 1.0 / B
    H.launch("d4pg_actor_loss", a)
    lg_req = lg.clone().requires_grad_(True)
    qr = ref.expected_q(lg_req, vmin, vmax)
    loss = -qr.mean()
    loss.backward()
    torch.testing.assert_close(q, qr.detach(), rtol=1e-5, atol=1e-3)
    torch.testing.assert_close(dl[:, :N], lg_req.grad, rtol=1e-4, atol=1e-6)


def test_d3pg_loss_kernels():
    H = _H()
    B, ldq = 256, 4
    qt, q = _rand(B, ldq, scale=5.0), _rand(B, ldq, scale=5.0)
    r, d = _rand(B), (torch.rand(B, device="cuda") < 0.3).float()
    g = torch.full((B,), 0.95, device="cuda")
    w = torch.rand(B, device="cuda") + 0.5
    dq, per, prio, met = torch.zeros(B, ldq, device="cuda"), torch.zeros(B, device="cuda"), torch.zeros(B, device="cuda"), torch.zeros(4, device="cuda")
    a = H.D3pgLossArgs()
    a.q_t, a.q, a.reward, a.done, a.gamma, a.weights = qt.data_ptr(), q.data_ptr(), r.data_ptr(), d.data_ptr(), g.data_ptr(), w.data_ptr()
    a.dq, a.per_loss, a.priority, a.metrics = dq.data_ptr(), per.data_ptr(), prio.data_ptr(), met.data_ptr()
    a.B, a.ldq, a.fixed_gamma, a.min_value, a.max_value, a.inv_B = B, ldq, -1.0, -2.0, 2.0, 1.0 / B
    H.launch("d3pg_critic_loss", a)
    y = ref.d3pg_td_target(r, d, g, qt[:, :1], -2.0, 2.0)
    qq = q[:, :1].clone().requires_grad_(True)
    per_ref = (qq - y).pow(2).reshape(-1)
    loss = (per_ref * w).mean()
    loss.backward()
    torch.testing.assert_close(per, per_ref.detach(), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(dq[:, 0], qq.grad[:, 0], rtol=1e-5, atol=1e-7)
    b = H.D3pgActorLossArgs()
    dq2, qo = torch.zeros(B, ldq, device="cuda"), torch.zeros(B, device="cuda")
    b.q, b.dq, b.q_out, b.B, b.ldq, b.inv_B = q.data_ptr(), dq2.data_ptr(), qo.data_ptr(), B, ldq, 1.0 / B
    H.launch("d3pg_actor_loss", b)
    torch.testing.assert_close(dq2[:, 0], torch.full((B,), -1.0 / B, device="cuda"))
    torch.testing.assert_close(qo, q[:, 0])


# --------------------------------------------------------------------------- optimiser
def test_adam_polyak_kernel_matches_torch_adam():
    import ctypes as C
    H = _H()
    n = 4 * 5001
    torch.manual_seed(5)
    p0 = _rand(n)
    t0 = _rand(n)
    p, t = p0.clone(), t0.clone()
    m, v = torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    st = torch.zeros(C.sizeof(H.DevStateMirror), dtype=torch.uint8, device="cuda")
    pr = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([pr], lr=5e-4)
    tr = t0.clone()
    met = torch.zeros(4, device="cuda")
    pl, wts, qv = _rand(300), torch.rand(300, device="cuda"), _rand(300)
    for step in range(1, 6):
        g = _rand(n, scale=0.1)
        H._check(H.lib().d4pg_tick(st.data_ptr(), 0.9, 0.999, 0.4, 1.0, 100, H.stream_ptr()), "tick")
        a = H.AdamPolyakArgs()
        a.p, a.g, a.m, a.v, a.t, a.st, a.metrics, a.n = p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), t.data_ptr(), st.data_ptr(), None, n
        a.lr, a.beta1, a.beta2, a.eps, a.tau, a.gscale, a.finish_step = 5e-4, 0.9, 0.999, 1e-8, 0.001, 1.0, 1
        a.metrics, a.per_loss, a.weights, a.q_val = met.data_ptr(), pl.data_ptr(), wts.data_ptr(), qv.data_ptr()
        a.B, a.inv_B = 300, 1.0 / 300
        H.launch("d4pg_adam_polyak", a)
        pr.grad = g.clone()
        opt.step()
        ref.polyak_(tr, pr.data, 0.001)
    torch.testing.assert_close(p, pr.data, rtol=1e-5, atol=1e-7)
    torch.testing.assert_close(t, tr, rtol=1e-5, atol=1e-7)
    raw = H.DevStateMirror.from_buffer_copy(bytes(st.cpu().numpy().tobytes()))
    assert raw.step_done == 5 and raw.step_cur == 5
    want = torch.stack([5 * (pl * wts).mean(), 5 * (-qv).mean(), torch.tensor(5.0, device="cuda")])
    torch.testing.assert_close(met[:3], want, rtol=1e-4, atol=1e-5)


def test_nstep_kernel():
    H = _H()
    T, E, n, gamma = 40, 7, 5, 0.9
    torch.manual_seed(6)
    r = _rand(T, E)
    d = (torch.rand(T, E, device="cuda") < 0.1).float()
    tr = (torch.rand(T, E, device="cuda") < 0.05).float()
    R, gk, dd = torch.zeros(T, E, device="cuda"), torch.zeros(T, E, device="cuda"), torch.zeros(T, E, device="cuda")
    nxt = torch.zeros(T, E, dtype=torch.int32, device="cuda")
    a = H.NstepArgs()
    a.reward, a.done, a.trunc, a.R, a.gk, a.nxt, a.d = r.data_ptr(), d.data_ptr(), tr.data_ptr(), R.data_ptr(), gk.data_ptr(), nxt.data_ptr(), dd.data_ptr()
    a.T, a.E, a.n, a.gamma = T, E, n, gamma
    H.launch("d4pg_nstep", a)
    Rr, gr, nr, dr = ref.nstep_assemble(r.cpu(), d.cpu(), n, gamma, tr.cpu())
    torch.testing.assert_close(R.cpu(), Rr, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(gk.cpu(), gr, rtol=1e-5, atol=1e-6)
    assert torch.equal(nxt.cpu().long(), nr)
    assert torch.equal(dd.cpu(), dr)


# --------------------------------------------------------------------------- full update
def _cfg(model, **kw):
    base = dict(env="LunarLanderContinuous-v2", model=model, batch_size=256, dense_size=400,
                v_min=-20.0, v_max=20.0, device="cuda", num_steps_train=1000, random_seed=11)
    base.update(kw)
    return default_config(**base)


def _batch(cfg, seed, device):
    from d4pg_pytorch_amd.models.core import Batch
    g = torch.Generator().manual_seed(seed)
    B, S, A = cfg["batch_size"], cfg["state_dim"], cfg["action_dim"]
    return Batch(state=torch.randn(B, S, generator=g), action=torch.rand(B, A, generator=g) * 2 - 1,
                 reward=torch.randn(B, generator=g), next_state=torch.randn(B, S, generator=g),
                 done=(torch.rand(B, generator=g) < 0.1).float(),
                 gamma=torch.full((B,), cfg["discount_rate"] ** cfg["n_step_returns"]),
                 weights=torch.rand(B, generator=g) + 0.5,
                 indices=torch.arange(B)).to(device)


@pytest.mark.parametrize("model,kw", [("d4pg", {}), ("d4pg", {"critic_loss": "ce"}), ("d3pg", {}),
                                      ("d4pg", {"env": "Ant-v2", "batch_size": 128, "use_per_sample_gamma": False}),
                                      ("ddpg", {"env": "Pendulum-v0", "dense_size": 100})])
@pytest.mark.parametrize("graph", [False, True])
def test_hip_core_tracks_torch_core(model, kw, graph):
    from d4pg_pytorch_amd.models.core import TorchCore
    from d4pg_pytorch_amd.models.hip_core import HipCore
    cfg = _cfg(model, use_hip_graph=graph, **kw)
    hipc = HipCore(cfg, "cuda")
    torc = TorchCore(cfg, "cuda")
    torch.testing.assert_close(hipc.online.flat, torc.online.flat.detach())
    for it in range(6):
        b = _batch(cfg, 100 + it, "cuda")
        p1 = hipc.update(b).clone()
        p2 = torc.update(b)
        torch.testing.assert_close(p1, p2, rtol=2e-3, atol=1e-5)
    # Adam normalises the step, so parameters agree to ~lr * (relative grad error)
    torch.testing.assert_close(hipc.online.flat, torc.online.flat.detach(), rtol=0, atol=2e-5)
    torch.testing.assert_close(hipc.target.flat, torc.target.flat, rtol=0, atol=2e-6)
    mh, mt = hipc.read_metrics(), torc.read_metrics()
    assert mh["count"] == 6 and mt["count"] == 6
    assert abs(mh["value_loss"] - mt["value_loss"]) < 1e-3 * max(1.0, abs(mt["value_loss"]))
    assert abs(mh["policy_loss"] - mt["policy_loss"]) < 1e-3 * max(1.0, abs(mt["policy_loss"]))
    assert hipc.read_dev_state().step_done == 6


def test_hip_core_gradients_match_autograd_exactly_once():
    """One step, compare raw gradients (not Adam-normalised parameters)."""
    from d4pg_pytorch_amd.models.core import TorchCore
    from d4pg_pytorch_amd.models.hip_core import HipCore
    cfg = _cfg("d4pg", use_hip_graph=False, env="Walker2d-v2")
    hipc, torc = HipCore(cfg, "cuda"), TorchCore(cfg, "cuda")
    b = _batch(cfg, 7, "cuda")
    grads = {}
    torc.grad_sync = lambda g: grads.setdefault(len(grads), g.detach().clone())
    hgr = {}
    hipc.grad_sync = lambda g: hgr.setdefault(len(hgr), g.detach().clone())
    torc.update(b)
    hipc.update(b)
    for k in (0, 1):
        scale = grads[k].abs().max().item()
        torch.testing.assert_close(hgr[k], grads[k], rtol=0, atol=2e-4 * scale)
