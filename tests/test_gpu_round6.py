"""GPU tests added in round 6 (VERDICT r05 "next round" items 1, 4, 5 and the ADVICE r05 findings).

* the hand-written exponential of the kernel functions (agp_cavi.h: exp_mhalf / exp_nonpos, 17 VALU instructions, v_ldexp_f64) against
  mpmath over the whole argument range, exact zeros far out, through the ABI's agp_kernelmatrix;
* the branch-free streaming predictor against the materialised K_*m alpha;
* the count of task-graph fallbacks behind the ABI (agp_ctx_task_graph_fallbacks);
* the bounded grid barrier of the in-stream fallback: an over-subscribed fallback grid ends in an error status, not in a hang.
"""
import ctypes as C
import os
import subprocess
import sys
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def mods(built):
    import torch

    assert torch.cuda.is_available()
    import agp_amd as AGP
    from agp_amd import capi

    from oracle import agp_ref as R

    return AGP, R, capi, torch


def _kdesc(capi, kind, variance, scale):
    d = capi.KernelDesc()
    d.kind, d.variance = kind, variance
    d.ard, d.scale, d.ard_scales_host = 0, scale, None
    return d


@pytest.mark.parametrize("kind", [0, 1, 2, 3])
def test_kernel_exponential_over_the_whole_argument_range(mods, kind):
    """k(x, y) for one-dimensional points at distances that drive the exponential's argument from 0 to below -745 (the last finite
    exp in fp64), against 40-digit mpmath: relative error <= 3e-14 + 8 eps |argument| (the polynomial's truncation is 9e-15; the
    argument itself -- a squared distance, a square root -- arrives with a few ulp of relative error, which the exponential turns
    into |argument| times that) wherever the value is a normal number, and EXACTLY zero beyond the underflow threshold -- the
    padded inducing points of the online model rely on exact zeros (online.py, _pad_inducing)."""
    AGP, R, capi, torch = mods
    from mpmath import exp as mexp, mp, mpf, sqrt as msqrt

    mp.dps = 40
    L = capi.lib()
    ctx = C.c_void_p()
    assert L.agp_ctx_create(0, C.c_void_p(torch.cuda.current_stream().cuda_stream), C.byref(ctx)) == 0
    rng = np.random.default_rng(kind)
    # arguments: SE -d^2/2, Matern52 -sqrt5 d, Matern32 -sqrt3 d, Exponential -d
    dmax = {0: 39.0, 1: 340.0, 2: 440.0, 3: 760.0}[kind]
    d = np.concatenate([[0.0, 1e-9, 1e-4], rng.uniform(0, 6, 300), rng.uniform(0, dmax, 300), [dmax, 2 * dmax, 1e6, 1e150]])
    X = np.zeros((1, 1))
    Y = d[:, None].copy()
    xd = torch.tensor(X, dtype=torch.float64, device="cuda")
    yd = torch.tensor(Y, dtype=torch.float64, device="cuda")
    out = torch.empty(1, len(d), dtype=torch.float64, device="cuda")
    kd = _kdesc(capi, kind, 1.0, 1.0)
    assert L.agp_kernelmatrix(ctx, 0, C.byref(kd), xd.data_ptr(), 1, 1, None, yd.data_ptr(), len(d), 1, 1, out.data_ptr(), len(d)) == 0
    got = out.cpu().numpy()[0]

    def ref(t):
        t = mpf(float(t))
        if kind == 0:
            return mexp(-t * t / 2)
        if kind == 1:
            return (1 + msqrt(5) * t + 5 * t * t / 3) * mexp(-msqrt(5) * t)
        if kind == 2:
            return (1 + msqrt(3) * t) * mexp(-msqrt(3) * t)
        return mexp(-t)

    want = [ref(t) for t in d]
    worst = 0.0
    arg = {0: lambda t: t * t / 2, 1: lambda t: 5 ** 0.5 * t, 2: lambda t: 3 ** 0.5 * t, 3: lambda t: t}[kind]
    for g, w, t in zip(got, want, d):
        wf = float(w)
        if wf >= 1e-290:  # relative
            worst = max(worst, abs(float((mpf(float(g)) - w) / w)) / (3e-14 + 8 * 2.2e-16 * arg(t)))
        elif wf == 0.0:  # below half the smallest denormal: exactly zero
            assert g == 0.0, (t, g)
        else:
            # the exponential itself is (nearly) denormal here: it carries few bits, and the Matern kernels multiply it by a
            # polynomial of ~1e5 afterwards -- absolute, at the size of that product's last bit
            assert abs(g - wf) <= 1e-9 * wf + 1e-315, (t, g, wf)
    assert worst <= 1.0, worst  # (in units of the tolerance above)
    assert got[0] == 1.0
    L.agp_ctx_destroy(ctx)


@pytest.mark.parametrize("kname", ["sqexponential", "matern52"])
def test_streaming_predictor_equals_the_materialised_product(mods, kname):
    """predict_f (means) streams K_*m through registers (k_kernelmatrix_mma<..., 1>; round 6: a branch-free epilogue for the
    squared-exponential kernel, variance folded into alpha, no repair of near-coincident points) -- against K_*m from
    agp_kernelmatrix times alpha = K^-1 mu on the host, with test points that COINCIDE with inducing points among them."""
    AGP, R, capi, torch = mods
    rng = np.random.default_rng(5)
    N, D, m, B = 3000, 7, 96, 200
    X = rng.random((N, D))
    y = np.sin(3 * X[:, 0]) + 0.1 * rng.standard_normal(N)
    Z = X[:m].copy()  # the first m test points ARE the inducing points
    kern = {"sqexponential": AGP.SqExponentialKernel, "matern52": AGP.Matern52Kernel}[kname]()
    model = AGP.SVGP(1.7 * AGP.with_lengthscale(kern, 0.6), AGP.GaussianLikelihood(0.05), AGP.AnalyticSVI(B), Z, optimiser=False)
    idx = [rng.choice(N, B, replace=False) for _ in range(5)]
    AGP.train_(model, X, y, 5, idx_stream=idx)
    mu_f = np.asarray(AGP.predict_f(model, X)).reshape(-1)  # (cov=False: the means-only streaming launch)
    mu, Sig, e1, e2 = model.get_state(0)
    rk = R.Kernel(kname, 1.0 / 0.6, 1.7)
    Kmm = rk.matrix(Z, Z) + 1e-4 * np.eye(m)
    want = rk.matrix(X, Z) @ np.linalg.solve(Kmm, mu)
    assert np.max(np.abs(mu_f - want)) <= 1e-10 * np.max(np.abs(want)), np.max(np.abs(mu_f - want)) / np.max(np.abs(want))


_FB_CODE = r"""
import sys, ctypes as C
sys.path.insert(0, '.')
import numpy as np
import __graft_entry__ as g; g.build()
import agp_amd as AGP
from agp_amd import capi
rng = np.random.default_rng(6)
N, D, m, B, iters = 4000, 6, 256, 256, 4
X = rng.random((N, D)); f = np.sin(4 * X[:, 0]) + X[:, 1] - 0.8
y = np.sign(f + 0.3 * rng.standard_normal(N))
Z = X[rng.permutation(N)[:m]].copy()
idx = [rng.choice(N, B, replace=False) for _ in range(iters)]
ma = AGP.SVGP(AGP.SqExponentialKernel() @ AGP.ScaleTransform(3.0), AGP.LogisticLikelihood(), AGP.AnalyticSVI(B), Z, optimiser=False)
try:
    AGP.train_(ma, X, y, iters, idx_stream=idx)
    e2 = ma.get_state(0)[3]
    print('OK', float(np.abs(e2).max()))
except capi.AGPError as e:
    print('AGPERROR', e.status, str(e)[:200])
n = C.c_int64(-1)
assert capi.lib().agp_ctx_task_graph_fallbacks(ma._ctx, C.byref(n)) == 0
print('FALLBACKS', n.value)
"""


def _child(env_extra, code, timeout=300):
    env = dict(os.environ, **env_extra)
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    return r.stdout


def test_fallback_counter_behind_the_abi(mods):
    import _knobs

    if _knobs.no_task_graph():
        pytest.skip("AGP_CHOL_DAG=0: no task-graph launches, hence nothing for the fallback to re-run")
    _fallback_counter_body()


def _fallback_counter_body():
    """agp_ctx_task_graph_fallbacks: 0 for a run on a GPU the process has to itself, the number of re-run launches under the test
    hook that aborts every task-graph launch of a CAVI step (bench.py prints the same count as `task_graph_fallbacks`)."""
    out = _child({}, _FB_CODE)
    assert "FALLBACKS 0" in out and "OK" in out, out
    out = _child({"AGP_DAG_TEST_ABORT": "1"}, _FB_CODE)
    n = int([l for l in out.splitlines() if l.startswith("FALLBACKS")][0].split()[1])
    assert n >= 1 and "OK" in out, out


def test_an_oversubscribed_fallback_ends_in_an_error_status_not_in_a_hang(mods):
    import _knobs

    if _knobs.no_task_graph():
        pytest.skip("AGP_CHOL_DAG=0: no task-graph launches, hence no in-stream fallback")
    _oversubscribed_body()


def _oversubscribed_body():
    """The in-stream fallback separates its phases by grid barriers and relies on all of its workgroups being resident.  Until round 5
    a workgroup that could not get a compute unit made the others spin forever (docs/DESIGN_LOG.md section 14, "the one unbounded
    wait"); now the barrier is limited on the device's 100 MHz clock (8 s) and latches status -3.  The hook makes the fallback's
    grid four workgroups per CU (each needs most of a CU's LDS), every launch is aborted: the run must come back within the limit
    plus slack with AGP_ERR_HIP and a message that names the barrier -- and not hang."""
    t0 = time.time()
    out = _child({"AGP_DAG_TEST_ABORT": "1", "AGP_DAG_TEST_OVERSUBSCRIBE": "1"},
                 _FB_CODE.replace("m, B, iters = 4000, 6, 256, 256, 4", "m, B, iters = 4000, 6, 1024, 1024, 2"), timeout=200)
    dt = time.time() - t0
    assert "AGPERROR 7" in out and "grid barrier" in out, out
    assert dt < 120, dt


def test_reloaded_multioutput_model_keeps_optimising_its_mixing_weights(mods, tmp_path):
    """load_trained_model built the MOSVGP with Aoptimiser=False, so a resumed run froze A silently.  The optimiser's
    rule now travels in the file; a reloaded model's A moves on the next steps (its moments restart, as documented), and a model saved
    with Aoptimiser=False stays frozen."""
    AGP, R, capi, torch = mods
    rng = np.random.default_rng(8)
    N, D, m, B, Q = 600, 3, 24, 100, 3
    X = rng.random((N, D))
    ys = [np.sin(4 * X[:, 0]) + 0.1 * rng.standard_normal(N), np.sign(X[:, 1] - 0.5 + 0.1 * rng.standard_normal(N))]
    A = rng.standard_normal((2, Q))
    A /= np.linalg.norm(A, axis=1, keepdims=True)
    Zs = [X[rng.permutation(N)[:m]].copy() for _ in range(Q)]
    idx = [rng.choice(N, B, replace=False) for _ in range(8)]
    for aopt in (AGP.ADAM(0.02), False):
        ma = AGP.MOSVGP(AGP.SqExponentialKernel() @ AGP.ScaleTransform(3.0), [AGP.GaussianLikelihood(0.1), AGP.LogisticLikelihood()],
                        AGP.AnalyticSVI(B), Zs, A=A.copy(), Aoptimiser=aopt, optimiser=False)
        AGP.train_(ma, X, ys, 4, idx_stream=idx[:4])
        path = str(tmp_path / f"mo_{bool(aopt)}.npz")
        AGP.save_trained_model(path, ma)
        mb = AGP.load_trained_model(path)
        A0 = np.array(mb.get_A())
        assert np.allclose(A0, ma.get_A(), atol=0, rtol=0)
        AGP.train_(mb, X, ys, 4, idx_stream=idx[4:])
        moved = float(np.max(np.abs(np.array(mb.get_A()) - A0)))
        if aopt:
            assert mb.A_opt is not None and abs(mb.A_opt.eta - 0.02) < 1e-15
            assert moved > 1e-4, moved
        else:
            assert mb.A_opt is None and moved == 0.0, moved


@pytest.mark.parametrize("m,B,EVAL,dtype", [(64, 128, 512, "f64"), (1024, 1024, 2048, "f64"), (256, 256, 1024, "f32")])
def test_side_stream_objective_equals_the_inline_one(mods, m, B, EVAL, dtype):
    """SideObjective (round 6): the ELBO check of a training loop evaluated on a side stream from a snapshot of (eta1, eta2), next to
    the steps that follow, against agp_svgp_elbo (fresh local variables) evaluated in line on the training handle at the same
    points of the same run.  Trajectories and values agree to rounding -- the values to a few ulp, not bitwise: the shadow handle forms Sigma = Xa' Xa with the product workgroups inside its factorisation launch
    (agp_svgp_set_state -> refactor), the in-line evaluation on a handle in the middle of training with the stand-alone balanced
    product, and the two add in different orders.  The side evaluations are really in flight while training continues (tickets
    fetched two checks later)."""
    AGP, R, capi, torch = mods
    L = capi.lib()
    rng = np.random.default_rng(21)
    N, D, iters, every = 6000, 5, 24, 4
    X = rng.random((N, D))
    y = np.sign(np.sin(4 * X[:, 0]) + X[:, 1] - 0.8 + 0.3 * rng.standard_normal(N))
    Z = X[rng.permutation(N)[:m]].copy()
    idx = np.stack([rng.choice(N, B, replace=False) for _ in range(iters)]).astype(np.int64)
    eval_idx = torch.as_tensor(rng.choice(N, EVAL, replace=False).astype(np.int64), device="cuda")
    T = np.float64 if dtype == "f64" else np.float32

    def make():
        mdl = AGP.SVGP(AGP.SqExponentialKernel() @ AGP.ScaleTransform(3.0), AGP.LogisticLikelihood(), AGP.AnalyticSVI(B), Z,
                       optimiser=False, T=T)
        mdl._ensure_handle(max(EVAL, B))  # (capacity for the in-line evaluation on EVAL points)
        AGP.train_(mdl, X, y, 1, idx_stream=[idx[0]])  # binds the data
        return mdl

    def steps(mdl, i0, i1):
        Xd, yd, _ = mdl._data
        ia = torch.as_tensor(idx, device="cuda")
        for i in range(i0, i1):
            assert L.agp_svgp_cavi_step(mdl._h, C.c_void_p(Xd.data_ptr()), Xd.stride(0), C.c_void_p(yd.data_ptr()),
                                        C.c_void_p(ia[i].data_ptr()), B, N / B) == 0
            if i + 1 < iters:
                L.agp_svgp_prefetch(mdl._h, C.c_void_p(Xd.data_ptr()), Xd.stride(0), C.c_void_p(ia[i + 1].data_ptr()), B)
        return Xd, yd

    # run A: in line
    ma = make()
    inline = []
    for i0 in range(1, iters, every):
        Xd, yd = steps(ma, i0, min(i0 + every, iters))
        e = C.c_double()
        ma._chk(L.agp_svgp_elbo(ma._h, C.c_void_p(Xd.data_ptr()), Xd.stride(0), C.c_void_p(yd.data_ptr()),
                                C.c_void_p(eval_idx.data_ptr()), EVAL, N / EVAL, 1, C.byref(e)))
        inline.append(e.value)
    # run B: side stream, two tickets in flight
    mb = make()
    side = AGP.SideObjective(mb, EVAL)
    tickets, got = [], []
    for i0 in range(1, iters, every):
        Xd, yd = steps(mb, i0, min(i0 + every, iters))
        tickets.append(side.enqueue(Xd, yd, eval_idx, EVAL, N / EVAL))
        if len(tickets) > 2:
            got.append(side.fetch(tickets.pop(0)))
    got += [side.fetch(t) for t in tickets]
    assert len(got) == len(inline) and all(np.isfinite(got))
    # (the ELBO is a difference of terms of size ~N: the agreement is relative to those, not to a value that happens to be near zero)
    tol = 1e-12 if dtype == "f64" else 1e-5
    assert np.allclose(got, inline, rtol=tol, atol=tol * N), (got, inline)
    # the trajectories of the three runs -- in-line checks (A), side-stream checks (B), no checks at all (C) -- agree to rounding: a
    # check takes the pending natural-gradient step with the stand-alone kernel instead of the next launch's prologue, and an in-line
    # evaluation also leaves the factorisation of -2 eta2 with its inverse behind for the next step (other summation orders)
    mc = make()
    steps(mc, 1, iters)
    e2b, e2c, e2a = mb.get_state(0)[3], mc.get_state(0)[3], ma.get_state(0)[3]
    tolt = (1e-9 if dtype == "f64" else 1e-3) * np.max(np.abs(e2c))
    assert np.max(np.abs(e2b - e2c)) <= tolt and np.max(np.abs(e2a - e2c)) <= tolt
    # and the values are the oracle's ELBO of that posterior (rtol: fp32 states are compared in double)
    assert inline[-1] < 0 and abs(inline[-1] - inline[-2]) < abs(inline[0])


@pytest.mark.parametrize("likname", ["bayesiansvm", "negbinomial"])
def test_device_fixed_points_of_the_8f2_likelihoods_maximise_their_collapsed_bounds(mods, likname):
    """The HIP path itself against scipy's argmax of bounds written down from the literature (tests/test_oracle_third_party.py, round 6):
    BayesianSVM -- the collapsed location-scale-mixture bound -sqrt(E(1 - y f)^2) - (1 - y E f) of Polson & Scott / Wenzel et al.;
    NegBinomial -- the Jaakkola-Jordan / Polya-Gamma bound with its Polya-Gamma part DOUBLED, which is what the reference's
    theta = (r + y) tanh(c/2) / c (negativebinomial.jl:78: twice E[omega]) ascends.  Full-batch AnalyticVI to the fixed point."""
    AGP, R, capi, torch_ = mods
    import torch
    from test_oracle_third_party import _maximise_collapsed_bound, _sparse_pieces, _toy_sparse

    rng = np.random.default_rng(71)
    X, f, Z = _toy_sparse(rng)
    if likname == "bayesiansvm":
        y = np.where(f + 0.3 * rng.standard_normal(len(f)) > 0, 1.0, -1.0)
        lik, kern, ka = AGP.BayesianSVM(), R.Kernel("sqexponential", 2.0, 1.5), 1.5 * (AGP.SqExponentialKernel() @ AGP.ScaleTransform(2.0))

        def term(mf, vf, yy):
            return -torch.sqrt((1.0 - yy * mf) ** 2 + vf) - (1.0 - yy * mf)
    else:
        r = 4.0
        y = rng.negative_binomial(int(r), 1.0 / (1.0 + np.exp(0.7 * f))).astype(np.int64)
        lik, kern, ka = AGP.NegBinomialLikelihood(r), R.Kernel("sqexponential", 2.0, 1.2), 1.2 * (AGP.SqExponentialKernel() @ AGP.ScaleTransform(2.0))

        def term(mf, vf, yy):
            c = torch.sqrt(mf ** 2 + vf)
            return 0.5 * (yy - r) * mf + 2.0 * (yy + r) * (torch.nn.functional.logsigmoid(c) - 0.5 * c)
    m = AGP.SVGP(ka, lik, AGP.AnalyticVI(), Z, optimiser=False)
    AGP.train_(m, X, y, 800)
    mu_d, Sig_d, e1, e2 = m.get_state(0)
    K, kappa, Kt = _sparse_pieces(kern, X, Z, 1e-4)
    mu, Sig, val = _maximise_collapsed_bound(term, K, kappa, Kt, np.asarray(y, dtype=np.float64), 0.9 * mu_d, np.linalg.cholesky(1.1 * Sig_d))
    assert np.max(np.abs(mu_d - mu)) < 2e-6 * np.max(np.abs(mu)) and np.max(np.abs(Sig_d - Sig)) < 2e-6 * np.max(np.abs(Sig))


def test_evaluation_batch_kappa_cache_and_its_invalidation(mods):
    """Round 6: a handle that evaluates ELBO(model, X[idx], y[idx]) (fresh local variables) again and again on the SAME batch -- the
    shadow handle of a SideObjective -- keeps K_nm and kappa = K_nm K^-1 of that batch between the calls (pointer identity, like the
    full-batch kappa cache).  The cached evaluation equals the uncached one of a new handle in the same state; another index buffer,
    agp_svgp_invalidate_data after an in-place change of X, a new kernel and a training step in between all recompute."""
    AGP, R, capi, torch = mods
    L = capi.lib()
    rng = np.random.default_rng(5)
    N, D, m, B, EVAL = 3000, 4, 64, 128, 512
    X = rng.random((N, D))
    y = np.sign(np.sin(4 * X[:, 0]) + X[:, 1] - 0.8 + 0.3 * rng.standard_normal(N))
    Z = X[rng.permutation(N)[:m]].copy()
    idx = [rng.choice(N, B, replace=False) for _ in range(6)]
    ia = torch.as_tensor(rng.choice(N, EVAL, replace=False).astype(np.int64), device="cuda")
    ib = torch.as_tensor(rng.choice(N, EVAL, replace=False).astype(np.int64), device="cuda")

    def make(scale=3.0):
        mdl = AGP.SVGP(AGP.SqExponentialKernel() @ AGP.ScaleTransform(scale), AGP.LogisticLikelihood(), AGP.AnalyticSVI(B), Z, optimiser=False)
        mdl._ensure_handle(EVAL)
        AGP.train_(mdl, X, y, 3, idx_stream=idx[:3])
        return mdl

    def elbo(mdl, it):
        Xd, yd, _ = mdl._data
        e = C.c_double()
        mdl._chk(L.agp_svgp_elbo(mdl._h, C.c_void_p(Xd.data_ptr()), Xd.stride(0), C.c_void_p(yd.data_ptr()), C.c_void_p(it.data_ptr()),
                                 EVAL, N / EVAL, 1, C.byref(e)))
        return e.value

    a = make()
    v1 = elbo(a, ia)
    v1b = elbo(a, ia)                       # cached kappa, same state
    assert v1b == pytest.approx(v1, rel=1e-13)
    # another state through set_state (what the shadow handle of a SideObjective sees), cached kappa against a new handle
    b = make()
    AGP.train_(b, X, y, 3, idx_stream=idx[3:])
    mu, Sig, e1, e2 = b.get_state(0)
    a.set_state(0, e1, e2)
    v2 = elbo(a, ia)
    assert v2 == pytest.approx(elbo(b, ia), rel=1e-12) and abs(v2 - v1) > 1e-6 * abs(v1)
    # another index buffer: recomputed
    assert elbo(a, ib) == pytest.approx(elbo(b, ib), rel=1e-12)
    # in-place change of the data + agp_svgp_invalidate_data
    Xd, yd, _ = a._data
    v_before = elbo(a, ia)
    Xd[ia[:50]] += 0.05
    a._chk(L.agp_svgp_invalidate_data(a._h))
    v_after = elbo(a, ia)
    assert abs(v_after - v_before) > 1e-8 * abs(v_before)
    Xb = b._data[0]
    Xb[ia[:50]] += 0.05
    b._chk(L.agp_svgp_invalidate_data(b._h))
    assert v_after == pytest.approx(elbo(b, ia), rel=1e-12)
