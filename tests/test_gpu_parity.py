"""GPU parity tests: the HIP path (through the C ABI) against the NumPy oracle on identical seeded inputs.

Tolerances (fp64): single operations <= 1e-11 relative; posterior / predictive quantities after several CAVI steps
<= 1e-8 relative (jitter 1e-4 bounds cond(K)).  fp32: 2e-3 relative on mu_f (Q6).
"""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


@pytest.fixture(scope="module")
def env(built):
    import torch

    assert torch.cuda.is_available(), "these tests need the MI355X"
    import agp_amd as AGP
    from agp_amd import capi
    from oracle import agp_ref as R

    L = capi.lib()
    ctx = C.c_void_p()
    st = L.agp_ctx_create(0, C.c_void_p(torch.cuda.current_stream().cuda_stream), C.byref(ctx))
    assert st == 0
    yield dict(torch=torch, AGP=AGP, capi=capi, R=R, L=L, ctx=ctx)
    L.agp_ctx_destroy(ctx)


def _kdesc(capi, kind, variance, scale, ard=None):
    d = capi.KernelDesc()
    d.kind, d.variance = kind, variance
    keep = None
    if ard is not None:
        keep = (C.c_double * len(ard))(*ard)
        d.ard, d.scale, d.ard_scales_host = 1, 1.0, C.cast(keep, C.POINTER(C.c_double))
    else:
        d.ard, d.scale, d.ard_scales_host = 0, scale, None
    return d, keep


KINDS = [("sqexponential", 0), ("matern52", 1), ("matern32", 2), ("exponential", 3)]


@pytest.mark.parametrize("kname,kid", KINDS)
@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_kernelmatrix(env, kname, kid, dtype):
    torch, capi, R, L, ctx = env["torch"], env["capi"], env["R"], env["L"], env["ctx"]
    rng = np.random.default_rng(1)
    n, p, D = 150, 77, 11  # ragged on purpose: nothing is a multiple of 64 / 32
    X, Y = rng.random((n, D)), rng.random((p, D))
    ard = rng.random(D) + 0.5
    td = torch.float64 if dtype == "f64" else torch.float32
    tol = 1e-12 if dtype == "f64" else 2e-5
    for scale, a in [(1.7, None), (1.0, ard)]:
        ref = R.Kernel(kname, a if a is not None else scale, 1.3).matrix(X, Y)
        kd, keep = _kdesc(capi, kid, 1.3, scale, a)
        xd = torch.tensor(X, dtype=td, device="cuda")
        yd = torch.tensor(Y, dtype=td, device="cuda")
        out = torch.empty(n, p, dtype=td, device="cuda")
        st = L.agp_kernelmatrix(ctx, 0 if dtype == "f64" else 1, C.byref(kd), xd.data_ptr(), n, D, None,
                                yd.data_ptr(), p, D, D, out.data_ptr(), p)
        assert st == 0, L.agp_last_error(ctx)
        assert _rel(out.cpu().numpy(), ref) < tol
    # gathered rows + symmetric form
    idx = rng.choice(n, 40, replace=False)
    kd, keep = _kdesc(capi, kid, 1.0, 2.0)
    xd = torch.tensor(X, dtype=td, device="cuda")
    idd = torch.tensor(idx, dtype=torch.int64, device="cuda")
    out = torch.empty(40, 40, dtype=td, device="cuda")
    st = L.agp_kernelmatrix(ctx, 0 if dtype == "f64" else 1, C.byref(kd), xd.data_ptr(), 40, D, idd.data_ptr(), None,
                            0, 0, D, out.data_ptr(), 40)
    assert st == 0
    # symmetric call with idx: rows gathered, columns are x[0:40] -> compare accordingly
    ref = R.Kernel(kname, 2.0, 1.0).matrix(X[idx], X[:40])
    assert _rel(out.cpu().numpy(), ref) < tol


# 2048: largest task-graph size; 2112: per-column launches; 6144: blocked (diagonal block / panel / trailing launches, look-ahead)
@pytest.mark.parametrize("n", [5, 64, 100, 257, 1024, 2048, 2112, 6144])
def test_potrf_and_inverse(env, n):
    torch, L, ctx = env["torch"], env["L"], env["ctx"]
    rng = np.random.default_rng(n)
    G = rng.standard_normal((n, n + 3))
    A = G @ G.T / n + 0.5 * np.eye(n)
    ad = torch.tensor(A, dtype=torch.float64, device="cuda")
    info = C.c_int32(-1)
    st = L.agp_potrf_jitter(ctx, 0, ad.data_ptr(), n, n, 1e-4, C.byref(info))
    assert st == 0 and info.value == 0, L.agp_last_error(ctx)
    Lref = np.linalg.cholesky(A + 1e-4 * np.eye(n))
    assert _rel(ad.cpu().numpy(), Lref) < 1e-11
    a2 = torch.tensor(A, dtype=torch.float64, device="cuda")
    inv = torch.empty(n, n, dtype=torch.float64, device="cuda")
    ld = C.c_double()
    st = L.agp_spd_inverse(ctx, 0, a2.data_ptr(), n, n, inv.data_ptr(), n, C.byref(ld), C.byref(info))
    assert st == 0 and info.value == 0
    assert _rel(inv.cpu().numpy(), np.linalg.inv(A)) < 1e-10
    assert abs(ld.value - np.linalg.slogdet(A)[1]) < 1e-9 * max(1.0, abs(ld.value))
    # X = B / A
    r = 37
    Bm = rng.standard_normal((r, n))
    bd = torch.tensor(Bm, dtype=torch.float64, device="cuda")
    xd = torch.empty(r, n, dtype=torch.float64, device="cuda")
    st = L.agp_solve_right_spd(ctx, 0, a2.data_ptr(), n, n, bd.data_ptr(), n, r, xd.data_ptr(), n, C.byref(info))
    assert st == 0
    assert _rel(xd.cpu().numpy(), np.linalg.solve(A, Bm.T).T) < 1e-10


def test_task_graph_cholesky_repeated(env):
    """The one-launch task-graph factorisation hands tiles between workgroups with coherent stores / loads and flags instead of
    fences: 60 different matrices through the same context (flags are epoch-stamped, never reset), factor and full inverse
    checked every time -- a stale or torn tile would show as an O(1) error."""
    torch, L, ctx = env["torch"], env["L"], env["ctx"]
    rng = np.random.default_rng(77)
    n = 1024
    info = C.c_int32(-1)
    ld = C.c_double()
    inv = torch.empty(n, n, dtype=torch.float64, device="cuda")
    for rep in range(60):
        G = rng.standard_normal((n, n + 8))
        A = G @ G.T / n + (0.2 + 0.01 * rep) * np.eye(n)
        a = torch.tensor(A, dtype=torch.float64, device="cuda")
        assert L.agp_spd_inverse(ctx, 0, a.data_ptr(), n, n, inv.data_ptr(), n, C.byref(ld), C.byref(info)) == 0
        assert info.value == 0
        got = inv.cpu().numpy()
        assert _rel(got @ A, np.eye(n)) < 1e-9, rep
        assert abs(ld.value - np.linalg.slogdet(A)[1]) < 1e-8 * abs(ld.value)


@pytest.mark.parametrize("likname", ["logisticsoftmax", "heteroscedastic"])
def test_multi_latent_task_graph_matches_oracle(env, likname):
    """Several latents (3-class LogisticSoftMax, the two heteroscedastic latents) with m = 150 inducing points: three block
    columns, so their factorisations run as ONE interleaved task-graph launch with dependencies between tiles (the toy models
    of the other tests have a single block column)."""
    AGP, R = env["AGP"], env["R"]
    rng = np.random.default_rng(17)
    B, iters = 100, 5
    X, y, ma, mr = _models(env, likname, rng, True, B, m=150, N=500)
    idx = [rng.choice(len(X), B, replace=False) for _ in range(iters)]
    AGP.train_(ma, X, y, iters, idx_stream=idx)
    mr.train(X, y, iters, idx_stream=idx)
    assert ma.n_latent >= 2
    for k in range(ma.n_latent):
        mu, Sig, e1, e2 = ma.get_state(k)
        g = mr.latents[k]
        assert _rel(e1, g.eta1) < 1e-8 and _rel(e2, g.eta2) < 1e-8
        assert _rel(mu, g.mu) < 1e-7 and _rel(np.diag(Sig), np.diag(g.Sigma)) < 1e-7


@pytest.mark.parametrize("B", [12800, 16000])
def test_large_minibatch_either_cholesky_driver(env, B):
    """m = 64 with a very large minibatch: B = 12800 still runs the task-graph Cholesky (203 tiles in its single block column),
    B = 16000 exceeds the residency bound and takes the per-column launches; both against the oracle."""
    AGP, R = env["AGP"], env["R"]
    rng = np.random.default_rng(5)
    X, y, ma, mr = _models(env, "logistic", rng, True, B, m=64, N=20000)
    idx = [rng.choice(len(X), B, replace=False) for _ in range(2)]
    AGP.train_(ma, X, y, 2, idx_stream=idx)
    mr.train(X, y, 2, idx_stream=idx)
    mu, Sig, e1, e2 = ma.get_state(0)
    g = mr.latents[0]
    assert _rel(e1, g.eta1) < 1e-8 and _rel(e2, g.eta2) < 1e-8 and _rel(mu, g.mu) < 1e-7


def test_training_is_bitwise_reproducible(env):
    """Every reduction in the library has a fixed order, so the same run twice must agree BITWISE; with the fence-free tile
    hand-over of the task-graph Cholesky this doubles as a race detector (tools/soak_determinism.py runs 2 x 20000 steps)."""
    AGP = env["AGP"]
    rng = np.random.default_rng(3)
    N, D, m, B, iters = 4000, 8, 256, 256, 150
    X = rng.random((N, D))
    y = np.sign(np.sin(X @ rng.standard_normal(D)) + 0.1 * rng.standard_normal(N))
    Z = X[rng.permutation(N)[:m]].copy()
    idx = [rng.choice(N, B, replace=False) for _ in range(iters)]
    res = []
    for _ in range(2):
        model = AGP.SVGP(AGP.with_lengthscale(AGP.SqExponentialKernel(), 0.7), AGP.LogisticLikelihood(), AGP.AnalyticSVI(B), Z,
                         optimiser=False)
        AGP.train_(model, X, y, iters, idx_stream=idx)
        res.append(model.get_state(0))
    for a, b in zip(res[0], res[1]):
        assert np.array_equal(a, b)


def test_potrf_not_posdef(env):
    torch, L, ctx, capi = env["torch"], env["L"], env["ctx"], env["capi"]
    n = 130
    A = np.eye(n)
    A[100, 100] = -1.0  # leading minor 101 fails, like LAPACK info = 101
    ad = torch.tensor(A, dtype=torch.float64, device="cuda")
    info = C.c_int32(0)
    st = L.agp_potrf_jitter(ctx, 0, ad.data_ptr(), n, n, 0.0, C.byref(info))
    assert st == 2 and info.value == 101  # AGP_ERR_NOT_POSDEF


def _toy(rng, N=300, D=3, m=20):
    X = rng.random((N, D))
    f = np.sin(3 * X[:, 0]) + X[:, 1] ** 2 - 0.7
    Z = X[rng.permutation(N)[:m]].copy()
    return X, f, Z


def _models(env, likname, rng, stochastic, B=64, T=np.float64, m=20, N=300):
    AGP, R = env["AGP"], env["R"]
    X, f, Z = _toy(rng, N=N, m=m)
    if likname == "gaussian":
        y = f + 0.1 * rng.standard_normal(len(f))
        la, lr = AGP.GaussianLikelihood(0.01), R.GaussianLikelihood(0.01)
    elif likname == "logistic":
        y = (f + 0.2 * rng.standard_normal(len(f)) > 0).astype(int)
        la, lr = AGP.LogisticLikelihood(), R.LogisticLikelihood()
    elif likname == "studentt":
        y = f + 0.1 * rng.standard_t(3, len(f))
        la, lr = AGP.StudentTLikelihood(3.0), R.StudentTLikelihood(3.0)
    elif likname == "logisticsoftmax":
        y = 1 + np.digitize(f, np.quantile(f, [0.33, 0.66]))
        la, lr = AGP.LogisticSoftMaxLikelihood(3), R.LogisticSoftMaxLikelihood(3)
    else:
        from _liks import agp_lik, labels, oracle_lik

        y = labels(likname, f, X, rng)
        la, lr = agp_lik(AGP, likname), oracle_lik(R, likname)
    ka = 1.5 * (AGP.SqExponentialKernel() @ AGP.ScaleTransform(2.0))
    kr = R.Kernel("sqexponential", 2.0, 1.5)
    inf = AGP.AnalyticSVI(B) if stochastic else AGP.AnalyticVI()
    ma = AGP.SVGP(ka, la, inf, Z, optimiser=False, T=T)
    mr = R.SVGP(kr, lr, Z, stochastic=stochastic, batchsize=B, jitter=1e-4 if T == np.float64 else 1e-3)
    return X, y, ma, mr


NEW_LIKS = ["laplace", "bayesiansvm", "poisson", "negbinomial", "heteroscedastic"]


@pytest.mark.parametrize("likname", ["gaussian", "logistic", "studentt", "logisticsoftmax"] + NEW_LIKS)
@pytest.mark.parametrize("stochastic", [False, True])
def test_cavi_trajectory_fp64(env, likname, stochastic):
    AGP, R, capi = env["AGP"], env["R"], env["capi"]
    rng = np.random.default_rng(7)
    B, iters = 64, 6
    X, y, ma, mr = _models(env, likname, rng, stochastic, B)
    idx = [rng.choice(len(X), B, replace=False) for _ in range(iters)]
    elbos_a, elbos_r = [], []
    AGP.train_(ma, X, y, iters, idx_stream=idx, callback=lambda m, s, i: elbos_a.append(AGP.objective(m, s)))
    yt = R.treat_labels(y, mr.likelihood)
    mr.train(X, yt, iters, idx_stream=idx, labels_treated=True, callback=lambda M, it, xb, yb: elbos_r.append(M.elbo(yb)))
    for k in range(ma.n_latent):
        mu, Sig, e1, e2 = ma.get_state(k)
        g = mr.latents[k]
        assert _rel(e1, g.eta1) < 1e-9
        assert _rel(e2, g.eta2) < 1e-9
        assert _rel(mu, g.mu) < 1e-8
        assert _rel(Sig, g.Sigma) < 1e-8
    assert np.allclose(elbos_a, elbos_r, rtol=1e-8, atol=1e-7), (elbos_a, elbos_r)
    # last-step intermediates
    nb = B if stochastic else len(X)
    g = mr.latents[0]
    assert _rel(ma.get_matrix(capi.MAT_KAPPA, 0, nb), g.kappa) < 1e-9
    assert _rel(ma.get_matrix(capi.VEC_KTILDE, 0, nb), g.Kt) < 1e-8
    # predictions
    Xt = rng.random((131, X.shape[1]))
    if hasattr(mr.likelihood, "lam"):
        assert ma.likelihood.lam == pytest.approx(mr.likelihood.lam, rel=1e-10)
    if likname == "heteroscedastic":
        mf, vf = AGP.predict_f(ma, Xt, cov=True)
        mfr, vfr = mr.predict_f(Xt, cov=True)
        for k in range(2):
            assert _rel(mf[k], mfr[k]) < 1e-8 and _rel(vf[k], vfr[k]) < 1e-7
        pa, pr = AGP.proba_y(ma, Xt), mr.proba_y(Xt)
        assert _rel(pa[0], pr[0]) < 1e-8 and _rel(pa[1], pr[1]) < 1e-7
        assert _rel(AGP.predict_y(ma, Xt), mr.predict_y(Xt)) < 1e-8
    elif ma.n_latent == 1:
        mf, vf = AGP.predict_f(ma, Xt, cov=True)
        mfr, vfr = mr.predict_f(Xt, cov=True)
        assert _rel(mf, mfr[0]) < 1e-8 and _rel(vf, vfr[0]) < 1e-7
        pa = AGP.proba_y(ma, Xt)
        pr = mr.proba_y(Xt)
        assert _rel(pa[0], pr[0]) < 1e-8 and _rel(pa[1], pr[1]) < 1e-6
        if likname in ("gaussian", "studentt", "laplace", "poisson", "negbinomial"):
            assert _rel(AGP.predict_y(ma, Xt), mr.predict_y(Xt)) < 1e-8
        else:
            assert np.array_equal(np.asarray(AGP.predict_y(ma, Xt)), np.asarray(mr.predict_y(Xt)))
    else:
        mf = AGP.predict_f(ma, Xt)
        mfr = mr.predict_f(Xt)
        for k in range(ma.n_latent):
            assert _rel(mf[k], mfr[k]) < 1e-8
        assert np.array_equal(AGP.predict_y(ma, Xt), mr.predict_y(Xt))
        pa, pr = AGP.proba_y(ma, Xt), mr.proba_y(Xt)
        for k, lab in enumerate([1, 2, 3]):
            assert _rel(pa[lab], pr[:, k]) < 1e-8
    # external ELBO with fresh local variables
    ea = AGP.ELBO(ma, X, y, rho=1.0)
    er = mr.elbo_fresh(X, yt, 1.0)
    assert abs(ea - er) < 1e-7 * max(1.0, abs(er))


def test_titsias_optimum_gaussian(env):
    """KAT-1: Gaussian likelihood, full batch: after ONE step eta1 = kappa' y / s2, eta2 = -(kappa'kappa/s2 + Kinv)/2
    and a second step is a fixed point."""
    AGP, R, capi = env["AGP"], env["R"], env["capi"]
    rng = np.random.default_rng(3)
    X, y, ma, mr = _models(env, "gaussian", rng, False)
    AGP.train_(ma, X, y, 1)
    mu1, S1, e1, e2 = ma.get_state(0)
    kern = R.Kernel("sqexponential", 2.0, 1.5)
    K, Lk = R.compute_K(kern, ma.Zs[0], 1e-4)
    _, kappa, _ = R.compute_kappa(kern, X, ma.Zs[0], Lk, 1e-4)
    Kinv = np.linalg.inv(K)
    assert _rel(e1, kappa.T @ y / 0.01) < 1e-9
    assert _rel(e2, -0.5 * (kappa.T @ kappa / 0.01 + Kinv)) < 1e-9
    AGP.train_(ma, X, y, 1, state=True)
    mu2, S2, _, _ = ma.get_state(0)
    assert _rel(mu2, mu1) < 1e-9 and _rel(S2, S1) < 1e-9


@pytest.mark.parametrize("likname", ["studentt", "logistic", "laplace", "negbinomial", "poisson", "heteroscedastic"])
def test_fp32_mode(env, likname):
    """T = Float32 (SVGP.jl:43): same trajectory in fp32 arithmetic (jitter 1e-3), predictive mean within 2e-3 of the fp64 oracle"""
    AGP, R = env["AGP"], env["R"]
    rng = np.random.default_rng(11)
    B, iters = 64, 5
    X, y, ma, mr = _models(env, likname, rng, True, B, T=np.float32)
    idx = [rng.choice(len(X), B, replace=False) for _ in range(iters)]
    AGP.train_(ma, X, y, iters, idx_stream=idx)
    mr.train(X, y, iters, idx_stream=idx)
    Xt = rng.random((50, X.shape[1]))
    mf = AGP.predict_f(ma, Xt)
    mfr = mr.predict_f(Xt)
    if ma.n_latent == 1:
        assert _rel(mf, mfr[0]) < 2e-3
    else:
        for k in range(ma.n_latent):
            assert _rel(mf[k], mfr[k]) < 5e-3


def test_error_mapping(env):
    AGP, capi = env["AGP"], env["capi"]
    rng = np.random.default_rng(0)
    X, f, Z = _toy(rng)
    # duplicate inducing points with zero jitter-resistance: K is singular beyond jitter? -> still SPD with jitter;
    # force NOT_POSDEF through a state whose -2*eta2 is indefinite
    m = AGP.SVGP(AGP.SqExponentialKernel(), AGP.GaussianLikelihood(0.1), AGP.AnalyticVI(), Z, optimiser=False)
    AGP.train_(m, X, f, 1)
    mu, S, e1, e2 = m.get_state(0)
    bad = e2.copy()
    bad[0, 0] = +1.0
    with pytest.raises(capi.AGPError) as ei:
        m.set_state(0, e1, bad)
    assert ei.value.status == 2
    with pytest.raises(ValueError):
        AGP.train_(AGP.SVGP(AGP.SqExponentialKernel(), AGP.GaussianLikelihood(0.1), AGP.AnalyticSVI(10 ** 6), Z,
                            optimiser=False), X, f, 1)


def test_large_m1024_step_matches_oracle(env):
    """One C2-shaped step (m = 1024, B = 1024, D = 32, logistic) against the oracle."""
    AGP, R = env["AGP"], env["R"]
    rng = np.random.default_rng(5)
    N, D, m, B = 4096, 32, 1024, 1024
    X = rng.random((N, D))
    w = rng.standard_normal(D)
    y = np.sign(np.sin(X @ w) + 0.1 * rng.standard_normal(N))
    Z = X[rng.permutation(N)[:m]].copy()
    ell = np.sqrt(D) / 4
    idx = [rng.choice(N, B, replace=False) for _ in range(3)]
    ma = AGP.SVGP(AGP.with_lengthscale(AGP.SqExponentialKernel(), ell), AGP.LogisticLikelihood(), AGP.AnalyticSVI(B), Z,
                  optimiser=False)
    AGP.train_(ma, X, y, 3, idx_stream=idx)
    mr = R.SVGP(R.Kernel("sqexponential", 1 / ell, 1.0), R.LogisticLikelihood(), Z, stochastic=True, batchsize=B)
    mr.train(X, y, 3, idx_stream=idx)
    mu, Sig, e1, e2 = ma.get_state(0)
    g = mr.latents[0]
    assert _rel(e2, g.eta2) < 1e-9 and _rel(mu, g.mu) < 1e-8 and _rel(np.diag(Sig), np.diag(g.Sigma)) < 1e-8


@pytest.mark.parametrize("likname", ["logistic", "logisticsoftmax"])
def test_phase_split_engine_matches_oracle(env, likname):
    """The multi-GPU phase-split ABI (step_local / lsm_* / step_stats / step_global, stats + gsum buffers exposed
    zero-copy to torch) driven by parallel.latent_parallel_step / batch_parallel_step on one rank."""
    AGP, R = env["AGP"], env["R"]
    from agp_amd import parallel as P

    rng = np.random.default_rng(21)
    B, iters = 64, 4
    X, y, ma, mr = _models(env, likname, rng, True, B)
    idx = [rng.choice(len(X), B, replace=False) for _ in range(iters)]
    eng = P.train_parallel(ma, X, y, iters, idx, mode="latent")
    # the exposed buffers really alias device memory of the library
    assert eng.stats.is_cuda and eng.stats.numel() == ma.n_latent * (64 + 64 * 64)
    mr.train(X, y, iters, idx_stream=idx)
    for k in range(ma.n_latent):
        mu, Sig, e1, e2 = ma.get_state(k)
        assert _rel(e2, mr.latents[k].eta2) < 1e-9 and _rel(mu, mr.latents[k].mu) < 1e-8
    # batch mode on one rank == the same thing
    X2, y2, mb, mr2 = _models(env, likname, np.random.default_rng(21), True, B)
    P.train_parallel(mb, X2, y2, iters, idx, mode="batch")
    for k in range(mb.n_latent):
        assert _rel(mb.get_state(k)[3], mr.latents[k].eta2) < 1e-9


def test_tied_z_hyper_step_engine_matches_oracle(env):
    """tied-Z mode (parallel.tied_hyper_step): 4-class LogisticSoftMax, the hyper-gradient summed over the latents and the
    same ADAM step applied to each -- HipEngine against the oracle engine of the gloo test, one rank."""
    AGP, R = env["AGP"], env["R"]
    import ctypes as C

    from agp_amd import parallel as P
    from test_parallel_gloo import TiedOracleEngine, _data

    X, y, lik, Z, idx, N, B, iters = _data("logisticsoftmax")
    ma = AGP.SVGP(1.0 * (AGP.SqExponentialKernel() @ AGP.ScaleTransform(3.0)), AGP.LogisticSoftMaxLikelihood(4),
                  AGP.AnalyticSVI(B), Z, optimiser=False)
    eng = P.HipEngine(ma, B).bind_data(X, y)
    ma._chk(eng.L.agp_svgp_hyper_configure(eng.h, 1, 0.01, 1, 0.001, 0.9, 0.999, 1e-8))
    ref = TiedOracleEngine(R.Kernel("sqexponential", 3.0, 1.0), lik, Z, X, R.treat_labels(y, lik), 0, 4, batchsize=B)
    for it in range(iters):
        P.latent_parallel_step(eng, idx[it], N / B)
        P.tied_hyper_step(eng)
        P.latent_parallel_step(ref, idx[it], N / B)
        P.tied_hyper_step(ref)
    eng.check()
    ma.k_opt = ma.z_opt = AGP.ADAM()
    ma._pull_hypers()
    for k in range(4):
        g = ref.M.latents[k]
        assert _rel(ma.Zs[k], g.Z) < 1e-9 and np.array_equal(ma.Zs[k], ma.Zs[0])
        assert ma.kernels[k].variance == pytest.approx(g.kernel.sigma2, rel=1e-9)
        assert ma.kernels[k].transform.s == pytest.approx(g.kernel.scale, rel=1e-9)
        assert _rel(ma.get_state(k)[2], g.eta1) < 1e-7
    assert _rel(ma.Zs[0], Z) > 1e-5


@pytest.mark.parametrize("stochastic", [False, True])
@pytest.mark.parametrize("aopt", [False, True])
def test_multioutput_svgp_matches_oracle(env, stochastic, aopt):
    """MOSVGP (src/models/MOSVGP.jl): Q = 3 latents mixed into 3 tasks (Gaussian, Logistic, StudentT) -- deliberately with
    Q != n_task would break the reference (Appendix A Q7); also run 4 latents / 2 tasks below."""
    AGP, R = env["AGP"], env["R"]
    for Q, liks_a, liks_r in [
        (3, [AGP.GaussianLikelihood(0.05), AGP.LogisticLikelihood(), AGP.StudentTLikelihood(3.0)],
         [R.GaussianLikelihood(0.05), R.LogisticLikelihood(), R.StudentTLikelihood(3.0)]),
        (4, [AGP.LogisticLikelihood(), AGP.GaussianLikelihood(0.1)], [R.LogisticLikelihood(), R.GaussianLikelihood(0.1)]),
        (2, [AGP.LaplaceLikelihood(0.3), AGP.BayesianSVM(), AGP.NegBinomialLikelihood(5.0)],
         [R.LaplaceLikelihood(0.3), R.BayesianSVM(), R.NegBinomialLikelihood(5.0)]),
    ]:
        rng = np.random.default_rng(31 + Q)
        N, D, m, B, iters = 240, 2, 12, 60, 5
        X = rng.random((N, D))
        f = [np.sin(4 * X[:, 0]), X[:, 1] - 0.5, np.cos(3 * X[:, 0] * X[:, 1])]
        ys_all = {"gaussian": f[0] + 0.1 * rng.standard_normal(N), "logistic": np.sign(f[1] + 0.1 * rng.standard_normal(N)),
                  "studentt": f[2] + 0.1 * rng.standard_t(3, N), "laplace": f[0] + rng.laplace(0, 0.3, N),
                  "bayesiansvm": np.sign(f[1] + 0.1 * rng.standard_normal(N)),
                  "negbinomial": rng.negative_binomial(5, 1.0 / (1.0 + np.exp(f[2]))).astype(np.int64)}
        ys = [ys_all[l.name] for l in liks_r]
        T = len(ys)
        A = rng.standard_normal((T, Q))
        A /= np.linalg.norm(A, axis=1, keepdims=True)
        Zs = [X[rng.permutation(N)[:m]].copy() for _ in range(Q)]
        idx = [rng.choice(N, B, replace=False) for _ in range(iters)]
        inf = AGP.AnalyticSVI(B) if stochastic else AGP.AnalyticVI()
        ma = AGP.MOSVGP(AGP.SqExponentialKernel() @ AGP.ScaleTransform(3.0), liks_a, inf, Zs, A=A.copy(),
                        Aoptimiser=AGP.ADAM(0.01) if aopt else False)
        mr = R.MOSVGP(R.Kernel("sqexponential", 3.0, 1.0), liks_r, Zs, A.copy(), stochastic=stochastic, batchsize=B,
                      A_opt=R.Adam(0.01) if aopt else None)
        ea, er = [], []
        AGP.train_(ma, X, ys, iters, idx_stream=idx, callback=lambda mdl, s, i: ea.append(AGP.objective(mdl, s)))
        mr.train(X, ys, iters, idx_stream=idx, callback=lambda M, it, xb, yb: er.append(M.elbo(yb)))
        assert np.allclose(ea, er, rtol=1e-8, atol=1e-7), (ea, er)
        assert _rel(ma.get_A(), mr.A) < 1e-9
        for q in range(Q):
            mu, Sig, e1, e2 = ma.get_state(q)
            assert _rel(e2, mr.latents[q].eta2) < 1e-9 and _rel(mu, mr.latents[q].mu) < 1e-8
        Xt = rng.random((50, D))
        mf, vf = AGP.predict_f(ma, Xt, cov=True)
        mfr, vfr = mr.predict_f(Xt, cov=True)
        for t in range(T):
            assert _rel(mf[t], mfr[t]) < 1e-8 and _rel(vf[t], vfr[t]) < 1e-7
        pa, pr = AGP.proba_y(ma, Xt), mr.proba_y(Xt)
        ya, yr = AGP.predict_y(ma, Xt), mr.predict_y(Xt)
        for t in range(T):
            assert _rel(pa[t][0], pr[t][0]) < 1e-8 and _rel(pa[t][1], pr[t][1]) < 1e-6
            assert np.allclose(np.asarray(ya[t], float), np.asarray(yr[t], float), atol=1e-8)


@pytest.mark.parametrize("likname,kname,ard", [("logistic", "sq", False), ("studentt", "m52", True),
                                               ("gaussian", "m32", False), ("logisticsoftmax", "sq", True),
                                               ("laplace", "sq", True), ("bayesiansvm", "m52", False),
                                               ("poisson", "sq", False), ("negbinomial", "m32", True),
                                               ("heteroscedastic", "sq", False)])
def test_hypergrad_matches_oracle(env, likname, kname, ard):
    """Hand-derived reverse mode (agp_hyper.h) vs the oracle's analytic gradient (itself pinned by finite differences in
    tests/test_oracle_kat.py) of the objective the reference hands to Zygote (autotuning.jl:96-98)."""
    AGP, R = env["AGP"], env["R"]
    rng = np.random.default_rng(41)
    N, D, m, B, iters = 200, 3, 14, 70, 3
    X = rng.random((N, D))
    f = np.sin(4 * X[:, 0]) + X[:, 1] * X[:, 2]
    if likname in NEW_LIKS:
        from _liks import agp_lik, labels, oracle_lik

        la, lr, y = agp_lik(AGP, likname), oracle_lik(R, likname), labels(likname, f, X, rng)
    else:
        la, lr, y = {
            "gaussian": (AGP.GaussianLikelihood(0.05), R.GaussianLikelihood(0.05), f + 0.1 * rng.standard_normal(N)),
            "logistic": (AGP.LogisticLikelihood(), R.LogisticLikelihood(), (f > f.mean()).astype(int)),
            "studentt": (AGP.StudentTLikelihood(3.0), R.StudentTLikelihood(3.0), f + 0.1 * rng.standard_t(3, N)),
            "logisticsoftmax": (AGP.LogisticSoftMaxLikelihood(3), R.LogisticSoftMaxLikelihood(3),
                                1 + np.digitize(f, np.quantile(f, [0.33, 0.66]))),
        }[likname]
    kcls, kn = {"sq": (AGP.SqExponentialKernel, "sqexponential"), "m52": (AGP.Matern52Kernel, "matern52"),
                "m32": (AGP.Matern32Kernel, "matern32")}[kname]
    sc = np.array([2.0, 3.0, 1.5]) if ard else 2.5
    ka = 1.3 * (kcls() @ (AGP.ARDTransform(sc) if ard else AGP.ScaleTransform(sc)))
    Z = X[rng.permutation(N)[:m]].copy()
    idx = [rng.choice(N, B, replace=False) for _ in range(iters)]
    ma = AGP.SVGP(ka, la, AGP.AnalyticSVI(B), Z, optimiser=False)
    AGP.train_(ma, X, y, iters, idx_stream=idx)
    mr = R.SVGP(R.Kernel(kn, sc, 1.3), lr, Z, stochastic=True, batchsize=B)
    yt = R.treat_labels(y, lr)
    mr.train(X, yt, iters, idx_stream=idx, labels_treated=True)
    xb, yb = X[idx[-1]], yt[idx[-1]]
    mr.hp_updated = True
    mr.compute_kernel_matrices(xb)
    for k in range(ma.n_latent):
        g = R.hyper_gradient(mr, xb, yb, k, N / B)
        dv, ds, dz = ma.hypergrad(k)
        assert abs(dv - g["dvariance"]) < 1e-8 * max(1.0, abs(g["dvariance"]))
        assert _rel(ds, g["dscale"]) < 1e-8
        assert _rel(dz, g["dZ"]) < 1e-8


@pytest.mark.parametrize("likname,ard,zopt", [("logistic", False, True), ("gaussian", True, False),
                                              ("logisticsoftmax", False, True)])
def test_training_with_hyper_steps_matches_oracle(env, likname, ard, zopt):
    """train! with optimiser / Zoptimiser (update_hyperparameters!, training.jl:65-69) against the oracle's training loop."""
    AGP, R = env["AGP"], env["R"]
    rng = np.random.default_rng(43)
    N, D, m, B, iters = 180, 2, 10, 60, 9
    X = rng.random((N, D))
    f = np.sin(5 * X[:, 0]) - X[:, 1]
    la, lr, y = {
        "gaussian": (AGP.GaussianLikelihood(0.05), R.GaussianLikelihood(0.05), f + 0.1 * rng.standard_normal(N)),
        "logistic": (AGP.LogisticLikelihood(), R.LogisticLikelihood(), (f > f.mean()).astype(int)),
        "logisticsoftmax": (AGP.LogisticSoftMaxLikelihood(3), R.LogisticSoftMaxLikelihood(3),
                            1 + np.digitize(f, np.quantile(f, [0.33, 0.66]))),
    }[likname]
    sc = np.array([3.0, 2.0]) if ard else 3.0
    ka = 1.2 * (AGP.SqExponentialKernel() @ (AGP.ARDTransform(sc) if ard else AGP.ScaleTransform(sc)))
    Z = X[rng.permutation(N)[:m]].copy()
    idx = [rng.choice(N, B, replace=False) for _ in range(iters)]
    ma = AGP.SVGP(ka, la, AGP.AnalyticSVI(B), Z, optimiser=AGP.ADAM(0.05), Zoptimiser=AGP.ADAM(0.01) if zopt else False)
    AGP.train_(ma, X, y, iters, idx_stream=idx)
    mr = R.SVGP(R.Kernel("sqexponential", sc, 1.2), lr, Z, stochastic=True, batchsize=B, k_opt=R.Adam(0.05),
                z_opt=R.Adam(0.01) if zopt else None, ard=ard)
    mr.train(X, y, iters, idx_stream=idx)
    for k in range(ma.n_latent):
        kr = mr.latents[k].kernel
        assert ma.kernels[k].variance == pytest.approx(kr.sigma2, rel=1e-8)
        assert _rel(ma.kernels[k].scales(D), np.broadcast_to(kr.scale, (D,))) < 1e-8
        assert _rel(ma.Zs[k], mr.latents[k].Z) < 1e-8
        assert abs(ma.kernels[k].variance - 1.2) > 1e-3  # the hypers really moved
        mu, Sig, e1, e2 = ma.get_state(k)
        assert _rel(e2, mr.latents[k].eta2) < 1e-7 and _rel(mu, mr.latents[k].mu) < 1e-7


@pytest.mark.parametrize("ard,zopt", [(False, True), (True, False)])
def test_multioutput_hyper_steps_match_oracle(env, ard, zopt):
    """MOSVGP(...; optimiser=ADAM, Zoptimiser=ADAM): gradient of every latent through the A-mixed data term, then the ADAM
    trajectory (kernel parameters in log space, Z directly) against the oracle (FD-pinned in tests/test_oracle_kat.py)."""
    AGP, R = env["AGP"], env["R"]
    rng = np.random.default_rng(77)
    N, D, m, B, Q, iters = 240, 2, 10, 60, 3, 7
    X = rng.random((N, D))
    f = [np.sin(4 * X[:, 0]), X[:, 1] - 0.5]
    ys = [f[0] + 0.1 * rng.standard_normal(N), np.sign(f[1] + 0.1 * rng.standard_normal(N))]
    A = rng.standard_normal((2, Q))
    A /= np.linalg.norm(A, axis=1, keepdims=True)
    Zs = [X[rng.permutation(N)[:m]].copy() for _ in range(Q)]
    idx = [rng.choice(N, B, replace=False) for _ in range(iters)]
    sc = np.array([2.0, 3.0]) if ard else 2.5
    ka = 1.3 * (AGP.SqExponentialKernel() @ (AGP.ARDTransform(sc) if ard else AGP.ScaleTransform(sc)))
    ma = AGP.MOSVGP(ka, [AGP.GaussianLikelihood(0.05), AGP.LogisticLikelihood()], AGP.AnalyticSVI(B), Zs, A=A.copy(),
                    Aoptimiser=False, optimiser=AGP.ADAM(0.01), Zoptimiser=AGP.ADAM(0.001) if zopt else False)
    mr = R.MOSVGP(R.Kernel("sqexponential", sc, 1.3), [R.GaussianLikelihood(0.05), R.LogisticLikelihood()], Zs, A.copy(),
                  stochastic=True, batchsize=B, k_opt=R.Adam(0.01), z_opt=R.Adam(0.001) if zopt else None, ard=ard)
    AGP.train_(ma, X, ys, iters, idx_stream=idx)
    mr.train(X, ys, iters, idx_stream=idx)
    for q in range(Q):
        kq = ma.kernels[q]
        assert kq.variance == pytest.approx(mr.latents[q].kernel.sigma2, rel=1e-8)
        got = np.asarray(kq.transform.s if hasattr(kq.transform, "s") else kq.transform.v, dtype=float)
        assert _rel(got, np.asarray(mr.latents[q].kernel.scale, dtype=float)) < 1e-8
        assert _rel(ma.Zs[q], mr.latents[q].Z) < 1e-8
        mu, Sig, e1, e2 = ma.get_state(q)
        assert _rel(e2, mr.latents[q].eta2) < 1e-7 and _rel(mu, mr.latents[q].mu) < 1e-7


@pytest.mark.parametrize("likname", ["logistic", "poisson", "logisticsoftmax"])
def test_save_and_load_trained_model_round_trip(env, likname, tmp_path):
    """save_trained_model / load_trained_model (docs/src/userguide.md:207-215; checkpoint / resume of SURVEY section 5):
    the reloaded model predicts identically AND continues training on the same trajectory as the uninterrupted one."""
    AGP = env["AGP"]
    rng = np.random.default_rng(12)
    B = 64
    X, y, ma, _ = _models(env, likname, rng, True, B)
    _, _, mb, _ = _models(env, likname, np.random.default_rng(12), True, B)
    idx = [rng.choice(len(X), B, replace=False) for _ in range(8)]
    AGP.train_(ma, X, y, 8, idx_stream=idx)              # uninterrupted
    AGP.train_(mb, X, y, 4, idx_stream=idx[:4])          # interrupted after 4 ...
    path = str(tmp_path / "model.npz")
    AGP.save_trained_model(path, mb)
    mc = AGP.load_trained_model(path)
    Xt = rng.random((40, X.shape[1]))
    pb, pc = AGP.predict_f(mb, Xt, cov=True), AGP.predict_f(mc, Xt, cov=True)
    for a, b in zip(np.atleast_2d(pb[0]), np.atleast_2d(pc[0])):
        assert _rel(b, a) < 1e-10
    AGP.train_(mc, X, y, 4, idx_stream=idx[4:], state=True)  # ... resumed from the file
    for l in range(ma.n_latent):
        mu_a, Sig_a, e1a, e2a = ma.get_state(l)
        mu_c, Sig_c, e1c, e2c = mc.get_state(l)
        assert _rel(e2c, e2a) < 1e-8 and _rel(mu_c, mu_a) < 1e-7
    if hasattr(ma.likelihood, "lam"):
        assert mc.likelihood.lam == pytest.approx(ma.likelihood.lam, rel=1e-8)


def test_hypergrad_plus_apply_equals_hyper_step(env):
    """agp_svgp_hypergrad + agp_svgp_hyper_apply (the split a tied-Z multi-GPU driver all-reduces in between) takes exactly
    the step agp_svgp_hyper_step takes."""
    import ctypes as C

    AGP, capi = env["AGP"], env["capi"]
    torch = env["torch"]
    L = capi.lib()
    B = 64
    out = []
    for split in (False, True):
        X, y, m, _ = _models(env, "logistic", np.random.default_rng(5), True, B)
        idx = [np.random.default_rng(9).choice(len(X), B, replace=False) for _ in range(3)]
        AGP.train_(m, X, y, 1, idx_stream=idx[:1])
        m.k_opt, m.z_opt = AGP.ADAM(0.01), AGP.ADAM(0.001)  # after the step: only this test's explicit hyper step runs
        h = m._h
        m._chk(L.agp_svgp_hyper_configure(h, 1, 0.01, 1, 0.001, 0.9, 0.999, 1e-8))
        if split:
            dz = torch.empty(m.m, m.D, dtype=m.tdtype, device="cuda")
            dv, ds = C.c_double(), (C.c_double * m.D)()
            m._chk(L.agp_svgp_hypergrad(h, 0, C.byref(dv), ds, C.c_void_p(dz.data_ptr())))
            m._chk(L.agp_svgp_hyper_apply(h, 0, C.byref(dv), ds, C.c_void_p(dz.data_ptr())))
        else:
            m._chk(L.agp_svgp_hyper_step(h))
        m._pull_hypers()
        out.append((m.kernels[0].variance, m.kernels[0].transform.s, m.Zs[0].copy()))
    assert out[0][0] == out[1][0] and out[0][1] == out[1][1] and np.array_equal(out[0][2], out[1][2])
