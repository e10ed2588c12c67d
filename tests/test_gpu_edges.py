"""Edge cases at the boundary (SURVEY.md 8b error convention; the reference's own argument checks):

* train!: iterations <= 0 (training.jl:23), minibatch larger than the data or non-positive (training.jl:27-29);
* the C ABI: B = 0 / B > max_batch -> AGP_ERR_BAD_BATCH with the reference's wording, null pointers -> AGP_ERR_INVALID;
* ragged sizes: one inducing point, one data point per minibatch, 1 / 63 / 65 test points, m and B off the 64-grid;
* empty prediction input;
* a Cholesky that fails reports the LAPACK-style pivot index (latentgp.jl:206 PosDefException.info), also beyond the task graph.
"""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mods(built):
    import torch

    assert torch.cuda.is_available()
    import agp_amd as AGP
    from agp_amd import capi

    from oracle import agp_ref as R

    return AGP, R, capi, torch


def _toy(rng, N=300, D=3):
    X = rng.random((N, D))
    y = np.sin(3 * X[:, 0]) + 0.1 * rng.standard_normal(N)
    return X, y


def test_train_argument_checks(mods):
    AGP, R, capi, torch = mods
    rng = np.random.default_rng(0)
    X, y = _toy(rng)
    Z = X[:10].copy()
    model = AGP.SVGP(AGP.SqExponentialKernel(), AGP.GaussianLikelihood(0.1), AGP.AnalyticSVI(50), Z, optimiser=False)
    with pytest.raises(ValueError, match="Number of iterations should be positive"):
        AGP.train_(model, X, y, 0)
    big = AGP.SVGP(AGP.SqExponentialKernel(), AGP.GaussianLikelihood(0.1), AGP.AnalyticSVI(len(X) + 1), Z, optimiser=False)
    with pytest.raises(ValueError, match="size of mini-batch"):
        AGP.train_(big, X, y, 2)


def test_abi_batch_and_pointer_checks(mods):
    AGP, R, capi, torch = mods
    rng = np.random.default_rng(1)
    X, y = _toy(rng)
    Z = X[:10].copy()
    model = AGP.SVGP(AGP.SqExponentialKernel(), AGP.GaussianLikelihood(0.1), AGP.AnalyticSVI(50), Z, optimiser=False)
    AGP.train_(model, X, y, 2)
    L, h = capi.lib(), model._h
    xd = torch.tensor(X, dtype=torch.float64, device="cuda")
    yd = torch.tensor(y, dtype=torch.float64, device="cuda")
    idx = torch.arange(50, dtype=torch.int64, device="cuda")
    args = (C.c_void_p(xd.data_ptr()), xd.stride(0), C.c_void_p(yd.data_ptr()), C.c_void_p(idx.data_ptr()))
    for bad in (0, -3, 51):  # the handle was created for minibatches of at most 50
        st = L.agp_svgp_cavi_step(h, *args, bad, 6.0)
        assert st == 4
        assert b"size of mini-batch" in L.agp_last_error(model._ctx)
    assert L.agp_svgp_cavi_step(h, None, xd.stride(0), C.c_void_p(yd.data_ptr()), C.c_void_p(idx.data_ptr()), 50, 6.0) == 1
    assert L.agp_svgp_cavi_step(h, C.c_void_p(xd.data_ptr()), 2, C.c_void_p(yd.data_ptr()), C.c_void_p(idx.data_ptr()), 50, 6.0) == 1  # ldx < D
    # the handle is still usable
    assert L.agp_svgp_cavi_step(h, *args, 50, 6.0) == 0
    assert L.agp_svgp_check_status(h) == 0


@pytest.mark.parametrize("m,B,N", [(1, 1, 40), (3, 7, 100), (65, 63, 200), (130, 129, 400)])
def test_ragged_sizes_match_oracle(mods, m, B, N):
    """Sizes off the 64-grid (the library pads to tiles internally): eta, predictions against the oracle on the same indices."""
    AGP, R, capi, torch = mods
    rng = np.random.default_rng(100 + m)
    X, y = _toy(rng, N, 2)
    Z = X[rng.permutation(N)[:m]].copy()
    idx = [rng.choice(N, B, replace=False) for _ in range(6)]
    ma = AGP.SVGP(AGP.with_lengthscale(AGP.SqExponentialKernel(), 0.5), AGP.GaussianLikelihood(0.05), AGP.AnalyticSVI(B), Z,
                  optimiser=False)
    mr = R.SVGP(R.Kernel("sqexponential", 2.0, 1.0), R.GaussianLikelihood(0.05), Z, stochastic=True, batchsize=B)
    AGP.train_(ma, X, y, 6, idx_stream=idx)
    mr.train(X, y, 6, idx_stream=idx)
    mu, Sig, e1, e2 = ma.get_state()
    # (K_ZZ of 130 points at lengthscale 0.5 in the unit square is ill-conditioned: the per-column launches -- AGP_CHOL_DAG=0 -- sum
    # K^-1 in another order and land 1.6e-9 from the oracle where the task graph lands 3e-10)
    assert np.max(np.abs(e1 - mr.latents[0].eta1)) <= 5e-9 * max(1.0, np.max(np.abs(mr.latents[0].eta1)))
    assert np.max(np.abs(e2 - mr.latents[0].eta2)) <= 5e-9 * max(1.0, np.max(np.abs(mr.latents[0].eta2)))
    for nt in (1, 63, 65):
        Xt = rng.random((nt, 2))
        pm, pv = AGP.predict_f(ma, Xt, cov=True)
        rm, rv = mr.predict_f(Xt, cov=True)
        assert pm.shape == (nt,) and pv.shape == (nt,)
        assert np.max(np.abs(pm - rm)) <= 1e-8 * max(1.0, np.max(np.abs(rm)))
        assert np.max(np.abs(pv - rv)) <= 1e-8 * max(1.0, np.max(np.abs(rv)))


def test_empty_prediction_input(mods):
    AGP, R, capi, torch = mods
    rng = np.random.default_rng(2)
    X, y = _toy(rng)
    model = AGP.SVGP(AGP.SqExponentialKernel(), AGP.GaussianLikelihood(0.1), AGP.AnalyticSVI(50), X[:10].copy(), optimiser=False)
    AGP.train_(model, X, y, 2)
    pm, pv = AGP.predict_f(model, np.zeros((0, X.shape[1])), cov=True)
    assert pm.shape == (0,) and pv.shape == (0,)
    assert AGP.predict_y(model, np.zeros((0, X.shape[1]))).shape == (0,)


@pytest.mark.parametrize("n,bad", [(100, 37), (2112, 2000), (6144, 4100)])  # task graph / per-column launches / blocked
def test_failed_cholesky_reports_the_pivot(mods, n, bad):
    """A matrix whose leading minor of order `bad` + 1 is the first non-positive one: info = bad + 1 (1-based, LAPACK potrf)."""
    AGP, R, capi, torch = mods
    L = capi.lib()
    ctx = C.c_void_p()
    assert L.agp_ctx_create(0, C.c_void_p(torch.cuda.current_stream().cuda_stream), C.byref(ctx)) == 0
    try:
        g = torch.Generator(device="cuda").manual_seed(n)
        G = torch.randn(n, n + 8, dtype=torch.float64, device="cuda", generator=g)
        A = G @ G.T / n + 0.5 * torch.eye(n, dtype=torch.float64, device="cuda")
        A[bad, bad] = -1.0  # the Schur complement at `bad` is at most A[bad, bad] < 0
        info = C.c_int32(-7)
        st = L.agp_potrf_jitter(ctx, 0, C.c_void_p(A.data_ptr()), n, n, 0.0, C.byref(info))
        assert st == 2 and info.value == bad + 1, (st, info.value, L.agp_last_error(ctx))
    finally:
        L.agp_ctx_destroy(ctx)


@pytest.mark.parametrize("n", [300, 2112, 6144])  # task graph / per-column launches / blocked, in fp32
def test_fp32_factorisation_paths(mods, n):
    AGP, R, capi, torch = mods
    L = capi.lib()
    ctx = C.c_void_p()
    assert L.agp_ctx_create(0, C.c_void_p(torch.cuda.current_stream().cuda_stream), C.byref(ctx)) == 0
    try:
        g = torch.Generator(device="cuda").manual_seed(n)
        G = torch.randn(n, n + 8, dtype=torch.float64, device="cuda", generator=g)
        A64 = G @ G.T / n + 0.5 * torch.eye(n, dtype=torch.float64, device="cuda")
        A = A64.to(torch.float32).contiguous()
        info = C.c_int32(-7)
        assert L.agp_potrf_jitter(ctx, 1, C.c_void_p(A.data_ptr()), n, n, 1e-3, C.byref(info)) == 0 and info.value == 0
        Lf = torch.tril(A).to(torch.float64)
        ref = A64 + 1e-3 * torch.eye(n, dtype=torch.float64, device="cuda")
        err = (Lf @ Lf.T - ref).abs().max().item() / ref.abs().max().item()
        assert err < 2e-5, err
    finally:
        L.agp_ctx_destroy(ctx)


def test_fp32_hyper_gradient_and_multiclass(mods):
    """T = Float32 beyond the plain step: the hand-derived hyper-gradient and a three-class LogisticSoftMax trajectory against the
    fp64 oracle (tolerances of single precision: gradients 2e-2 relative to their largest entry, predictive means 5e-3)."""
    AGP, R, capi, torch = mods
    rng = np.random.default_rng(77)
    N, D, m, B, iters = 240, 3, 16, 80, 4
    X = rng.random((N, D))
    f = np.sin(4 * X[:, 0]) + X[:, 1] * X[:, 2]
    idx = [rng.choice(N, B, replace=False) for _ in range(iters)]
    Z = X[rng.permutation(N)[:m]].copy()
    # hyper-gradient, logistic
    y = (f > f.mean()).astype(int)
    ma = AGP.SVGP(1.3 * (AGP.SqExponentialKernel() @ AGP.ScaleTransform(2.5)), AGP.LogisticLikelihood(), AGP.AnalyticSVI(B), Z,
                  optimiser=False, T=np.float32)
    AGP.train_(ma, X, y, iters, idx_stream=idx)
    mr = R.SVGP(R.Kernel("sqexponential", 2.5, 1.3), R.LogisticLikelihood(), Z, stochastic=True, batchsize=B, jitter=1e-3)
    yt = R.treat_labels(y, mr.likelihood)
    mr.train(X, yt, iters, idx_stream=idx, labels_treated=True)
    mr.hp_updated = True
    mr.compute_kernel_matrices(X[idx[-1]])
    g = R.hyper_gradient(mr, X[idx[-1]], yt[idx[-1]], 0, N / B)
    dv, ds, dz = ma.hypergrad(0)
    assert abs(dv - g["dvariance"]) < 2e-2 * max(1.0, abs(g["dvariance"]))
    assert np.max(np.abs(np.asarray(ds) - g["dscale"])) < 2e-2 * max(1.0, np.max(np.abs(g["dscale"])))
    assert np.max(np.abs(np.asarray(dz) - g["dZ"])) < 2e-2 * max(1.0, np.max(np.abs(g["dZ"])))
    # three classes
    y3 = 1 + np.digitize(f, np.quantile(f, [0.33, 0.66]))
    m3 = AGP.SVGP(1.5 * (AGP.SqExponentialKernel() @ AGP.ScaleTransform(2.0)), AGP.LogisticSoftMaxLikelihood(3), AGP.AnalyticSVI(B),
                  Z, optimiser=False, T=np.float32)
    r3 = R.SVGP(R.Kernel("sqexponential", 2.0, 1.5), R.LogisticSoftMaxLikelihood(3), Z, stochastic=True, batchsize=B, jitter=1e-3)
    AGP.train_(m3, X, y3, iters, idx_stream=idx)
    r3.train(X, y3, iters, idx_stream=idx)
    Xt = rng.random((60, D))
    pa, pr = AGP.predict_f(m3, Xt), r3.predict_f(Xt)
    for k in range(3):
        assert np.max(np.abs(np.asarray(pa[k]) - pr[k])) < 5e-3 * max(1.0, np.max(np.abs(pr[k])))
    assert np.mean(AGP.predict_y(m3, Xt) == r3.predict_y(Xt)) > 0.95
