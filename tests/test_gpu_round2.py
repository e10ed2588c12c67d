"""GPU tests of what round 2 added or closed:

* BASELINE.json configs[0] (C1) at its exact workload against the oracle; an oracle-checked multi-output model at m = 512, Q != T;
* error paths AGP_ERR_NEG_KTILDE / AGP_ERR_LABELS / get_matrix capacity;
* predict_f(...; cov=true, diag=false)  (predictions.jl:45-49);
* reference_compat_stale_K (training.jl:187-208) against the oracle's switch;
* structural hyper step: only kernel parameters that exist in the object are stepped (autotuning.jl:99-118);
* init_state semantics of train! without a state (training.jl:41-45); the data-invalidation contract of the kappa cache;
* the K = 8 multi-latent soak (bitwise), shortened from tools/soak_multilatent.py.
"""
import ctypes as C
import os

import numpy as np
import pytest

import _knobs as K_

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


@pytest.fixture(scope="module")
def mods(built):
    import torch

    assert torch.cuda.is_available()
    import agp_amd as AGP
    from agp_amd import capi

    from oracle import agp_ref as R

    return AGP, R, capi, torch


def _rff_targets(rng, X, ell, R=64):
    D = X.shape[1]
    om = rng.standard_normal((D, R)) / ell
    b = rng.random(R) * 2 * np.pi
    w = rng.standard_normal(R)
    return np.cos(X @ om + b) @ w * np.sqrt(2.0 / R)


def test_c1_exact_workload_matches_oracle(mods):
    """BASELINE.json configs[0]: SVGP SqExponential + Gaussian, AnalyticSVI(256), m = 64, N = 10k, D = 8, fp64 -- the
    reference's own CPU-runnable case, here through the HIP path against the oracle on the same index stream."""
    AGP, R, capi, torch = mods
    rng = np.random.default_rng(1234)
    N, D, m, B, iters = 10_000, 8, 64, 256, 20
    X = rng.random((N, D))
    ell = np.sqrt(D) / 4
    y = _rff_targets(rng, X, ell) + 0.1 * rng.standard_normal(N)
    Z = X[rng.permutation(N)[:m]].copy()
    idx = [rng.choice(N, B, replace=False) for _ in range(iters)]
    ma = AGP.SVGP(AGP.with_lengthscale(AGP.SqExponentialKernel(), ell), AGP.GaussianLikelihood(0.01), AGP.AnalyticSVI(B), Z,
                  optimiser=False)
    mr = R.SVGP(R.Kernel("sqexponential", 1.0 / ell, 1.0), R.GaussianLikelihood(0.01), Z, stochastic=True, batchsize=B)
    ea, er = [], []
    AGP.train_(ma, X, y, iters, idx_stream=idx, callback=lambda mdl, s, i: ea.append(AGP.objective(mdl, s)))
    mr.train(X, y, iters, idx_stream=idx, callback=lambda M, it, xb, yb: er.append(M.elbo(yb)))
    mu, Sig, e1, e2 = ma.get_state(0)
    g = mr.latents[0]
    assert _rel(e1, g.eta1) < 1e-9 and _rel(e2, g.eta2) < 1e-9
    assert _rel(mu, g.mu) < 1e-8 and _rel(Sig, g.Sigma) < 1e-8
    assert np.allclose(ea, er, rtol=1e-8)
    Xt = rng.random((500, D))
    pm, pv = AGP.predict_f(ma, Xt, cov=True)
    rm, rv = mr.predict_f(Xt, cov=True)
    assert _rel(pm, rm[0]) < 1e-8 and _rel(pv, rv[0]) < 1e-7
    assert _rel(AGP.predict_y(ma, Xt), mr.predict_y(Xt)) < 1e-8


def test_multioutput_m512_q_ne_t_matches_oracle(mods):
    """Multi-output SVGP with several block columns per latent (m = 512 -> 8 x 8 tiles, interleaved task graphs) and more
    latents than outputs (Q = 3, T = 2): eta, A and the per-iteration ELBO against the oracle -- not just properties."""
    AGP, R, capi, torch = mods
    rng = np.random.default_rng(8)
    N, D, m, B, Q, iters = 3000, 4, 512, 512, 3, 3
    X = rng.random((N, D))
    f = [np.sin(4 * X[:, 0]) + X[:, 1], X[:, 2] - X[:, 3]]
    ys = [f[0] + 0.1 * rng.standard_normal(N), np.sign(f[1] + 0.1 * rng.standard_normal(N))]
    A = rng.standard_normal((2, Q))
    A /= np.linalg.norm(A, axis=1, keepdims=True)
    Zs = [X[rng.permutation(N)[:m]].copy() for _ in range(Q)]
    idx = [rng.choice(N, B, replace=False) for _ in range(iters)]
    ka = 1.3 * (AGP.SqExponentialKernel() @ AGP.ScaleTransform(2.5))
    ma = AGP.MOSVGP(ka, [AGP.GaussianLikelihood(0.05), AGP.LogisticLikelihood()], AGP.AnalyticSVI(B), Zs, A=A.copy(),
                    Aoptimiser=AGP.ADAM(0.01), optimiser=False)
    mr = R.MOSVGP(R.Kernel("sqexponential", 2.5, 1.3), [R.GaussianLikelihood(0.05), R.LogisticLikelihood()], Zs, A.copy(),
                  stochastic=True, batchsize=B, A_opt=R.Adam(0.01))
    ea, er = [], []
    AGP.train_(ma, X, ys, iters, idx_stream=idx, callback=lambda mdl, s, i: ea.append(AGP.objective(mdl, s)))
    mr.train(X, ys, iters, idx_stream=idx, callback=lambda M, it, xb, yb: er.append(M.elbo(yb)))
    for q in range(Q):
        mu, Sig, e1, e2 = ma.get_state(q)
        assert _rel(e2, mr.latents[q].eta2) < 1e-8 and _rel(e1, mr.latents[q].eta1) < 1e-8
        assert _rel(mu, mr.latents[q].mu) < 1e-7
    assert _rel(ma.get_A(), mr.A) < 1e-9
    assert np.allclose(ea, er, rtol=1e-8)


def test_label_errors_mirror_treat_labels(mods):
    """treat_labels! ArgumentErrors (classification.jl:36-44, multiclass.jl:81-83) surface before anything reaches the device."""
    AGP, R, capi, torch = mods
    rng = np.random.default_rng(0)
    X = rng.random((50, 2))
    Z = X[:10].copy()
    m = AGP.SVGP(AGP.SqExponentialKernel(), AGP.LogisticLikelihood(), AGP.AnalyticVI(), Z, optimiser=False)
    with pytest.raises((ValueError, TypeError)):
        AGP.train_(m, X, rng.integers(0, 3, 50), 1)  # three classes for a Bernoulli likelihood
    mk = AGP.SVGP(AGP.SqExponentialKernel(), AGP.LogisticSoftMaxLikelihood(3), AGP.AnalyticVI(), Z, optimiser=False)
    with pytest.raises((ValueError, TypeError, RuntimeError)):
        AGP.train_(mk, X, rng.integers(0, 5, 50), 1)  # more classes than the likelihood was built for
    mk = AGP.SVGP(AGP.SqExponentialKernel(), AGP.LogisticSoftMaxLikelihood(3), AGP.AnalyticVI(), Z, optimiser=False)
    # the device-side guard: a class index outside [0, K) must not be accepted silently by the ABI either
    AGP.train_(mk, X, 1 + rng.integers(0, 3, 50), 1)
    L, h = capi.lib(), mk._h
    Xd, yd, _ = mk._data
    bad = torch.full((50,), 7, dtype=torch.int32, device="cuda")
    st = L.agp_svgp_cavi_step(h, C.c_void_p(Xd.data_ptr()), Xd.stride(0), C.c_void_p(bad.data_ptr()), None, 50, 1.0)
    st = st or L.agp_svgp_check_status(h)
    assert st == 6, capi.ERR_NAMES.get(st, st)  # AGP_ERR_LABELS


def test_get_matrix_capacity_and_last_batch(mods):
    """B-sized exports follow the LAST batch the handle saw (an ELBO on a larger set counts) and the library
    refuses a smaller buffer instead of overrunning it; alpha is state and is exported by capacity."""
    AGP, R, capi, torch = mods
    rng = np.random.default_rng(3)
    X = rng.random((400, 2))
    y = 1 + rng.integers(0, 3, 400)
    Z = X[:20].copy()
    B = 64
    m = AGP.SVGP(AGP.SqExponentialKernel(), AGP.LogisticSoftMaxLikelihood(3), AGP.AnalyticSVI(B), Z, optimiser=False)
    AGP.train_(m, X, y, 2)
    L, h = capi.lib(), m._h
    nb = C.c_int64()
    assert L.agp_svgp_last_batch(h, C.byref(nb)) == 0 and nb.value == B
    assert m.get_matrix(capi.VEC_THETA, 0).shape == (B,)
    AGP.ELBO(m, X, y, rho=1.0)  # the handle is re-created for 400 points and evaluates on them
    h = m._h
    assert L.agp_svgp_last_batch(h, C.byref(nb)) == 0 and nb.value == 400
    small = torch.empty(B, dtype=torch.float64, device="cuda")
    st = L.agp_svgp_get_matrix(h, 0, capi.VEC_THETA, C.c_void_p(small.data_ptr()), 1, B)
    assert st == 1 and b"capacity" in L.agp_last_error(m._ctx)  # AGP_ERR_INVALID, nothing written
    assert m.get_matrix(capi.VEC_THETA, 0).shape == (400,)
    alpha = m.get_matrix(capi.VEC_ALPHA, 0, B)  # state: any capacity up to max_batch
    assert alpha.shape == (B,) and np.all(alpha > 1.0)
    AGP.save_trained_model("/tmp/agp_capacity_test.npz", m)  # used to overrun a 64-element buffer after the ELBO call


def test_predict_f_full_covariance(mods):
    AGP, R, capi, torch = mods
    rng = np.random.default_rng(5)
    for likname, K in (("gaussian", 1), ("logisticsoftmax", 3)):
        X = rng.random((300, 3))
        f = np.sin(3 * X[:, 0]) + X[:, 1]
        Z = X[rng.permutation(300)[:70]].copy()
        if K == 1:
            y, la, lr = f + 0.1 * rng.standard_normal(300), AGP.GaussianLikelihood(0.01), R.GaussianLikelihood(0.01)
        else:
            y = 1 + np.digitize(f, np.quantile(f, [0.33, 0.66]))
            la, lr = AGP.LogisticSoftMaxLikelihood(3), R.LogisticSoftMaxLikelihood(3)
        ma = AGP.SVGP(1.5 * (AGP.Matern52Kernel() @ AGP.ScaleTransform(2.0)), la, AGP.AnalyticVI(), Z, optimiser=False)
        mr = R.SVGP(R.Kernel("matern52", 2.0, 1.5), lr, Z, stochastic=False)
        AGP.train_(ma, X, y, 3)
        mr.train(X, y, 3)
        Xt = rng.random((77, 3))  # ragged: not a multiple of the tile size
        mu, cov = AGP.predict_f(ma, Xt, cov=True, diag=False)
        rmu, rcov = mr.predict_f(Xt, cov=True, diag=False)
        _, var = AGP.predict_f(ma, Xt, cov=True)
        for k in range(K):
            mk, ck = (mu, cov) if K == 1 else (mu[k], cov[k])
            vk = var if K == 1 else var[k]
            assert ck.shape == (77, 77) and _rel(mk, rmu[k]) < 1e-8 and _rel(ck, rcov[k]) < 1e-8
            # (symmetric up to the rounding of k** - k* A k*': two GEMM results of O(1) subtracted, entries of O(0.1))
            assert _rel(np.diag(ck), vk) < 1e-9 and np.max(np.abs(ck - ck.T)) < 1e-11


@pytest.mark.parametrize("stochastic", [True, False])
def test_reference_compat_stale_K_matches_oracle_switch(mods, stochastic):
    """with reference_compat_stale_K the Cholesky of K_ZZ of the first iteration is kept across the
    hyper-parameter steps of one train! (training.jl:187-208), the gradient still sees fresh matrices (ELBO.jl:15-21), and
    train! ends with compute_Ks.  Same switch in the oracle; the default (refresh) must differ from it.  (Small learning rates:
    with the stale factor K~ = kdiag - diag(kappa Knm') drifts negative within a few larger steps -- see the next test.)"""
    AGP, R, capi, torch = mods
    rng = np.random.default_rng(43)
    N, D, m, B, iters = 180, 2, 10, 60, 9
    X = rng.random((N, D))
    f = np.sin(5 * X[:, 0]) - X[:, 1]
    y = (f > f.mean()).astype(int)
    Z = rng.random((m, D))
    idx = [rng.choice(N, B, replace=False) for _ in range(iters)]
    klr, zlr = (0.002, 0.0005) if stochastic else (0.02, 0.005)
    out = {}
    for compat in (True, False):
        ka = AGP.SqExponentialKernel() @ AGP.ScaleTransform(10.0 if stochastic else 6.0)
        inf = AGP.AnalyticSVI(B) if stochastic else AGP.AnalyticVI()
        ma = AGP.SVGP(ka, AGP.LogisticLikelihood(), inf, Z, optimiser=AGP.ADAM(klr), Zoptimiser=AGP.ADAM(zlr),
                      reference_compat_stale_K=compat)
        mr = R.SVGP(R.Kernel("sqexponential", 10.0 if stochastic else 6.0, 1.0, has_variance=False), R.LogisticLikelihood(), Z,
                    stochastic=stochastic, batchsize=B, k_opt=R.Adam(klr), z_opt=R.Adam(zlr), reference_compat_stale_K=compat)
        ea, er = [], []
        AGP.train_(ma, X, y, iters, idx_stream=idx, callback=lambda mdl, s, i: ea.append(AGP.objective(mdl, s)))
        mr.train(X, y, iters, idx_stream=idx, callback=lambda M, it, xb, yb: er.append(M.elbo(yb)))
        kr = mr.latents[0].kernel
        assert ma.kernels[0].variance == 1.0 and kr.sigma2 == 1.0  # not a ScaledKernel: never stepped
        assert ma.kernels[0].transform.s == pytest.approx(float(kr.scale), rel=1e-8)
        assert _rel(ma.Zs[0], mr.latents[0].Z) < 1e-8
        mu, Sig, e1, e2 = ma.get_state(0)
        assert _rel(e2, mr.latents[0].eta2) < 1e-7 and _rel(mu, mr.latents[0].mu) < 1e-7
        assert np.allclose(ea, er, rtol=1e-7), (ea, er)
        # after train! the kernel matrices are the fresh ones again (compute_Ks): predictions agree with the oracle's
        Xt = rng.random((30, D))
        assert _rel(AGP.predict_f(ma, Xt), mr.predict_f(Xt)[0]) < 1e-7
        out[compat] = e2
    assert _rel(out[True], out[False]) > 1e-3  # the flag really changes the trajectory


def test_stale_K_runs_into_negative_ktilde_like_the_reference(mods):
    """What the quirk does at ordinary learning rates: K~ goes negative a few hyper steps in and the reference throws
    error("K~ has negative values") (latentgp.jl:213).  Oracle and device agree on that too, and the failure is reported as
    AGP_ERR_NEG_KTILDE, not as the loss of positive-definiteness that follows from it."""
    AGP, R, capi, torch = mods
    rng = np.random.default_rng(43)
    N, D, m, B, iters = 180, 2, 10, 60, 9
    X = rng.random((N, D))
    f = np.sin(5 * X[:, 0]) - X[:, 1]
    y = (f > f.mean()).astype(int)
    Z = X[rng.permutation(N)[:m]].copy()
    idx = [rng.choice(N, B, replace=False) for _ in range(iters)]
    mr = R.SVGP(R.Kernel("sqexponential", 3.0, 1.2), R.LogisticLikelihood(), Z, stochastic=True, batchsize=B,
                k_opt=R.Adam(0.05), z_opt=R.Adam(0.01), reference_compat_stale_K=True)
    with pytest.raises(Exception, match="negative"):
        mr.train(X, y, iters, idx_stream=idx)
    ma = AGP.SVGP(1.2 * (AGP.SqExponentialKernel() @ AGP.ScaleTransform(3.0)), AGP.LogisticLikelihood(), AGP.AnalyticSVI(B), Z,
                  optimiser=AGP.ADAM(0.05), Zoptimiser=AGP.ADAM(0.01), reference_compat_stale_K=True)
    with pytest.raises(capi.AGPError) as ei:
        AGP.train_(ma, X, y, iters, idx_stream=idx)
    assert ei.value.status == 3 and "negative" in str(ei.value)  # AGP_ERR_NEG_KTILDE
    # the corrected default trains through
    mb = AGP.SVGP(1.2 * (AGP.SqExponentialKernel() @ AGP.ScaleTransform(3.0)), AGP.LogisticLikelihood(), AGP.AnalyticSVI(B), Z,
                  optimiser=AGP.ADAM(0.05), Zoptimiser=AGP.ADAM(0.01))
    AGP.train_(mb, X, y, iters, idx_stream=idx)


def test_hyper_step_is_structural(mods):
    """the reference's kernel gradient is a NamedTuple over the kernel OBJECT (autotuning.jl:99-118), so
    `SqExponentialKernel()` has nothing to step, `with_lengthscale(k, l)` only its scale, `sigma2 * k` also the variance."""
    AGP, R, capi, torch = mods
    rng = np.random.default_rng(11)
    N, D, m, B, iters = 200, 2, 12, 50, 8
    X = rng.random((N, D))
    y = np.sin(5 * X[:, 0]) - X[:, 1] + 0.1 * rng.standard_normal(N)
    Z = X[rng.permutation(N)[:m]].copy()
    idx = [rng.choice(N, B, replace=False) for _ in range(iters)]
    cases = [
        (AGP.SqExponentialKernel(), R.Kernel("sqexponential", 1.0, 1.0, has_variance=False, has_transform=False)),
        (AGP.with_lengthscale(AGP.SqExponentialKernel(), 0.4),
         R.Kernel("sqexponential", 2.5, 1.0, has_variance=False, has_transform=True)),
        (2.0 * AGP.SqExponentialKernel(), R.Kernel("sqexponential", 1.0, 2.0, has_variance=True, has_transform=False)),
    ]
    for ka, kr in cases:
        ma = AGP.SVGP(ka, AGP.GaussianLikelihood(0.05), AGP.AnalyticSVI(B), Z)  # default optimiser = ADAM(0.01), SVGP.jl:39
        mr = R.SVGP(kr, R.GaussianLikelihood(0.05), Z, stochastic=True, batchsize=B, k_opt=R.Adam(0.01))
        AGP.train_(ma, X, y, iters, idx_stream=idx)
        mr.train(X, y, iters, idx_stream=idx)
        k = ma.kernels[0]
        if not kr.has_variance:
            assert k.variance == 1.0 and not k.has_variance
        else:
            assert k.variance != 2.0 and k.variance == pytest.approx(mr.latents[0].kernel.sigma2, rel=1e-9)
        if not kr.has_transform:
            assert k.transform is None
        else:
            assert k.transform.s != 2.5 and k.transform.s == pytest.approx(float(mr.latents[0].kernel.scale), rel=1e-9)
        assert _rel(ma.get_state(0)[3], mr.latents[0].eta2) < 1e-8
    # a non-zero prior mean with the (default) optimiser: like the reference, the model is built and trains until its first hyper
    # step (n_iter >= 3, training.jl:65-69), where the reference's prior-mean update cannot run (constantmean.jl:31 vs its call)
    mm = AGP.SVGP(AGP.SqExponentialKernel(), AGP.GaussianLikelihood(0.05), AGP.AnalyticSVI(B), Z, mean=1.0)
    AGP.train_(mm, X, y, 3, idx_stream=idx)
    assert np.all(np.isfinite(AGP.predict_f(mm, X[:10])))
    with pytest.raises(NotImplementedError):
        AGP.train_(mm, X, y, 3, idx_stream=idx, state=True)


def test_train_without_state_restarts_the_state(mods):
    """train! without `state` runs init_state (training.jl:41-45): the RobbinsMonro counter restarts at 1 and the
    LogisticSoftMax alpha at K; passing the state continues both.  The posterior is the model's and is kept either way."""
    AGP, R, capi, torch = mods
    rng = np.random.default_rng(21)
    N, B = 300, 64
    X = rng.random((N, 3))
    f = np.sin(3 * X[:, 0]) + X[:, 1] ** 2 - 0.7
    y = 1 + np.digitize(f, np.quantile(f, [0.33, 0.66]))
    Z = X[rng.permutation(N)[:20]].copy()
    idx = [rng.choice(N, B, replace=False) for _ in range(6)]

    def make():
        return AGP.SVGP(1.5 * (AGP.SqExponentialKernel() @ AGP.ScaleTransform(2.0)), AGP.LogisticSoftMaxLikelihood(3),
                        AGP.AnalyticSVI(B), Z, optimiser=False)

    whole, cont, fresh = make(), make(), make()
    AGP.train_(whole, X, y, 6, idx_stream=idx)
    AGP.train_(cont, X, y, 3, idx_stream=idx[:3])
    AGP.train_(cont, X, y, 3, idx_stream=idx[3:], state=True)
    AGP.train_(fresh, X, y, 3, idx_stream=idx[:3])
    AGP.train_(fresh, X, y, 3, idx_stream=idx[3:])
    n = C.c_int64()
    for mdl, want in ((whole, 7), (cont, 7), (fresh, 4)):
        capi.lib().agp_svgp_get_opt_state(mdl._h, C.byref(n))
        assert n.value == want
    assert _rel(cont.get_state(0)[3], whole.get_state(0)[3]) < 1e-12
    assert _rel(fresh.get_state(0)[3], whole.get_state(0)[3]) > 1e-4  # larger steps again: a different trajectory
    # oracle with the same restart: counters and local variables as new, posterior kept
    mr = R.SVGP(R.Kernel("sqexponential", 2.0, 1.5), R.LogisticSoftMaxLikelihood(3), Z, stochastic=True, batchsize=B)
    mr.train(X, y, 3, idx_stream=idx[:3])
    mr.local_vars = None
    for g in mr.latents:
        g.n_eta1 = g.n_eta2 = 1
    mr.train(X, y, 3, idx_stream=idx[3:])
    for k in range(3):
        assert _rel(fresh.get_state(k)[3], mr.latents[k].eta2) < 1e-8


def test_refilled_buffer_needs_invalidate_data(mods):
    """The AnalyticVI kappa cache is keyed on pointers (documented in include/agp_hip.h): refilling X in place is only seen
    after agp_svgp_invalidate_data -- which train_ calls itself at the start of every call."""
    AGP, R, capi, torch = mods
    rng = np.random.default_rng(2)
    X1, X2 = rng.random((200, 2)), rng.random((200, 2))
    y = np.sin(4 * X1[:, 0])
    Z = X1[:15].copy()
    m = AGP.SVGP(AGP.SqExponentialKernel() @ AGP.ScaleTransform(2.0), AGP.GaussianLikelihood(0.1), AGP.AnalyticVI(), Z,
                 optimiser=False)
    buf = torch.tensor(X1, dtype=torch.float64, device="cuda")
    AGP.train_(m, buf, y, 1)
    L, h = capi.lib(), m._h
    Xd, yd, _ = m._data
    assert Xd.data_ptr() == buf.data_ptr()  # no copy was made: the library reads the caller's buffer
    buf.copy_(torch.tensor(X2, dtype=torch.float64, device="cuda"))
    args = (h, C.c_void_p(Xd.data_ptr()), Xd.stride(0), C.c_void_p(yd.data_ptr()), None, 200, 1.0)
    assert L.agp_svgp_cavi_step(*args) == 0
    stale = m.get_matrix(capi.MAT_KNM, 0)
    assert L.agp_svgp_invalidate_data(h) == 0 and L.agp_svgp_cavi_step(*args) == 0
    fresh = m.get_matrix(capi.MAT_KNM, 0)
    ref = R.Kernel("sqexponential", 2.0, 1.0)
    assert _rel(stale, ref.matrix(X1, Z)) < 1e-12 and _rel(fresh, ref.matrix(X2, Z)) < 1e-12


def test_two_devices_are_left_alone(mods):
    """entry points select the ctx's device for their own duration only and restore the caller's."""
    AGP, R, capi, torch = mods
    before = torch.cuda.current_device()
    rng = np.random.default_rng(0)
    X = rng.random((100, 2))
    m = AGP.SVGP(AGP.SqExponentialKernel(), AGP.GaussianLikelihood(0.1), AGP.AnalyticVI(), X[:10].copy(), optimiser=False,
                 device=0)
    AGP.train_(m, X, np.sin(X[:, 0]), 2)
    AGP.predict_f(m, X[:5], cov=True)
    assert torch.cuda.current_device() == before
    cur = C.c_int(-1)
    hip = C.CDLL("libamdhip64.so")
    assert hip.hipGetDevice(C.byref(cur)) == 0 and cur.value == before


def test_multilatent_task_graph_soak_bitwise(mods):
    """The check that found the round-1 ordering hole (commit 7e42634), shortened: 8-class LogisticSoftMax at the C4 shape
    (m = B = 1024, two interleaved task-graph launches of four problems per step), 2 x 300 steps, eta2 of every latent compared
    BITWISE between the runs."""
    AGP, R, capi, torch = mods
    K, steps, m, B, D, N = 8, 300, 1024, 1024, 32, 100_000
    rng = np.random.default_rng(0)
    X = rng.random((N, D))
    y = 1 + rng.integers(K, size=N)
    Z = X[rng.permutation(N)[:m]].copy()
    idx = np.stack([rng.choice(N, B, replace=False) for _ in range(64)])
    out = []
    L = capi.lib()
    for rep in range(2):
        model = AGP.SVGP(AGP.with_lengthscale(AGP.SqExponentialKernel(), np.sqrt(D) / 4), AGP.LogisticSoftMaxLikelihood(K),
                         AGP.AnalyticSVI(B), Z, optimiser=False)
        AGP.train_(model, X, y, 1, idx_stream=idx[:1])
        h = model._h
        Xd, yd, _ = model._data
        ia = torch.as_tensor(idx, device="cuda")
        for i in range(steps):
            j = i % 64
            assert L.agp_svgp_cavi_step(h, C.c_void_p(Xd.data_ptr()), Xd.stride(0), C.c_void_p(yd.data_ptr()),
                                        C.c_void_p(ia[j].data_ptr()), B, N / B) == 0
            L.agp_svgp_prefetch(h, C.c_void_p(Xd.data_ptr()), Xd.stride(0), C.c_void_p(ia[(j + 1) % 64].data_ptr()), B)
        model._chk(L.agp_svgp_check_status(h))
        out.append([model.get_state(k)[3].copy() for k in range(K)])
        assert all(np.isfinite(e).all() for e in out[-1])
        del model
    for k in range(K):
        assert np.array_equal(out[0][k], out[1][k]), f"latent {k}: runs differ"


def _fallback_child(q, K):
    """runs in a fresh process: AGP_DAG_TEST_ABORT=1 makes every task-graph launch of a CAVI step look like it lost a dependency"""
    try:
        import sys

        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import agp_amd as AGP
        from agp_amd import capi

        rng = np.random.default_rng(31)
        N, D, m, B, iters = 3000, 4, 200, 256, 6
        X = rng.random((N, D))
        f = np.sin(4 * X[:, 0]) + X[:, 1]
        y = (f > f.mean()).astype(int) if K == 1 else 1 + np.digitize(f, np.quantile(f, [0.33, 0.66]))
        Z = X[rng.permutation(N)[:m]].copy()
        idx = [rng.choice(N, B, replace=False) for _ in range(iters)]
        lik = AGP.LogisticLikelihood() if K == 1 else AGP.LogisticSoftMaxLikelihood(3)
        ma = AGP.SVGP(1.5 * (AGP.SqExponentialKernel() @ AGP.ScaleTransform(2.0)), lik, AGP.AnalyticSVI(B), Z, optimiser=False)
        AGP.train_(ma, X, y, iters, idx_stream=idx)
        n = C.c_int64()
        f_ = capi.lib().agp_dev_dag_retries
        f_.restype, f_.argtypes = C.c_int32, [C.c_void_p, C.POINTER(C.c_int64)]
        assert f_(ma._ctx, C.byref(n)) == 0
        q.put((int(n.value), [ma.get_state(k)[3] for k in range(ma.n_latent)], X, y, Z, idx))
    except BaseException as e:
        q.put(repr(e))


@pytest.mark.parametrize("K", [1, 3])
def test_task_graph_fallback_reruns_the_factorisation_in_stream(mods, K):
    """a task-graph launch that loses a dependency (info = -1) is re-run by k_chol_safe on the same stream --
    inputs restored from eta2 / kappa / eta1, per-column algorithm with grid barriers -- so the step neither fails nor stalls.
    The loss is simulated after every launch (AGP_DAG_TEST_ABORT=1, read at library load: hence a child process); several block
    columns (m = 200 -> 4) and, for K = 3, the interleaved multi-problem graph.  The result must match the oracle."""
    import multiprocessing as mp

    AGP, R, capi, torch = mods
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    os.environ["AGP_DAG_TEST_ABORT"] = "1"
    try:
        p = ctx.Process(target=_fallback_child, args=(q, K))
        p.start()
        got = q.get(timeout=600)
        p.join(timeout=60)
    finally:
        del os.environ["AGP_DAG_TEST_ABORT"]
    assert not isinstance(got, str), got
    retries, eta2, X, y, Z, idx = got
    if not K_.no_task_graph():
        assert retries >= 6  # every step went through the fallback
    lik = R.LogisticLikelihood() if K == 1 else R.LogisticSoftMaxLikelihood(3)
    mr = R.SVGP(R.Kernel("sqexponential", 2.0, 1.5), lik, Z, stochastic=True, batchsize=256)
    mr.train(X, y, len(idx), idx_stream=idx)
    for k in range(len(eta2)):
        assert _rel(eta2[k], mr.latents[k].eta2) < 1e-8


def _probation_child(q):
    """fresh process with AGP_DAG_TEST_ABORT=1: every task-graph launch 'loses a dependency'; the host notices at the end of each
    train_ call (its status check) and pauses the task graph for 512 steps"""
    try:
        import sys

        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import agp_amd as AGP
        from agp_amd import capi

        rng = np.random.default_rng(32)
        N, D, m, B = 2000, 3, 130, 128
        X = rng.random((N, D))
        y = (np.sin(4 * X[:, 0]) + X[:, 1] > 1.0).astype(int)
        Z = X[rng.permutation(N)[:m]].copy()
        idx = [rng.choice(N, B, replace=False) for _ in range(6 + 500 + 40)]
        ma = AGP.SVGP(1.5 * (AGP.SqExponentialKernel() @ AGP.ScaleTransform(2.0)), AGP.LogisticLikelihood(), AGP.AnalyticSVI(B), Z,
                      optimiser=False)
        f_ = capi.lib().agp_dev_dag_retries
        f_.restype, f_.argtypes = C.c_int32, [C.c_void_p, C.POINTER(C.c_int64)]
        counts, st = [], None
        for lo, hi in ((0, 6), (6, 506), (506, 546)):
            _, st = AGP.train_(ma, X, y, hi - lo, idx_stream=idx[lo:hi], state=st)
            n = C.c_int64()
            assert f_(ma._ctx, C.byref(n)) == 0
            counts.append(int(n.value))
        q.put((counts, ma.get_state(0)[3], X, y, Z, idx))
    except BaseException as e:
        import traceback

        q.put(repr(e) + traceback.format_exc())


def test_task_graph_is_tried_again_after_a_pause(mods):
    """A lost dependency no longer switches the context to plain launches for good: after 512 steps the task graph is tried again
    (here it 'fails' again, by construction, and the next pause is four times longer).  Counts of in-stream re-runs after 6, 506
    and 546 steps: the first call's six, nothing during the pause, again from step 513 on.  The posterior still matches the oracle."""
    import multiprocessing as mp

    AGP, R, capi, torch = mods
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    os.environ["AGP_DAG_TEST_ABORT"] = "1"
    try:
        p = ctx.Process(target=_probation_child, args=(q,))
        p.start()
        got = q.get(timeout=900)
        p.join(timeout=60)
    finally:
        del os.environ["AGP_DAG_TEST_ABORT"]
    assert not isinstance(got, str), got
    counts, eta2, X, y, Z, idx = got
    # every step of the first call went through the fallback -- and (round 3) so did the factorisation of K_ZZ at the start of
    # train!, which has an in-stream fallback of its own now
    if not K_.no_task_graph():
        assert counts[0] == 7
        assert counts[1] == counts[0]              # paused: 500 steps of plain launches, no task graph, nothing to re-run
        assert counts[2] > counts[1]               # steps 513.. use the task graph again (and lose it again)
    mr = R.SVGP(R.Kernel("sqexponential", 2.0, 1.5), R.LogisticLikelihood(), Z, stochastic=True, batchsize=128)
    mr.train(X, y, len(idx), idx_stream=idx)
    assert _rel(eta2, mr.latents[0].eta2) < 1e-7


def test_lookahead_handover_word_equals_event_protocol(mods):
    """The CAVI step's task graph tells the look-ahead stream through a word in signal memory that the previous step has released
    its kappa buffers (DagSync, agp_chol.h) instead of recording an event on the step's stream.  Ordering only: the trajectory
    must be bitwise the one of the event protocol (AGP_PF_INKERNEL=0, read when a handle creates its look-ahead stream), through
    the fused step (train_) and through the phase API (step_local / step_stats / step_global with a prefetch in between)."""
    AGP, R, capi, torch = mods
    rng = np.random.default_rng(11)
    N, D, m, B, it = 4000, 4, 256, 256, 60
    X = rng.random((N, D))
    y = np.sign(np.sin(X @ rng.standard_normal(D)) + 0.1 * rng.standard_normal(N))
    Z = X[rng.permutation(N)[:m]].copy()
    idx = [rng.choice(N, B, replace=False) for _ in range(it)]
    L = capi.lib()

    def run(word, phases):
        os.environ["AGP_PF_INKERNEL"] = "1" if word else "0"
        try:
            ma = AGP.SVGP(AGP.SqExponentialKernel() @ AGP.ScaleTransform(2.0), AGP.LogisticLikelihood(), AGP.AnalyticSVI(B), Z,
                          optimiser=False)
            AGP.train_(ma, X, y, 1, idx_stream=idx[:1])  # creates the handle and uploads the data
            if not phases:
                AGP.train_(ma, X, y, it - 1, idx_stream=idx[1:], state=True)
            else:
                h = ma._h
                Xd, yd, _ = ma._data
                ia = torch.as_tensor(np.stack(idx), device="cuda")
                xp, yp, ld = C.c_void_p(Xd.data_ptr()), C.c_void_p(yd.data_ptr()), Xd.stride(0)
                for i in range(1, it):
                    ip = C.c_void_p(ia[i].data_ptr())
                    assert L.agp_svgp_step_local(h, xp, ld, yp, ip, B, N / B) == 0
                    assert L.agp_svgp_step_stats(h) == 0
                    if i + 1 < it:  # the look-ahead between the statistics and the global step, as the batch-parallel driver does
                        assert L.agp_svgp_prefetch(h, xp, ld, C.c_void_p(ia[i + 1].data_ptr()), B) == 0
                    assert L.agp_svgp_step_global(h) == 0
                ma._chk(L.agp_svgp_check_status(h))
            return ma.get_state(0)
        finally:
            del os.environ["AGP_PF_INKERNEL"]

    for phases in (False, True):
        a, b = run(True, phases), run(False, phases)
        for u, v in zip(a, b):
            assert np.array_equal(u, v), phases
    # and the fused and the phase-split sequences agree with each other to rounding
    assert _rel(run(True, False)[3], run(True, True)[3]) < 1e-10


def test_two_handles_interleaved_with_lookahead(mods):
    """Two models of one process share the context's stream, its task-graph flags and hand-over area, but each has its own
    look-ahead stream and hand-over word.  Stepping them alternately (each announcing its own next minibatch) must give each
    exactly the trajectory it has when trained alone."""
    AGP, R, capi, torch = mods
    rng = np.random.default_rng(21)
    N, D, B, it = 6000, 5, 256, 40
    X = rng.random((N, D))
    y = np.sign(np.sin(X @ rng.standard_normal(D)) + 0.1 * rng.standard_normal(N))
    L = capi.lib()
    specs = [(256, 1.5), (384, 2.5)]
    Zs = [X[rng.permutation(N)[:m]].copy() for m, _ in specs]
    idx = np.stack([rng.choice(N, B, replace=False) for _ in range(it)])

    def make(k):
        m, sc = specs[k]
        ma = AGP.SVGP(AGP.SqExponentialKernel() @ AGP.ScaleTransform(sc), AGP.LogisticLikelihood(), AGP.AnalyticSVI(B), Zs[k],
                      optimiser=False)
        AGP.train_(ma, X, y, 1, idx_stream=list(idx[:1]))
        Xd, yd, _ = ma._data
        return ma, (C.c_void_p(Xd.data_ptr()), C.c_void_p(yd.data_ptr()), Xd.stride(0))

    ia = torch.as_tensor(idx, device="cuda")

    def step(ma, ptrs, i):
        xp, yp, ld = ptrs
        assert L.agp_svgp_cavi_step(ma._h, xp, ld, yp, C.c_void_p(ia[i].data_ptr()), B, N / B) == 0
        if i + 1 < it:
            assert L.agp_svgp_prefetch(ma._h, xp, ld, C.c_void_p(ia[i + 1].data_ptr()), B) == 0

    solo = []
    for k in range(2):
        ma, ptrs = make(k)
        for i in range(1, it):
            step(ma, ptrs, i)
        ma._chk(L.agp_svgp_check_status(ma._h))
        solo.append(ma.get_state(0))
    pair = [make(0), make(1)]
    for i in range(1, it):
        for ma, ptrs in pair:
            step(ma, ptrs, i)
    for k, (ma, _) in enumerate(pair):
        ma._chk(L.agp_svgp_check_status(ma._h))
        for u, v in zip(ma.get_state(0), solo[k]):
            assert np.array_equal(u, v), k
