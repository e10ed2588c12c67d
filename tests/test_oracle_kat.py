"""CPU tests pinning the oracle (oracle/agp_ref.py): closed-form known answers, the reference's own unit-test
identities (test/functions/utils.jl, test/likelihood/multiclass.jl), mpmath tables, behavioural thresholds of
test/testingtools.jl, and the committed golden fixtures.  No GPU needed."""
import glob
import math
import os

import numpy as np
import pytest

from oracle import agp_ref as R

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_jitter_constants():
    # test/functions/utils.jl:2-5
    assert R.jitt(np.float64) == pytest.approx(1e-4)
    assert R.jitt(np.float32) == pytest.approx(1e-3)
    assert R.jitt(np.float16) == pytest.approx(1e-2)


def test_utils_identities():
    # test/functions/utils.jl:7-30, 47-49
    rng = np.random.default_rng(0)
    A, B, x = rng.random((2, 2)), rng.random((2, 2)), rng.random(2)
    assert R.delta(0, 1) == 0.0 and R.delta(1, 1) == 1.0
    assert np.array_equal(R.hadamard(A, B), A * B)
    assert np.allclose(R.add_transpose(A), A + A.T)
    Dm = A @ A.T + np.eye(2)
    Lc = np.linalg.cholesky(Dm)
    assert R.invquad(Lc, x) == pytest.approx(x @ np.linalg.solve(Dm, x))
    assert R.trace_ABt(A, B) == pytest.approx(np.trace(A @ B.T))
    assert np.allclose(R.diag_ABt(A, B), np.diag(A @ B.T))
    assert np.allclose(R.diagv_B(x, B), np.diag(x) @ B)
    assert np.allclose(R.kappa_diag_theta_kappa(A, x), A.T @ np.diag(x) @ A)
    assert np.allclose(R.rho_kappa_diag_theta_kappa(2.0, A, x), 2.0 * A.T @ np.diag(x) @ A)
    assert np.allclose(R.opt_add_diag_mat(x, A), A + np.diag(x))
    assert R.safe_expcosh(2.0, 1.0) == pytest.approx(np.exp(2.0) / np.cosh(1.0))
    assert R.logcosh(2.0) == pytest.approx(np.log(np.cosh(2.0)))
    # overflow fallback of safe_expcosh (utils.jl:84-86)
    assert np.isfinite(R.safe_expcosh(800.0, 900.0))


def test_multiclass_label_mapping():
    # test/likelihood/multiclass.jl:1-40
    y = [1, 2, 3, 1, 1, 2, 3]
    l = R.LogisticSoftMaxLikelihood(3)
    R.create_mapping(l, y)
    assert sorted(l.class_mapping) == [1, 2, 3]
    assert l.ind_mapping == {1: 1, 2: 2, 3: 3}
    assert np.array_equal(R.create_one_hot(l, y[:3]), np.eye(3, dtype=bool))
    with pytest.raises(RuntimeError):
        R.create_mapping(R.LogisticSoftMaxLikelihood(2), y)
    y = [1, 2, 1, 1]
    l = R.LogisticSoftMaxLikelihood(3)
    R.create_mapping(l, y)
    assert l.class_mapping == [1, 2, 3]
    assert np.array_equal(R.create_one_hot(l, y), np.array([[1, 0, 0], [0, 1, 0], [1, 0, 0], [1, 0, 0]], bool))
    y = ["b", "a", "c", "a", "a"]
    l = R.LogisticSoftMaxLikelihood(3)
    R.create_mapping(l, y)
    assert l.class_mapping == ["b", "a", "c"] and l.ind_mapping == {"b": 1, "a": 2, "c": 3}
    assert np.array_equal(R.create_one_hot(l, y),
                          np.array([[1, 0, 0], [0, 1, 0], [0, 0, 1], [0, 1, 0], [0, 1, 0]], bool))
    l = R.LogisticSoftMaxLikelihood(3)
    l.class_mapping = ["a", "b", "c"]
    assert np.array_equal(R.create_one_hot(l, y),
                          np.array([[0, 1, 0], [1, 0, 0], [0, 0, 1], [1, 0, 0], [1, 0, 0]], bool))


def test_binary_labels():
    # classification.jl:29-44
    l = R.LogisticLikelihood()
    assert np.array_equal(R.treat_labels(np.array([0, 1, 1, 0]), l), [-1, 1, 1, -1])
    assert np.array_equal(R.treat_labels(np.array([-1, 1]), l), [-1, 1])
    with pytest.raises(ValueError):
        R.treat_labels(np.array([0, 1, 2]), l)


def test_special_functions_vs_mpmath():
    mp = pytest.importorskip("mpmath")
    mp.mp.dps = 50
    for c in [1e-9, 1e-4, 0.3, 1.0, 7.5, 40.0]:
        ref = mp.tanh(mp.mpf(c) / 2) / (2 * mp.mpf(c))
        assert float(R.theta_pg(np.array([c]))[0]) == pytest.approx(float(ref), rel=1e-14)
        assert float(R.logcosh(c)) == pytest.approx(float(mp.log(mp.cosh(c))), rel=1e-13, abs=5e-16)  # the reference formula cancels near 0 (utils.jl:89-91)
    assert float(R.theta_pg(np.array([0.0]))[0]) == 0.25


def _toy(seed=0, N=120, D=2, m=15):
    rng = np.random.default_rng(seed)
    X = rng.random((N, D))
    f = np.sin(5 * X[:, 0]) * np.cos(3 * X[:, 1])
    return rng, X, f, X[rng.permutation(N)[:m]].copy()


def test_kat1_titsias_optimum():
    """Gaussian likelihood, full batch, rho = 1, mu0 = 0: after ONE step eta1 = kappa'y/s2,
    eta2 = -(kappa'kappa/s2 + Kinv)/2 (analyticVI.jl:168,179,241-242 + gaussian.jl:74-80); step two is a fixed point."""
    rng, X, f, Z = _toy()
    y = f + 0.1 * rng.standard_normal(len(f))
    s2 = 0.01
    kern = R.Kernel("sqexponential", 4.0, 1.0)
    M = R.SVGP(kern, R.GaussianLikelihood(s2), Z)
    M.train(X, y, 1)
    g = M.latents[0]
    K = kern.matrix(Z) + 1e-4 * np.eye(len(Z))
    kappa = np.linalg.solve(K, kern.matrix(Z, X)).T
    assert np.allclose(g.eta1, kappa.T @ y / s2, rtol=1e-10)
    assert np.allclose(g.eta2, -0.5 * (kappa.T @ kappa / s2 + np.linalg.inv(K)), rtol=1e-9, atol=1e-9)
    mu1, S1 = g.mu.copy(), g.Sigma.copy()
    M.train(X, y, 1)
    assert np.allclose(M.latents[0].mu, mu1, rtol=1e-9, atol=1e-12)
    assert np.allclose(M.latents[0].Sigma, S1, rtol=1e-9, atol=1e-12)
    # the optimum equals the Titsias posterior  Sigma = K (K + Kmn Knm / s2)^-1 K
    Kmn = kern.matrix(Z, X)
    S_t = K @ np.linalg.solve(K + Kmn @ Kmn.T / s2, K)
    assert np.allclose(S1, S_t, rtol=1e-7, atol=1e-10)


def test_kat2_exact_gp_limit():
    """Z = X, m = N: predict_f equals exact GP regression up to the jitter terms."""
    rng = np.random.default_rng(2)
    N = 40
    X = rng.random((N, 1))
    y = np.sin(6 * X[:, 0]) + 0.05 * rng.standard_normal(N)
    s2 = 0.05
    kern = R.Kernel("sqexponential", 5.0, 1.0)
    M = R.SVGP(kern, R.GaussianLikelihood(s2), X.copy())
    M.train(X, y, 2)
    Xt = rng.random((25, 1))
    mu, var = M.predict_f(Xt, cov=True)
    Kxx = kern.matrix(X) + 1e-4 * np.eye(N)
    Ks = kern.matrix(Xt, X)
    mu_exact = Ks @ np.linalg.solve(Kxx + s2 * np.eye(N), y)
    var_exact = kern.diag(Xt) + 1e-4 - np.sum(Ks * np.linalg.solve(Kxx + s2 * np.eye(N), Ks.T).T, axis=1)
    assert np.allclose(mu[0], mu_exact, atol=5e-3)
    assert np.allclose(var[0], var_exact, atol=5e-3)


@pytest.mark.parametrize("lik", ["gaussian", "logistic", "studentt", "logisticsoftmax"])
def test_kat6_elbo_monotone_full_batch(lik):
    rng, X, f, Z = _toy(3)
    if lik == "gaussian":
        L, y = R.GaussianLikelihood(0.05), f + 0.2 * rng.standard_normal(len(f))
    elif lik == "logistic":
        L, y = R.LogisticLikelihood(), (f > 0).astype(int)
    elif lik == "studentt":
        L, y = R.StudentTLikelihood(3.0), f + 0.2 * rng.standard_t(3, len(f))
    else:
        L, y = R.LogisticSoftMaxLikelihood(3), 1 + np.digitize(f, np.quantile(f, [0.33, 0.66]))
    M = R.SVGP(R.Kernel("sqexponential", 4.0, 1.0), L, Z, elbo_mode="corrected")
    es = []
    M.train(X, y, 12, callback=lambda m, it, xb, yb: es.append(m.elbo(yb)))
    d = np.diff(es)
    assert np.all(d > -1e-7 * np.abs(es[1:])), es


def test_reference_thresholds_on_reference_toy_setups():
    """test/testingtools.jl:223-253 (testconv) and :14,17 (proba_y variance > 0) on the reference's own set-ups:
    N = 20, d = 2, M = 10, SqExponentialKernel() ∘ ScaleTransform(10.0) (x2.0 for classification); 1 + 5 iterations."""
    rng = np.random.default_rng(42)
    N, d, Mi = 20, 2, 10
    X = rng.random((N, d))
    for lik, var in [(R.GaussianLikelihood(1e-3), 1.0), (R.StudentTLikelihood(3.0), 1.0), (R.LogisticLikelihood(), 2.0)]:
        kern = R.Kernel("sqexponential", 10.0, var)
        f = np.linalg.cholesky(kern.matrix(X) + 1e-5 * np.eye(N)) @ rng.standard_normal(N)
        y = (f > 0) if lik.name == "logistic" else f + 0.1 * rng.standard_normal(N)
        Z = X[rng.permutation(N)[:Mi]]
        for stoch in (False, True):
            M = R.SVGP(kern, lik, Z, stochastic=stoch, batchsize=10)
            idx = [rng.choice(N, 10, replace=False) for _ in range(6)]
            M.train(X, y, 1, idx_stream=idx)
            M.train(X, y, 5, idx_stream=idx[1:])
            yp = M.predict_y(X)
            if lik.name == "logistic":
                assert np.mean(yp != y) < 0.5
            else:
                assert np.mean(np.abs(yp - f)) < 15
            assert np.all(M.proba_y(X)[1] > 0)
    # multiclass: N = 100, d = 1, K = 3, error < 0.9
    X = rng.random((100, 1))
    y = 1 + np.digitize(X[:, 0], [0.33, 0.66])
    M = R.SVGP(R.Kernel("sqexponential", 10.0, 1.0), R.LogisticSoftMaxLikelihood(3), X[rng.permutation(100)[:10]])
    M.train(X, y, 6)
    assert np.mean(M.predict_y(X) != y) < 0.9


def test_robbins_monro_schedule():
    # optimisers.jl:12-19 : state starts at 1, lr_t = (tau + n)^-kappa ; first step 2^-0.51
    assert R.robbins_monro_lr(1) == pytest.approx(2.0 ** -0.51)
    assert R.robbins_monro_lr(2) == pytest.approx(3.0 ** -0.51)


def test_hyper_objective_matches_elbo_parts():
    """the hyper-parameter objective (ELBO.jl:15-21) at the current hypers equals ELBO + rho*AugmentedKL."""
    rng, X, f, Z = _toy(5)
    y = (f > 0).astype(int)
    M = R.SVGP(R.Kernel("sqexponential", 4.0, 1.3), R.LogisticLikelihood(), Z)
    yt = R.treat_labels(y, M.likelihood)
    M.train(X, yt, 3, labels_treated=True)
    M.compute_kernel_matrices(X, update=True)
    val = R.hyper_objective(M, X, yt, 0, 4.0, 1.3, Z, 1.0)
    assert val == pytest.approx(M.elbo(yt) + R.augmented_kl(M.likelihood, M.local_vars, yt), rel=1e-10)


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLD, "*.npz"))))
def test_oracle_reproduces_golden(path):
    """The committed fixtures are regenerated bit-for-bit-ish by the oracle (guards against silent oracle edits)."""
    g = np.load(path, allow_pickle=True)
    name = os.path.basename(path)
    from _liks import oracle_lik

    lik = oracle_lik(R, name.split("_")[0])
    M = R.SVGP(R.Kernel("sqexponential", float(g["scale"]), float(g["variance"])), lik, g["Z"],
               stochastic=bool(g["stochastic"]), batchsize=int(g["B"]))
    es = []
    M.train(g["X"], g["y"], 10, idx_stream=g["idx"], callback=lambda m, it, xb, yb: es.append(m.elbo(yb)))
    assert np.allclose(es, g["elbo"], rtol=1e-10)
    for k, lat in enumerate(M.latents):
        assert np.allclose(lat.eta2, g[f"eta2_it10_l{k}"], rtol=1e-10, atol=1e-12)
        assert np.allclose(lat.mu, g[f"mu_it10_l{k}"], rtol=1e-9, atol=1e-12)
    mu, var = M.predict_f(g["Xt"], cov=True)
    assert np.allclose(np.stack(mu), g["pred_mu"], rtol=1e-9, atol=1e-12)
    assert np.allclose(np.stack(var), g["pred_var"], rtol=1e-8, atol=1e-12)
    if "lam" in g.files and hasattr(lik, "lam"):
        assert lik.lam == pytest.approx(float(g["lam"][-1]), rel=1e-12)


def test_half_order_bessel_closed_forms():
    """The GIG terms of the Laplace / BayesianSVM ELBO (KLdivergences.jl:105-113, bayesiansvm.jl:89-92) call besselk at
    orders 1/2, 3/2, -1/2 only; the closed forms the oracle and the HIP kernel use agree with scipy's besselk."""
    from scipy.special import kv

    s = np.array([1e-3, 0.05, 0.7, 3.0, 25.0, 300.0])
    assert np.allclose(R.log2besselk_half(s[:5]), np.log(2.0 * kv(0.5, s[:5])), rtol=1e-13, atol=1e-13)
    ratio = s[:5] / kv(0.5, s[:5]) * (kv(1.5, s[:5]) + kv(-0.5, s[:5])) / 2.0
    assert np.allclose(ratio, s[:5] + 0.5, rtol=1e-12)


def test_gauss_hermite_expectation_of_logistic():
    """expectation(logistic, mu, s2) (utils.jl:16-19) against adaptive quadrature."""
    from scipy.integrate import quad

    for mu, s2 in [(0.3, 0.5), (-2.0, 4.0), (1.5, 1e-3)]:
        ref = quad(lambda t: R.logistic(mu + math.sqrt(s2) * t) * math.exp(-t * t / 2) / math.sqrt(2 * math.pi), -12, 12,
                   epsabs=1e-13, epsrel=1e-13)[0]
        assert R.expectation_logistic(np.array([mu]), np.array([s2]))[0] == pytest.approx(ref, rel=1e-10)


def _new_lik_case(name, seed=3, N=150, m=12):
    from _liks import labels, oracle_lik

    rng, X, f, Z = _toy(seed, N=N, D=2, m=m)
    lik = oracle_lik(R, name)
    y = labels(name, f, X, rng)
    return rng, X, f, Z, lik, y


@pytest.mark.parametrize("name", ["laplace", "bayesiansvm", "poisson", "negbinomial", "heteroscedastic"])
def test_new_likelihood_local_updates_follow_the_reference_formulas(name):
    """One local update evaluated by hand from the reference source lines."""
    rng = np.random.default_rng(5)
    B = 9
    mu, s2 = rng.standard_normal(B), rng.random(B) + 0.1
    lik = __import__("_liks").oracle_lik(R, name)
    lv = R.init_local_vars(lik, B)
    if name == "laplace":
        y = rng.standard_normal(B)
        lv = R.local_updates(lv, lik, y, (mu,), (s2,))
        b = np.sqrt((mu - y) ** 2 + s2)                                            # laplace.jl:67
        assert np.allclose(lv["theta"], (1 / 0.4) / b)                             # sqrt(a)/b, a = beta^-2  :68-70
        assert np.allclose(R.grad_E_mu(lik, y, lv)[0], lv["theta"] * y)            # :85-87
    elif name == "bayesiansvm":
        y = np.sign(rng.standard_normal(B))
        lv = R.local_updates(lv, lik, y, (mu,), (s2,))
        assert np.allclose(lv["c"], (1 - y * mu) ** 2 + s2)                        # bayesiansvm.jl:50-52
        assert np.allclose(lv["theta"], lv["c"] ** -0.5)                           # :53
        assert np.allclose(R.grad_E_mu(lik, y, lv)[0], y * (lv["theta"] + 1))      # :57-61
    elif name == "poisson":
        y = rng.poisson(2.0, B).astype(float)
        lam0 = lik.lam
        lv = R.local_updates(lv, lik, y, (mu,), (s2,))
        c = np.sqrt(mu ** 2 + s2)
        g = lam0 * np.exp(-mu / 2) / np.cosh(c / 2) / 2                            # poisson.jl:72-74
        assert np.allclose(lv["gamma"], g)
        assert np.allclose(lv["theta"], (y + g) / c * np.tanh(c / 2))              # :75-77
        assert lik.lam == pytest.approx(y.sum() / R.expectation_logistic(mu, s2).sum())  # :78
        assert np.allclose(R.grad_E_mu(lik, y, lv)[0], (y - g) / 2)                # :94-98
    elif name == "negbinomial":
        y = rng.poisson(3.0, B).astype(float)
        lv = R.local_updates(lv, lik, y, (mu,), (s2,))
        c = np.sqrt(mu ** 2 + s2)
        assert np.allclose(lv["theta"], (6.0 + y) * np.tanh(c / 2) / c)            # negativebinomial.jl:77-79
        assert np.allclose(R.grad_E_mu(lik, y, lv)[0], (y - 6.0) / 2)              # :94-96
    else:
        y = rng.standard_normal(B)
        mu2, s22 = rng.standard_normal(B), rng.random(B) + 0.1
        lam0 = lik.lam
        lv = R.local_updates(lv, lik, y, (mu, mu2), (s2, s22))
        phi = ((mu - y) ** 2 + s2) / 2                                             # heteroscedastic.jl:80-82
        c = np.sqrt(mu2 ** 2 + s22)                                                # :83
        sg = np.exp(-mu2 / 2) / np.cosh(c / 2) / 2                                 # :84-86
        gam = lam0 * phi * sg                                                      # :87-89
        assert np.allclose(lv["gamma"], gam)
        assert np.allclose(lv["theta"], (0.5 + gam) * np.tanh(c / 2) / (2 * c))    # :90-92
        assert lik.lam == pytest.approx(max(B / (2 * np.dot(phi, 1 - sg)), lam0))  # :95
        g1 = R.grad_E_mu(lik, y, lv)
        assert np.allclose(g1[0], y * lik.lam * sg / 2) and np.allclose(g1[1], (0.5 - gam) / 2)  # :113-120
        g2 = R.grad_E_Sigma(lik, y, lv)
        assert np.allclose(g2[0], lik.lam * sg / 2) and np.allclose(g2[1], lv["theta"] / 2)      # :122-129


def test_laplace_cavi_increases_the_collapsed_bound():
    """With q(omega) at its optimum the augmented bound collapses to sum_i [-log(2 beta) - sqrt(E(y_i - f_i)^2)/beta] - KL(q(u)||p(u));
    full-batch CAVI with the reference's updates (laplace.jl:60-90) must never decrease it."""
    rng, X, f, Z, lik, y = _new_lik_case("laplace")
    M = R.SVGP(R.Kernel("sqexponential", 3.0, 1.0), lik, Z)
    vals = []

    def cb(m, it, xb, yb):
        b = np.sqrt((m.mean_f()[0] - yb) ** 2 + m.var_f()[0])
        vals.append(np.sum(-math.log(2 * lik.beta) - b / lik.beta)
                    - R.gaussian_kl(m.latents[0].mu, m.latents[0].mu0, m.latents[0].Sigma, m.latents[0].L))

    M.train(X, y, 12, callback=cb)
    assert np.all(np.diff(vals) > -1e-9), vals


def test_heteroscedastic_elbo_monotone_full_batch():
    rng, X, f, Z, lik, y = _new_lik_case("heteroscedastic")
    M = R.SVGP(R.Kernel("sqexponential", 3.0, 1.0), lik, Z)
    es = []
    M.train(X, y, 12, callback=lambda m, it, xb, yb: es.append(m.elbo(yb)))
    assert np.all(np.diff(es) > -1e-8), es


@pytest.mark.parametrize("name,tol", [("laplace", 0.4), ("bayesiansvm", 0.3), ("poisson", 2.0), ("negbinomial", 8.0),
                                      ("heteroscedastic", 0.6)])
def test_new_likelihoods_reference_style_thresholds(name, tol):
    """Same spirit as test/testingtools.jl:223-253: a few iterations must give a usable predictor and positive proba_y
    variances (test/testingtools.jl:14,17)."""
    rng, X, f, Z, lik, y = _new_lik_case(name, N=200, m=20)
    M = R.SVGP(R.Kernel("sqexponential", 3.0, 1.0), lik, Z)
    M.train(X, y, 8)
    py = np.asarray(M.predict_y(X), dtype=float)
    yt = R.treat_labels(y, lik)
    if name == "bayesiansvm":
        assert np.mean(py != (yt > 0)) < tol
    elif name in ("laplace", "heteroscedastic"):
        assert np.mean(np.abs(py - f)) < tol
    else:
        assert np.mean(np.abs(py - yt)) < tol
    pr = M.proba_y(X)
    assert np.all(pr[1] > 0)


def test_event_labels_must_be_integers():
    with pytest.raises(ValueError):
        R.treat_labels(np.array([1.0, 2.0]), R.PoissonLikelihood(2.0))  # event.jl:11-13


@pytest.mark.parametrize("likname,kind", [("laplace", "sqexponential"), ("bayesiansvm", "matern32"),
                                          ("poisson", "sqexponential"), ("negbinomial", "matern52"),
                                          ("heteroscedastic", "sqexponential")])
@pytest.mark.parametrize("mode", ["corrected", "reference"])
def test_kat7_new_likelihoods_hyper_gradient_vs_finite_differences(likname, kind, mode):
    rng, X, f, Z, L, y = _new_lik_case(likname, seed=7, N=40, m=6)
    M = R.SVGP(R.Kernel(kind, np.array([2.0, 3.0]), 1.4), L, Z, elbo_mode=mode)
    yt = R.treat_labels(y, L)
    M.train(X, yt, 2, labels_treated=True)
    M.compute_kernel_matrices(X, update=True)
    h = 1e-6
    for lat in range(len(M.latents)):
        g = R.hyper_gradient(M, X, yt, lat, 1.7)
        gp = M.latents[lat]
        sc0, v0, Z0 = np.array(gp.kernel.scale, float), gp.kernel.sigma2, gp.Z.copy()
        obj = lambda sc, v, Zz: R.hyper_objective(M, X, yt, lat, sc, v, Zz, 1.7)
        fd_v = (obj(sc0, v0 + h, Z0) - obj(sc0, v0 - h, Z0)) / (2 * h)
        assert g["dvariance"] == pytest.approx(fd_v, rel=5e-6, abs=1e-6)
        for d in range(2):
            e = np.zeros(2)
            e[d] = h
            fd = (obj(sc0 + e, v0, Z0) - obj(sc0 - e, v0, Z0)) / (2 * h)
            assert g["dscale"][d] == pytest.approx(fd, rel=5e-6, abs=1e-6)
        for (a, d) in [(0, 0), (3, 1)]:
            Zp, Zm = Z0.copy(), Z0.copy()
            Zp[a, d] += h
            Zm[a, d] -= h
            fd = (obj(sc0, v0, Zp) - obj(sc0, v0, Zm)) / (2 * h)
            assert g["dZ"][a, d] == pytest.approx(fd, rel=1e-5, abs=2e-6)


@pytest.mark.parametrize("likname,kind", [("logistic", "sqexponential"), ("studentt", "matern52"),
                                          ("gaussian", "matern32"), ("logisticsoftmax", "sqexponential")])
def test_kat7_hyper_gradient_vs_finite_differences(likname, kind):
    """analytic hyper-gradient (formula sheet) == central finite differences of the objective the reference
    hands to Zygote (autotuning.jl:96-98)."""
    rng, X, f, Z = _toy(7, N=40, D=2, m=6)
    if likname == "gaussian":
        L, y = R.GaussianLikelihood(0.1), f + 0.1 * rng.standard_normal(len(f))
    elif likname == "logistic":
        L, y = R.LogisticLikelihood(), (f > 0).astype(int)
    elif likname == "studentt":
        L, y = R.StudentTLikelihood(3.0), f + 0.1 * rng.standard_t(3, len(f))
    else:
        L, y = R.LogisticSoftMaxLikelihood(3), 1 + np.digitize(f, np.quantile(f, [0.33, 0.66]))
    M = R.SVGP(R.Kernel(kind, np.array([2.0, 3.0]), 1.4), L, Z)
    yt = R.treat_labels(y, L)
    M.train(X, yt, 2, labels_treated=True)
    M.compute_kernel_matrices(X, update=True)
    lat = M.n_latent - 1 if hasattr(M, "n_latent") else 0
    lat = len(M.latents) - 1
    g = R.hyper_gradient(M, X, yt, lat, 1.7)
    gp = M.latents[lat]
    sc0, v0, Z0 = np.array(gp.kernel.scale, float), gp.kernel.sigma2, gp.Z.copy()
    obj = lambda sc, v, Zz: R.hyper_objective(M, X, yt, lat, sc, v, Zz, 1.7)
    h = 1e-6
    fd_v = (obj(sc0, v0 + h, Z0) - obj(sc0, v0 - h, Z0)) / (2 * h)
    assert g["dvariance"] == pytest.approx(fd_v, rel=2e-6, abs=1e-7)
    for d in range(2):
        e = np.zeros(2)
        e[d] = h
        fd = (obj(sc0 + e, v0, Z0) - obj(sc0 - e, v0, Z0)) / (2 * h)
        assert g["dscale"][d] == pytest.approx(fd, rel=2e-6, abs=1e-7)
    for (a, d) in [(0, 0), (3, 1), (5, 0)]:
        Zp, Zm = Z0.copy(), Z0.copy()
        Zp[a, d] += h
        Zm[a, d] -= h
        fd = (obj(sc0, v0, Zp) - obj(sc0, v0, Zm)) / (2 * h)
        assert g["dZ"][a, d] == pytest.approx(fd, rel=5e-6, abs=1e-6)


def test_kmeans_restatement_properties():
    """inducingpoints(KmeansAlg(m), X) restated (InducingPoints.jl / Clustering.jl are unvendored): GEMM-form nearest centre
    == brute force, Lloyd cost never increases, the result is a fixed point (centres = cluster means), separated blobs are
    recovered, the seeding returns data points and is reproducible."""
    rng = np.random.default_rng(11)
    X = rng.random((500, 3))
    Cc = rng.random((17, 3))
    lab, mind = R.nearest_center(X, Cc)
    d = ((X[:, None, :] - Cc[None, :, :]) ** 2).sum(-1)
    assert np.array_equal(lab, d.argmin(1)) and np.allclose(mind, d.min(1), rtol=1e-12, atol=1e-14)
    means = np.array([[0.0, 0.0], [5.0, 5.0], [0.0, 6.0], [7.0, -1.0]])
    Xb = np.concatenate([mu + 0.2 * rng.standard_normal((80, 2)) for mu in means])
    seeds = R.kmeans_seeding(Xb, 4, 10, np.random.default_rng(3))
    assert all(any(np.array_equal(s, x) for x in Xb) for s in seeds)
    assert np.array_equal(seeds, R.kmeans_seeding(Xb, 4, 10, np.random.default_rng(3)))
    costs = []
    Cc = seeds.copy()
    for _ in range(6):
        lab, mind = R.nearest_center(Xb, Cc)
        costs.append(mind.sum())
        Cc = np.stack([Xb[lab == j].mean(0) if np.any(lab == j) else Cc[j] for j in range(4)])
    assert np.all(np.diff(costs) <= 1e-12)
    Cf, labf, it, obj, conv = R.kmeans_lloyd(Xb, seeds, tol=1e-3)
    assert conv and it <= 20
    for j in range(4):
        assert np.allclose(Cf[j], Xb[labf == j].mean(0), atol=1e-12)
    assert max(np.min(np.linalg.norm(Cf - mu, axis=1)) for mu in means) < 0.1


def test_oips_and_online_svgp_restatement_properties():
    """OIPS keeps every pair of inducing points below rho_accept and covers the data (every point has a neighbour in Z at
    or above rho_accept); OnlineSVGP (onlinetraining.jl) improves as batches stream in, its first batch reproduces the
    closed form eta2 = -(kappa' diag(theta) kappa + I/2 + K^-1/2) of the init_opt_state prior (states.jl:85-97), and
    extraKL of the first batch is -(tr Sigma + mu'mu)/2 (KLdivergences.jl:30-54 with kappa_a = I, K~_a = 0)."""
    rng = np.random.default_rng(4)
    N = 300
    X = rng.random((N, 1)) * 6
    f = np.sin(2 * X[:, 0])
    y = f + 0.1 * rng.standard_normal(N)
    ker = R.Kernel("sqexponential", 2.0, 1.0)
    alg = R.OIPS(0.8)
    Z = alg.init(X, ker)
    Kz = ker.matrix(Z)
    assert np.max(Kz - np.eye(len(Z))) < 0.8
    assert np.all(ker.matrix(X, Z).max(axis=1) >= 0.8 - 1e-12)
    M = R.OnlineSVGP(ker, R.GaussianLikelihood(0.01), R.OIPS(0.8))
    M.train(X[:60], y[:60], 1)
    g = M.latents[0]
    k = len(g["Z"])
    want = -(g["kappa"].T @ (g["kappa"] / 0.01) / 2.0 + np.eye(k) / 2.0 + g["Kinv"] / 2.0)
    assert np.allclose(g["eta2"], want, rtol=1e-12, atol=1e-12)
    assert M.extra_kl() == pytest.approx(-(np.trace(g["Sigma"]) + g["mu"] @ g["mu"]) / 2.0, rel=1e-12)
    errs = [np.mean(np.abs(M.predict_y(X) - f))]
    for b in range(60, N, 60):
        M.train(X[b:b + 60], y[b:b + 60], 3)
        errs.append(np.mean(np.abs(M.predict_y(X) - f)))
        assert np.isfinite(M.elbo(y[b:b + 60]))
    assert errs[-1] < 0.05 and errs[-1] < errs[0]
    assert len(M.latents[0]["Z"]) >= k


def test_kat7_multioutput_hyper_gradient_vs_finite_differences():
    rng = np.random.default_rng(0)
    N, D, m, Q = 60, 2, 6, 3
    X = rng.random((N, D))
    ys = [np.sin(4 * X[:, 0]) + 0.1 * rng.standard_normal(N), np.sign(X[:, 1] - 0.5 + 0.1 * rng.standard_normal(N))]
    A = rng.standard_normal((2, Q))
    A /= np.linalg.norm(A, axis=1, keepdims=True)
    Zs = [X[rng.permutation(N)[:m]].copy() for _ in range(Q)]
    M = R.MOSVGP(R.Kernel("matern52", np.array([2.0, 3.0]), 1.3), [R.GaussianLikelihood(0.05), R.LogisticLikelihood()], Zs, A)
    M.train(X, ys, 2)
    h = 1e-6
    for q in range(Q):
        g = M.hyper_gradient(X, ys, q)
        gp = M.latents[q]
        sc0, v0, Z0 = np.array(gp.kernel.scale, float), gp.kernel.sigma2, gp.Z.copy()
        obj = lambda sc, v, Z: M.hyper_objective(X, ys, q, sc, v, Z)
        assert g["dvariance"] == pytest.approx((obj(sc0, v0 + h, Z0) - obj(sc0, v0 - h, Z0)) / (2 * h), rel=5e-6, abs=1e-6)
        e = np.array([0.0, h])
        assert g["dscale"][1] == pytest.approx((obj(sc0 + e, v0, Z0) - obj(sc0 - e, v0, Z0)) / (2 * h), rel=5e-6, abs=1e-6)
        Zp, Zm = Z0.copy(), Z0.copy()
        Zp[2, 1] += h
        Zm[2, 1] -= h
        assert g["dZ"][2, 1] == pytest.approx((obj(sc0, v0, Zp) - obj(sc0, v0, Zm)) / (2 * h), rel=1e-5, abs=2e-6)


def test_kat7_online_hyper_gradient_with_extra_kl_vs_finite_differences():
    rng = np.random.default_rng(1)
    N = 160
    X = rng.random((N, 2)) * np.array([4.0, 2.0])
    f = np.sin(2 * X[:, 0]) + 0.5 * np.cos(3 * X[:, 1])
    y = (f + 0.2 * rng.standard_normal(N) > 0).astype(int)
    M = R.OnlineSVGP(R.Kernel("sqexponential", np.array([1.5, 1.0]), 1.2), R.LogisticLikelihood(), R.OIPS(0.7))
    M.train(X[:80], y[:80], 3)
    M.train(X[80:], y[80:], 2)
    yt, Xb = R.treat_labels(y[80:], M.likelihood), X[80:]
    g, gp = M.hyper_gradient(Xb, yt, 0), M.latents[0]
    assert gp["Za"] is not None
    sc0, v0, Z0, h = np.array(gp["kernel"].scale, float), gp["kernel"].sigma2, gp["Z"].copy(), 1e-6
    obj = lambda sc, v, Z: M.hyper_objective(Xb, yt, 0, sc, v, Z)
    assert g["dvariance"] == pytest.approx((obj(sc0, v0 + h, Z0) - obj(sc0, v0 - h, Z0)) / (2 * h), rel=1e-5, abs=1e-6)
    for d in range(2):
        e = np.zeros(2)
        e[d] = h
        assert g["dscale"][d] == pytest.approx((obj(sc0 + e, v0, Z0) - obj(sc0 - e, v0, Z0)) / (2 * h), rel=1e-5, abs=1e-6)
    for a, d in [(0, 0), (3, 1), (len(Z0) - 1, 0)]:
        Zp, Zm = Z0.copy(), Z0.copy()
        Zp[a, d] += h
        Zm[a, d] -= h
        assert g["dZ"][a, d] == pytest.approx((obj(sc0, v0, Zp) - obj(sc0, v0, Zm)) / (2 * h), rel=1e-5, abs=2e-6)
