"""The oracle against an INDEPENDENT third-party implementation (scikit-learn 1.7, `sklearn.gaussian_process`) at the two
boundaries where the reference itself relies on third parties that are absent here (KernelFunctions.jl, LAPACK):

* kernel definitions -- `SqExponentialKernel`, `Matern32Kernel`, `Matern52Kernel`, `ExponentialKernel` composed with
  `ScaleTransform(s)` / `ARDTransform(v)` and a variance factor (call sites latentgp.jl:202,206,210,212) against sklearn's `RBF`,
  `Matern(nu = 1.5 / 2.5 / 0.5)` with `length_scale = 1 / s`, times `ConstantKernel`;
* the Gaussian-likelihood path end to end -- an SVGP whose inducing points ARE the data (Z = X, m = N), full-batch AnalyticVI: one
  CAVI step lands on the optimal q(u) (KAT-1), and `predict_f` / `proba_y` are then exact GP regression (KAT-2), which
  `GaussianProcessRegressor` computes independently (Cholesky of K + sigma^2 I, Rasmussen & Williams alg. 2.1).

This does not pin the oracle to the Julia package (nothing here can), but it removes "both sides were written by the same hand"
for the kernel functions and for the Gaussian posterior / predictive algebra.  CPU only."""
import numpy as np
import pytest

from oracle import agp_ref as R

sk = pytest.importorskip("sklearn.gaussian_process")
from sklearn.gaussian_process.kernels import RBF, ConstantKernel, Matern  # noqa: E402


def _sk_kernel(kind, ell, var):
    base = {"sqexponential": lambda: RBF(length_scale=ell), "matern52": lambda: Matern(length_scale=ell, nu=2.5),
            "matern32": lambda: Matern(length_scale=ell, nu=1.5), "exponential": lambda: Matern(length_scale=ell, nu=0.5)}[kind]()
    return ConstantKernel(var, constant_value_bounds="fixed") * base


@pytest.mark.parametrize("kind", ["sqexponential", "matern52", "matern32", "exponential"])
@pytest.mark.parametrize("ard", [False, True])
def test_kernel_definitions_match_sklearn(kind, ard):
    rng = np.random.default_rng(5)
    D = 4
    X, Y = rng.random((40, D)), rng.random((25, D))
    scale = rng.uniform(0.5, 3.0, D) if ard else 1.7
    k = R.Kernel(kind, scale, 1.3)
    ks = _sk_kernel(kind, 1.0 / np.asarray(scale, dtype=np.float64), 1.3)
    assert np.max(np.abs(k.matrix(X, Y) - ks(X, Y))) < 1e-13
    assert np.max(np.abs(k.matrix(X) - ks(X))) < 1e-13
    k.fast = True  # the GEMM form the CPU baseline uses
    assert np.max(np.abs(k.matrix(X, Y) - ks(X, Y))) < 1e-12
    assert np.allclose(k.diag(X), np.diag(ks(X)), rtol=0, atol=1e-14)


@pytest.mark.parametrize("kind", ["sqexponential", "matern52"])
def test_gaussian_svgp_with_Z_equal_X_is_sklearn_gp_regression(kind):
    rng = np.random.default_rng(11)
    N, D, noise = 60, 2, 0.05
    X = rng.random((N, D))
    y = np.sin(4 * X[:, 0]) + X[:, 1] ** 2 + np.sqrt(noise) * rng.standard_normal(N)
    Xt = rng.random((30, D))
    ell, var, jit = 0.4, 1.5, 1e-10  # (a jitter far below the noise: the reference's 1e-4 would show up at 1e-4)
    m = R.SVGP(R.Kernel(kind, 1.0 / ell, var), R.GaussianLikelihood(noise), X.copy(), stochastic=False, jitter=jit)
    m.train(X, y, 2)  # one step reaches the optimum, the second is a fixed point (analyticVI.jl:143-180 with gaussian.jl:74-80)
    mu, v = m.predict_f(Xt, cov=True)
    gpr = sk.GaussianProcessRegressor(kernel=_sk_kernel(kind, ell, var), alpha=noise, optimizer=None).fit(X, y)
    mu_sk, sd_sk = gpr.predict(Xt, return_std=True)
    assert np.max(np.abs(mu[0] - mu_sk)) < 1e-6 * max(1.0, np.max(np.abs(mu_sk)))
    assert np.max(np.abs(v[0] - sd_sk ** 2)) < 1e-6
    # proba_y of the Gaussian likelihood = (mean, latent variance + noise)  gaussian.jl:41-45
    pm, pv = m.proba_y(Xt)
    assert np.max(np.abs(pm - mu_sk)) < 1e-6 * max(1.0, np.max(np.abs(mu_sk))) and np.max(np.abs(pv - (sd_sk ** 2 + noise))) < 1e-6
    # the collapsed bound at the optimum is the exact log marginal likelihood when Z = X (Titsias 2009, eq. 9 with Q = K):
    # ELBO(q*) = log N(y | 0, K + sigma^2 I)
    elbo = m.elbo(R.treat_labels(y, m.likelihood))
    assert abs(elbo - gpr.log_marginal_likelihood_value_) < 1e-5 * abs(gpr.log_marginal_likelihood_value_)


# ---- the KL divergences of the ELBO (src/functions/KLdivergences.jl) against torch.distributions' closed forms -------------------
def test_gaussian_kl_matches_torch_distributions():
    torch = pytest.importorskip("torch")
    from torch.distributions import MultivariateNormal, kl_divergence

    rng = np.random.default_rng(3)
    m = 12
    A, B = rng.standard_normal((m, m)), rng.standard_normal((m, m))
    Sigma, K = A @ A.T + 0.5 * np.eye(m), B @ B.T + 0.5 * np.eye(m)
    mu, mu0 = rng.standard_normal(m), rng.standard_normal(m)
    t = lambda a: torch.as_tensor(a, dtype=torch.float64)
    ref = kl_divergence(MultivariateNormal(t(mu), covariance_matrix=t(Sigma)), MultivariateNormal(t(mu0), covariance_matrix=t(K))).item()
    assert R.gaussian_kl(mu, mu0, Sigma, np.linalg.cholesky(K)) == pytest.approx(ref, rel=1e-11)  # KLdivergences.jl:11-18


def test_gamma_and_poisson_kl_match_torch_distributions():
    torch = pytest.importorskip("torch")
    from torch.distributions import Gamma, Poisson, kl_divergence

    rng = np.random.default_rng(4)
    t = lambda a: torch.as_tensor(a, dtype=torch.float64)
    # KLdivergences.jl:62-67 (the StudentT likelihood's inverse-gamma local variables: scalar shape, vector rate; a KL is invariant
    # under the reparametrisation omega -> 1 / omega, so the gamma closed form applies)
    alpha, alpha_p, beta_p = 2.5, 1.5, 0.7
    beta = rng.uniform(0.3, 3.0, 20)
    ref = kl_divergence(Gamma(t(alpha), t(beta)), Gamma(t(alpha_p), t(beta_p))).sum().item()
    assert R.gamma_kl(alpha, beta, alpha_p, beta_p) == pytest.approx(ref, rel=1e-11)
    # KLdivergences.jl:83-89 with psi = log(lambda0): the plain Poisson KL (the likelihoods pass E[log lambda0] for psi)
    lam, lam0 = rng.uniform(0.1, 5.0, 30), rng.uniform(0.1, 5.0, 30)
    ref = kl_divergence(Poisson(t(lam)), Poisson(t(lam0))).sum().item()
    assert R.poisson_kl(lam, lam0, np.log(lam0)) == pytest.approx(ref, rel=1e-11)


def test_polya_gamma_kl_against_the_series_of_its_laplace_transform():
    """KLdivergences.jl:96-98: KL(PG(b, c) || PG(b, 0)) = b log cosh(c / 2) - c^2 / 2 E[omega], E[omega] = b tanh(c / 2) / (2 c)
    (Polson, Scott & Windle 2013: the PG(b, c) density is the PG(b, 0) density tilted by exp(-c^2 omega / 2) cosh^b(c / 2)).
    Independent check of E[omega] from the defining series omega = (1 / 2 pi^2) sum_k g_k / ((k - 1/2)^2 + c^2 / 4 pi^2),
    g_k ~ Gamma(b, 1): E[omega] = (b / 2 pi^2) sum_k 1 / ((k - 1/2)^2 + c^2 / (4 pi^2))."""
    b = np.array([1.0, 1.0, 2.0, 3.5])
    c = np.array([0.3, 2.0, 1.1, 4.0])
    k = np.arange(1, 2_000_001, dtype=np.float64)[None, :]
    series = (b / (2 * np.pi ** 2)) * np.sum(1.0 / ((k - 0.5) ** 2 + (c[:, None] ** 2) / (4 * np.pi ** 2)), axis=1)
    theta = b * np.tanh(c / 2) / (2 * c)
    assert np.allclose(series, theta, rtol=1e-6)  # (the tail of the series beyond 2e6 terms is ~ b / (4 pi^2 1e6))
    expect = float(np.sum(b * np.log(np.cosh(c / 2)) - c * c / 2 * theta))
    assert R.polya_gamma_kl(b, c, theta) == pytest.approx(expect, rel=1e-12)


# ---------------------------------------------------------------------------------------------------------------------------------
# Independent pins for the NON-Gaussian path (round 4).  The augmented-variable CAVI of the reference is coordinate ascent on a
# bound whose optimum over the local variables is known in closed form from the original papers -- none of the formulas below is
# taken from the reference or from the oracle: the collapsed bound is written down from the literature, maximised directly over
# q(u) = N(mu, Sigma) with a generic optimiser (scipy L-BFGS on a torch-autograd objective), and the maximiser is compared with the
# fixed point the oracle's update equations (analyticVI.jl:143-246 + the likelihood's local_updates!) converge to.
#   * logistic  (src/likelihood/logistic.jl:39-92): Polya-Gamma augmentation == the Jaakkola-Jordan bound
#         log sigma(y f) >= log sigma(xi) + (y f - xi) / 2 - lambda(xi) (f^2 - xi^2),  lambda(xi) = tanh(xi / 2) / (4 xi),
#     tight in xi at xi^2 = E f^2, where it collapses to  log sigma(c) + (y E f - c) / 2,  c = sqrt(E f^2)   (Jaakkola & Jordan 2000;
#     Wenzel et al. 2019, eq. 9)
#   * Student-t (src/likelihood/studentt.jl:68-127): scale mixture y | f, w ~ N(f, w), w ~ IG(nu / 2, nu sigma^2 / 2); integrating
#     the optimal q(w) out of the ELBO leaves  -alpha log c_i + const,  alpha = (nu + 1) / 2,  c_i = (E (y_i - f_i)^2 + nu sigma^2) / 2
#     (the EM bound of the scale-mixture model)
#   * logistic-softmax (src/likelihood/logisticsoftmax.jl:55-140): no closed-form collapsed bound; pinned by two properties that tie
#     the restated UPDATE equations to the separately restated ELBO terms -- every full-batch CAVI sweep increases the ELBO, and at
#     the fixed point no perturbation of q(u) (local variables re-converged) increases it.
def _sparse_pieces(kern, X, Z, jitter):
    m = len(Z)
    K = kern.matrix(Z) + jitter * np.eye(m)            # latentgp.jl:205-207
    Knm = kern.matrix(X, Z)
    kappa = np.linalg.solve(K, Knm.T).T                # latentgp.jl:209-211
    Kt = kern.diag(X) + jitter - np.sum(kappa * Knm, axis=1)  # :212
    return K, kappa, Kt


def _maximise_collapsed_bound(point_term, K, kappa, Kt, y, mu0, L0):
    """argmax over (mu, Sigma = L L') of  sum_i point_term(E f_i, Var f_i, y_i) - KL(N(mu, Sigma) || N(0, K))  by L-BFGS"""
    import torch
    from scipy.optimize import minimize

    m = len(K)
    Kt_, kap_, K_, y_ = (torch.tensor(a, dtype=torch.float64) for a in (Kt, kappa, K, y))
    Kinv = torch.linalg.inv(K_)
    logdetK = torch.logdet(K_)
    tril = np.tril_indices(m)

    def unpack(p):
        mu = p[:m]
        L = torch.zeros((m, m), dtype=torch.float64)
        L[tril[0], tril[1]] = p[m:]
        d = torch.diagonal(L)
        L = L - torch.diag(d) + torch.diag(torch.exp(d))  # positive diagonal
        return mu, L

    def neg(p_np):
        p = torch.tensor(p_np, dtype=torch.float64, requires_grad=True)
        mu, L = unpack(p)
        Sig = L @ L.T
        mf = kap_ @ mu
        vf = Kt_ + torch.sum((kap_ @ L) ** 2, dim=1)
        kl = 0.5 * (torch.trace(Kinv @ Sig) + mu @ Kinv @ mu - m + logdetK - 2.0 * torch.sum(torch.log(torch.diagonal(L))))
        val = -(torch.sum(point_term(mf, vf, y_)) - kl)
        val.backward()
        return float(val.detach()), p.grad.numpy().copy()

    p0 = np.concatenate([mu0, np.where(np.eye(m, dtype=bool), np.log(np.abs(L0) + 1e-300), L0)[tril]])
    res = minimize(neg, p0, jac=True, method="L-BFGS-B", options=dict(maxiter=20000, ftol=1e-16, gtol=1e-10, maxcor=50))
    with torch.no_grad():
        mu, L = unpack(torch.tensor(res.x, dtype=torch.float64))
        return mu.numpy(), (L @ L.T).numpy(), -res.fun


def _toy_sparse(rng, N=60, D=2, m=7):
    X = rng.random((N, D))
    f = 2.0 * np.sin(4 * X[:, 0]) + 1.5 * X[:, 1] - 1.0
    Z = X[rng.permutation(N)[:m]].copy()
    return X, f, Z


def test_logistic_cavi_fixed_point_maximises_the_jaakkola_jordan_bound():
    import torch

    rng = np.random.default_rng(31)
    X, f, Z = _toy_sparse(rng)
    y = np.where(f + 0.3 * rng.standard_normal(len(f)) > 0, 1.0, -1.0)
    kern = R.Kernel("sqexponential", 2.0, 1.5)
    mdl = R.SVGP(kern, R.LogisticLikelihood(), Z, stochastic=False)
    mdl.train(X, y, 400, labels_treated=True)
    g = mdl.latents[0]
    K, kappa, Kt = _sparse_pieces(kern, X, Z, mdl.jitter)

    def jj(mf, vf, yy):  # the Jaakkola-Jordan bound at its optimal xi^2 = E f^2
        c = torch.sqrt(mf ** 2 + vf)
        return torch.nn.functional.logsigmoid(c) + 0.5 * (yy * mf - c)

    mu, Sig, val = _maximise_collapsed_bound(jj, K, kappa, Kt, y, 0.9 * g.mu, np.linalg.cholesky(1.1 * g.Sigma))
    assert np.max(np.abs(mu - g.mu)) < 2e-6 * np.max(np.abs(g.mu))
    assert np.max(np.abs(Sig - g.Sigma)) < 2e-6 * np.max(np.abs(g.Sigma))
    # ... and the oracle's ELBO at its fixed point IS that bound (the augmented terms cancel at the optimal local variables; corrected
    # mode, analyticVI.jl:255-274 with logistic.jl:70-92) -- up to a constant: the reference writes -N log(2) / 2 (logistic.jl:78)
    # where the Polya-Gamma identity sigma(z) = exp(z / 2) / (2 cosh(z / 2)) gives -N log 2.  Mirrored by the oracle and the device
    # (it moves no optimum and no hyper-gradient); recorded here so that the offset is a known quantity, not a surprise.
    assert mdl.elbo(y) - len(y) * np.log(2.0) / 2 == pytest.approx(val, rel=1e-8)


def test_studentt_cavi_fixed_point_maximises_the_scale_mixture_em_bound():
    import torch
    from scipy.special import gammaln

    rng = np.random.default_rng(32)
    X, f, Z = _toy_sparse(rng)
    nu, sig = 4.0, 0.7
    y = f + sig * rng.standard_t(nu, len(f))
    kern = R.Kernel("matern52", 1.5, 2.0)
    mdl = R.SVGP(kern, R.StudentTLikelihood(nu, sig), Z, stochastic=False)
    mdl.train(X, y, 600)
    g = mdl.latents[0]
    K, kappa, Kt = _sparse_pieces(kern, X, Z, mdl.jitter)
    alpha = (nu + 1) / 2

    def em(mf, vf, yy):  # log int N(y | f-moments, w) IG(w; nu/2, nu sig^2/2) dw with q(w) optimal = -alpha log c + const
        c = 0.5 * ((yy - mf) ** 2 + vf + nu * sig ** 2)
        const = 0.5 * nu * np.log(0.5 * nu * sig ** 2) - gammaln(0.5 * nu) - 0.5 * np.log(2 * np.pi) + gammaln(alpha)
        return -alpha * torch.log(c) + const

    mu, Sig, val = _maximise_collapsed_bound(em, K, kappa, Kt, y, 0.9 * g.mu, np.linalg.cholesky(1.1 * g.Sigma))
    assert np.max(np.abs(mu - g.mu)) < 2e-6 * np.max(np.abs(g.mu))
    assert np.max(np.abs(Sig - g.Sigma)) < 2e-6 * np.max(np.abs(g.Sigma))


def test_logisticsoftmax_updates_and_elbo_agree():
    rng = np.random.default_rng(33)
    X, f, Z = _toy_sparse(rng, N=90, m=8)
    y = 1 + np.digitize(f + 0.2 * rng.standard_normal(len(f)), np.quantile(f, [0.33, 0.66]))
    kern = R.Kernel("sqexponential", 2.0, 1.0)
    mdl = R.SVGP(kern, R.LogisticSoftMaxLikelihood(3), Z, stochastic=False)
    yt = R.treat_labels(y, mdl.likelihood)
    elbos = []
    mdl.train(X, yt, 150, labels_treated=True, callback=lambda M, it, xb, yb: elbos.append(M.elbo(yb)))
    e = np.array(elbos)
    assert np.all(np.diff(e) > -1e-9 * np.abs(e[1:])), "a full-batch CAVI sweep decreased the ELBO"
    assert abs(e[-1] - e[-2]) < 1e-7 * abs(e[-1])  # converged
    # local maximum in q(u): perturb (mu, Sigma) of every latent, let the local variables re-converge at the perturbed q(u)
    # (they have a unique fixed point given q(u)), compare the ELBO
    import copy

    for trial in range(6):
        p = copy.deepcopy(mdl)
        for gp in p.latents:
            d = 1e-2 * rng.standard_normal(len(gp.mu))
            S = gp.Sigma + 1e-2 * np.outer(d, d) * np.trace(gp.Sigma)
            mu = gp.mu + d * np.max(np.abs(gp.mu))
            gp.eta2 = -0.5 * np.linalg.inv(S)
            gp.eta1 = -2.0 * gp.eta2 @ mu
            gp.mu, gp.Sigma = mu, S
        for _ in range(300):  # one call = one (gamma, alpha) round of logisticsoftmax.jl:55-79
            p.local_vars = R.local_updates(p.local_vars, p.likelihood, yt, p.mean_f(), p.var_f())
        assert p.elbo(yt) <= e[-1] + 1e-9 * abs(e[-1])
    # proba_y of the multi-class likelihoods is the link at the posterior MEAN (multiclass.jl:96-118 ignores the variance)
    Xt = rng.random((20, X.shape[1]))
    pr = np.asarray(mdl.proba_y(Xt))                    # (n_test, n_class)
    mus = np.stack(mdl.predict_f(Xt, cov=False), axis=1)
    s = 1.0 / (1.0 + np.exp(-mus))
    assert np.allclose(pr, s / s.sum(axis=1, keepdims=True), atol=1e-12) and np.allclose(pr.sum(axis=1), 1.0)


def test_logisticsoftmax_augmentation_identities():
    """The three identities the augmented model rests on (Galy-Fajou et al. 2019), by quadrature / truncated sums -- nothing of the
    reference or the oracle in here: sigma(f_k) / sum_j sigma(f_j) = int sum_n sigma(f_k) prod_j Po(n_j | lambda) sigma(-f_j)^n_j."""
    from scipy.integrate import quad
    from scipy.special import expit, gammaln as lg

    rng = np.random.default_rng(5)
    for _ in range(3):
        f = 1.5 * rng.standard_normal(3)
        s = expit(f)
        for k in range(3):
            direct = s[k] / s.sum()
            # sum over n_j of Po(n_j | lam) sigma(-f_j)^n_j = exp(-lam sigma(f_j)): checked by a truncated sum, then integrated over lam
            def integrand(lam):
                val = s[k]
                for j in range(3):
                    n = np.arange(0, 80)
                    val *= np.sum(np.exp(n * np.log(max(lam, 1e-300)) - lam - lg(n + 1)) * expit(-f[j]) ** n)
                return val
            aug, _ = quad(integrand, 0.0, 200.0, limit=400)
            assert aug == pytest.approx(direct, rel=1e-7)


def test_logisticsoftmax_fixed_point_is_stationary_for_the_augmented_bound_of_the_paper():
    """the independent pin of the LogisticSoftMax path.  The augmented bound (Polya-Gamma + Poisson + Gamma
    augmentation) is written from the paper as a torch-autograd objective over ALL variational parameters -- (mu_k, Sigma_k), gamma,
    alpha, beta, c -- in tests/_torch_elbo.py.  The fixed point of the oracle's update equations (logisticsoftmax.jl:55-79 +
    analyticVI.jl:143-246) has to be a stationary point of it, beta = K (which the reference never updates) included, and the
    oracle's ELBO at that point IS the bound up to the two constants the reference carries (-N K log 2 from `length(y)` of the
    one-hot view and log beta_1 counted once: SURVEY Q16)."""
    import os
    import sys

    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import _torch_elbo as TE

    rng = np.random.default_rng(33)
    X, f, Z = _toy_sparse(rng, N=90, m=8)
    Kc = 3
    y = 1 + np.digitize(f + 0.2 * rng.standard_normal(len(f)), np.quantile(f, [0.33, 0.66]))
    kern = R.Kernel("sqexponential", 2.0, 1.0)
    mdl = R.SVGP(kern, R.LogisticSoftMaxLikelihood(Kc), Z, stochastic=False)
    yt = R.treat_labels(y, mdl.likelihood)
    mdl.train(X, yt, 4000, labels_treated=True)
    Kmat, kappa, Kt = _sparse_pieces(kern, X, Z, mdl.jitter)
    lv = mdl.local_vars
    gamma = np.stack(lv["gamma"], axis=1)
    c = np.stack(lv["c"], axis=1)
    Y = yt.astype(np.float64)
    val, g = TE.lsm_bound_and_gradients(Y, kappa, Kt, [Kmat] * Kc, [gp.mu for gp in mdl.latents], [gp.Sigma for gp in mdl.latents],
                                        gamma, lv["alpha"], lv["beta"], c)
    N = len(y)
    assert np.all(lv["beta"] == Kc)
    # stationarity: every gradient vanishes against the scale of the terms it is the sum of (the bound is O(N) = 1e2, single terms O(1))
    assert max(np.max(np.abs(x)) for x in g["mu"]) < 1e-6
    assert max(np.max(np.abs(x)) for x in g["L"]) < 1e-6
    assert np.max(np.abs(g["gamma"])) < 1e-6 and np.max(np.abs(g["alpha"])) < 1e-6 and np.max(np.abs(g["c"])) < 1e-6
    assert np.max(np.abs(g["beta"])) < 1e-6  # beta = K is where the bound's own optimum sits: not updating it loses nothing
    # ... and it is not a trivial zero: away from the fixed point the same gradients are O(1)
    _, g2 = TE.lsm_bound_and_gradients(Y, kappa, Kt, [Kmat] * Kc, [1.2 * gp.mu for gp in mdl.latents], [gp.Sigma for gp in mdl.latents],
                                       1.1 * gamma, lv["alpha"], lv["beta"], c)
    assert max(np.max(np.abs(x)) for x in g2["mu"]) > 1e-2 and np.max(np.abs(g2["gamma"])) > 1e-2
    # the value: reference constants accounted for
    assert mdl.elbo(yt) == pytest.approx(val - N * Kc * np.log(2.0) + (N - 1) * np.log(Kc), rel=1e-9)


# ---------------------------------------------------------------------------------------------------------------------------------
# Round 6: pins for the likelihoods of SURVEY 8f-2 and for the multi-output mixing that do not pass through the
# oracle's own formulas.
#   * BayesianSVM (bayesiansvm.jl:43-67): the pseudo-likelihood exp(-2 max(1 - y f, 0)) as the location-scale mixture of Polson &
#     Scott (2011) / Wenzel et al. (2017): L(y | f) = int (2 pi lam)^-1/2 exp(-(1 + lam - y f)^2 / (2 lam)) dlam.  With the optimal
#     q(lam) = GIG(1/2, 1, c), c = E(1 - y f)^2, the integral int (2 pi lam)^-1/2 exp(-(c / lam + lam) / 2) dlam = exp(-sqrt c)
#     collapses the bound to  -sqrt(c) - (1 - y E f)  per point.
#   * NegBinomial (negativebinomial.jl:69-99): sigma(f)^y sigma(-f)^r with a Polya-Gamma variable of shape b = y + r.  The JJ / PG
#     collapsed bound is  (y - r) E f / 2 + b [log sigma(c) - c / 2], c = sqrt(E f^2)  (its stationarity gives grad_E_Sigma =
#     b tanh(c/2) / (4 c) = E[omega] / 2).  The reference writes theta = (r + y) tanh(c/2) / c and grad_E_Sigma = theta / 2, i.e. TWICE
#     that (negativebinomial.jl:78,97-99 -- its comment says "E[omega]"; the logistic model's theta = tanh(c/2) / (2c) IS E[omega]).
#     Its updates are therefore the exact coordinate ascent of the bound with the Polya-Gamma part DOUBLED,
#     (y - r) E f / 2 + 2 b [log sigma(c) - c / 2]: the test shows that the restated fixed point maximises that functional to 2e-6
#     and is measurably NOT the maximiser of the literature's bound.  Mirrored, like the other quirks of SURVEY Appendix A; the same
#     factor sits in the Poisson model's theta (poisson.jl:75), whose gamma and lambda updates follow the undoubled bound, so that
#     model's fixed point maximises no single functional and is pinned by its identities only (test_oracle_kat.py).
#   * multi-output mixing (single_and_multi_output_utils.jl:24-118): the data term of every admissible task likelihood is
#     E log p(y_t | f_t, omega) = g1 . E f_t - g2 . E f_t^2 + const with f_t = sum_q A_tq f_q; torch.autograd of that expression
#     in A against the restated update_A! gradient.
def test_bayesiansvm_cavi_fixed_point_maximises_the_collapsed_scale_mixture_bound():
    import torch

    rng = np.random.default_rng(61)
    X, f, Z = _toy_sparse(rng)
    y = np.where(f + 0.3 * rng.standard_normal(len(f)) > 0, 1.0, -1.0)
    kern = R.Kernel("sqexponential", 2.0, 1.5)
    mdl = R.SVGP(kern, R.BayesianSVM(), Z, stochastic=False)
    mdl.train(X, y, 800, labels_treated=True)
    g = mdl.latents[0]
    K, kappa, Kt = _sparse_pieces(kern, X, Z, mdl.jitter)

    def svm(mf, vf, yy):
        return -torch.sqrt((1.0 - yy * mf) ** 2 + vf) - (1.0 - yy * mf)

    mu, Sig, val = _maximise_collapsed_bound(svm, K, kappa, Kt, y, 0.9 * g.mu, np.linalg.cholesky(1.1 * g.Sigma))
    assert np.max(np.abs(mu - g.mu)) < 2e-6 * np.max(np.abs(g.mu))
    assert np.max(np.abs(Sig - g.Sigma)) < 2e-6 * np.max(np.abs(g.Sigma))
    # (the reference's ELBO for this likelihood is NOT that bound's value: it subtracts the GIG entropy as if it were a KL term,
    #  bayesiansvm.jl:85-92 -- mirrored by the oracle and the device; the fixed point, which the updates alone decide, is what is pinned)
    assert np.isfinite(val) and np.isfinite(mdl.elbo(y))


def test_negbinomial_fixed_point_maximises_the_bound_with_a_doubled_polya_gamma_part():
    import torch

    rng = np.random.default_rng(62)
    X, f, Z = _toy_sparse(rng)
    r = 4.0
    y = rng.negative_binomial(int(r), 1.0 / (1.0 + np.exp(0.7 * f))).astype(np.float64)
    kern = R.Kernel("sqexponential", 2.0, 1.2)
    mdl = R.SVGP(kern, R.NegBinomialLikelihood(r), Z, stochastic=False)
    mdl.train(X, y, 800, labels_treated=True)
    g = mdl.latents[0]
    K, kappa, Kt = _sparse_pieces(kern, X, Z, mdl.jitter)

    def bound(scale):
        def h(mf, vf, yy):
            c = torch.sqrt(mf ** 2 + vf)
            return 0.5 * (yy - r) * mf + scale * (yy + r) * (torch.nn.functional.logsigmoid(c) - 0.5 * c)
        return h

    mu2, Sig2, _ = _maximise_collapsed_bound(bound(2.0), K, kappa, Kt, y, 0.9 * g.mu, np.linalg.cholesky(1.1 * g.Sigma))
    assert np.max(np.abs(mu2 - g.mu)) < 2e-6 * np.max(np.abs(g.mu))
    assert np.max(np.abs(Sig2 - g.Sigma)) < 2e-6 * np.max(np.abs(g.Sigma))
    mu1, Sig1, _ = _maximise_collapsed_bound(bound(1.0), K, kappa, Kt, y, 0.9 * g.mu, np.linalg.cholesky(1.1 * g.Sigma))
    # ... and not the literature's: the posterior variances of the reference's fixed point are too small by a visible margin
    assert np.max(np.abs(np.diag(Sig1) - np.diag(g.Sigma))) > 0.05 * np.max(np.diag(g.Sigma))
    assert np.all(np.diag(Sig1) > np.diag(g.Sigma))


def _mo_toy(rng, liks):
    N, D, m, Q = 80, 2, 6, 3
    X = rng.random((N, D))
    fs = [np.sin(4 * X[:, 0]), X[:, 1] - 0.5, np.cos(3 * X[:, 0] * X[:, 1])]
    ys = []
    for t, l in enumerate(liks):
        ys.append(fs[t] + 0.1 * rng.standard_normal(N) if l.name in ("gaussian", "studentt", "laplace")
                  else np.sign(fs[t] + 0.1 * rng.standard_normal(N)))
    A = rng.standard_normal((len(liks), Q))
    A /= np.linalg.norm(A, axis=1, keepdims=True)
    Zs = [X[rng.permutation(N)[:m]].copy() for _ in range(Q)]
    return X, ys, A, Zs


def test_multioutput_mixing_gradient_is_what_autograd_gives():
    """update_A! (single_and_multi_output_utils.jl:87-118) against torch.autograd of the data term written from the likelihoods'
    expec_loglikelihood with the mixed moments E f_t = sum_q A_tq E f_q, Var f_t = sum_q A_tq^2 Var f_q (:24-45), local variables
    held fixed -- for a Gaussian, a logistic (corrected ELBO) and a Student-t task."""
    import torch

    rng = np.random.default_rng(63)
    liks = [R.GaussianLikelihood(0.2), R.LogisticLikelihood(), R.StudentTLikelihood(4.0, 0.8)]
    X, ys, A, Zs = _mo_toy(rng, liks)
    eta = 0.05
    mdl = R.MOSVGP(R.Kernel("sqexponential", 3.0, 1.0), liks, Zs, A.copy(), stochastic=False, A_opt=R.Adam(eta))
    mdl.train(X, ys, 3)  # some posterior, local variables of the last step in place
    A0 = mdl.A.copy()
    mu_q, var_q = mdl.lat_mean_var()
    Mq = torch.tensor(np.stack(mu_q), dtype=torch.float64)
    Vq = torch.tensor(np.stack(var_q), dtype=torch.float64)
    At = torch.tensor(A0, dtype=torch.float64, requires_grad=True)
    total = 0.0
    for t, lik in enumerate(liks):
        mf = At[t] @ Mq
        vf = (At[t] ** 2) @ Vq
        yt = torch.tensor(ys[t], dtype=torch.float64)
        lv = mdl.local_vars[t]
        if lik.name == "gaussian":     # gaussian.jl:82-93
            total = total - 0.5 * (((yt - mf) ** 2).sum() + vf.sum()) / lik.sigma2
        elif lik.name == "logistic":   # logistic.jl:73-84 (corrected: theta . mu^2)
            th = torch.tensor(lv["theta"], dtype=torch.float64)
            total = total + 0.5 * ((mf * yt).sum() - (th * vf).sum() - (th * mf * mf).sum())
        else:                           # studentt.jl:103-119
            th = torch.tensor(lv["theta"], dtype=torch.float64)
            total = total - 0.5 * (th * (vf + mf * mf - 2.0 * mf * yt + yt * yt)).sum()
    total.backward()
    G = At.grad.numpy()
    # the oracle's step: ADAM's first step from zero moments moves each entry by eta * sign(grad) (bias-corrected m / sqrt(v) = +-1), then
    # the rows are normalised -- so compare the gradient through the restated formula directly
    for t, lik in enumerate(liks):
        gmu = R.grad_E_mu(lik, ys[t], mdl.local_vars[t])[0]
        gS = R.grad_E_Sigma(lik, ys[t], mdl.local_vars[t])[0]
        for q in range(mdl.Q):
            others = sum(A0[t, qq] * mu_q[qq] for qq in range(mdl.Q) if qq != q)
            dA = np.dot(gmu, mu_q[q]) - 2.0 * np.dot(gS, mu_q[q] * others) - 2.0 * A0[t, q] * np.dot(gS, mu_q[q] ** 2 + var_q[q])
            assert dA == pytest.approx(G[t, q], rel=1e-10, abs=1e-12)
    # and update_A! itself takes exactly that gradient: one more call moves A by ADAM(dA) and renormalises
    import copy

    st = copy.deepcopy(mdl.A_state)
    mdl.update_A(ys)
    opt = R.Adam(eta)
    for t in range(len(liks)):
        _, delta = opt.apply(st[t], G[t])
        row = A0[t] + delta
        assert np.allclose(mdl.A[t], row / np.linalg.norm(row), rtol=1e-12, atol=1e-14)
