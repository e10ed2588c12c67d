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
