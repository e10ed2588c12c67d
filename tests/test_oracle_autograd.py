"""AD semantics of the hyper-parameter step (VERDICT r04 item 8 ii).  The reference obtains the gradient of
ELBO(model, X, y, mu0, ks, Zs, state) from Zygote (autotuning.jl:96-98); the oracle and the device use a hand-derived reverse mode
(formula sheet).  tests/_torch_elbo.py restates that objective in torch fp64 from the reference's definitions, and
torch.autograd differentiates it: the hand-derived gradient has to agree -- kernel variance, per-dimension scales, inducing points --
at m = 64 and m = 512, in both ELBO modes (the `reference` mode differentiates logistic.jl:82's dot(theta, mu) as written)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from oracle import agp_ref as R  # noqa: E402

import _torch_elbo as TE  # noqa: E402


def _rel(a, b):
    return float(np.max(np.abs(np.asarray(a) - np.asarray(b))) / max(np.max(np.abs(np.asarray(b))), 1e-300))


def _setup(likname, kind, m, mode, seed=11):
    rng = np.random.default_rng(seed)
    N, D, B, iters = 1500, 3, 700 if m >= 512 else 300, 3
    X = rng.random((N, D))
    f = 1.5 * np.sin(4 * X[:, 0]) + X[:, 1] * X[:, 2] - 0.4
    if likname == "logistic":
        lik, y = R.LogisticLikelihood(), (f + 0.3 * rng.standard_normal(N) > 0).astype(int)
    elif likname == "gaussian":
        lik, y = R.GaussianLikelihood(0.05), f + 0.2 * rng.standard_normal(N)
    else:
        lik, y = R.StudentTLikelihood(3.0, 0.5), f + 0.2 * rng.standard_t(3, N)
    # (rough enough for a K_ZZ whose inverse does not amplify fp64 rounding beyond ~1e-9: 64 / 512 of 1500 points in [0, 1]^3 as Z;
    #  two different solvers -- Cholesky in the oracle, LU in torch.linalg.inv -- are being compared)
    sc = np.array([4.0, 5.0, 3.5]) * (1.0 if m <= 64 else 3.0)
    Z = X[rng.permutation(N)[:m]].copy()
    M = R.SVGP(R.Kernel(kind, sc, 1.3), lik, Z, stochastic=True, batchsize=B, elbo_mode=mode)
    yt = R.treat_labels(y, lik)
    idx = [rng.choice(N, B, replace=False) for _ in range(iters)]
    M.train(X, yt, iters, idx_stream=idx, labels_treated=True)
    xb, yb = X[idx[-1]], yt[idx[-1]]
    M.hp_updated = True
    M.compute_kernel_matrices(xb)
    return M, xb, yb, N / B


@pytest.mark.parametrize("likname,kind,mode", [("logistic", "sqexponential", "corrected"), ("logistic", "matern52", "reference"),
                                               ("gaussian", "matern32", "corrected"), ("studentt", "sqexponential", "corrected")])
@pytest.mark.parametrize("m", [64, 512])
def test_hand_derived_hyper_gradient_is_what_autograd_gives(likname, kind, mode, m):
    M, xb, yb, rho = _setup(likname, kind, m, mode)
    gp = M.latents[0]
    g = R.hyper_gradient(M, xb, yb, 0, rho)
    lik = (likname, M.likelihood.sigma2) if likname == "gaussian" else (likname,)
    local = {} if likname == "gaussian" else {"theta": M.local_vars["theta"]}
    dv, ds, dz, val = TE.autograd_hypergrad(kind, lik, xb, np.asarray(yb, dtype=np.float64), gp.Z, gp.kernel.scale, gp.kernel.sigma2,
                                            gp.mu, gp.Sigma, gp.mu0, local, rho, M.jitter, mode)
    assert abs(g["dvariance"] - dv) < 1e-8 * max(1.0, abs(dv))
    assert _rel(g["dscale"], ds) < 1e-8
    assert _rel(g["dZ"], dz) < 1e-8
    # the objective itself: the oracle's (ELBO.jl:15-21 restated there) and this file's agree where they hold the same terms
    if likname in ("logistic", "gaussian"):
        ref = R.hyper_objective(M, xb, yb, 0, gp.kernel.scale, gp.kernel.sigma2, gp.Z, rho)
        assert ref == pytest.approx(val, rel=1e-9)
