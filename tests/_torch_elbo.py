"""The hyper-parameter objective of the reference, restated in torch fp64 so that torch.autograd plays the part Zygote plays there.

The reference differentiates `ELBO(model, X, y, mu0, ks, Zs, state)` (src/functions/ELBO.jl:15-21 -> analyticVI.jl:255-274) with
Zygote (src/hyperparameter/autotuning.jl:96-98, zygote_rules.jl:1-8) with respect to the kernel objects and the inducing points:
the kernel matrices are recomputed from the candidate kernel / Z, the variational parameters (mu, Sigma), the prior mean and the
local variables of the state are constants.  Nothing in this file comes from oracle/agp_ref.py or from the device code: the kernel
functions are KernelFunctions.jl's definitions (SqExponentialKernel exp(-d^2 / 2), Matern32 (1 + sqrt3 d) exp(-sqrt3 d), Matern52
(1 + sqrt5 d + 5 d^2 / 3) exp(-sqrt5 d), `sigma2 * k o ScaleTransform / ARDTransform`), the sparse-GP pieces are
src/gpblocks/latentgp.jl:199-215, the Gaussian KL is src/functions/KLdivergences.jl:11-18, the data terms are the likelihoods'
expec_loglikelihood (gaussian.jl:82-93, logistic.jl:73-84, studentt.jl:103-119) with theta, c held fixed.
"""
import math

import numpy as np
import torch


def kernel_matrix(kind, A, B, scale, variance):
    a, b = A * scale, B * scale
    d2 = (a * a).sum(1)[:, None] + (b * b).sum(1)[None, :] - 2.0 * a @ b.T
    if kind == "sqexponential":
        return variance * torch.exp(-0.5 * d2.clamp_min(0.0))
    # (coincident points: the derivative of sqrt at 0 is infinite, the kernel's is not; below the floor the pair contributes nothing,
    #  which is the truth for a point against itself)
    d = torch.sqrt(d2.clamp_min(1e-30))
    if kind == "matern32":
        s = math.sqrt(3.0)
        return variance * (1.0 + s * d) * torch.exp(-s * d)
    if kind == "matern52":
        s = math.sqrt(5.0)
        return variance * (1.0 + s * d + 5.0 * d * d / 3.0) * torch.exp(-s * d)
    raise ValueError(kind)


def hyper_elbo(kind, lik, X, y, Z, scale, variance, mu, Sigma, mu0, local, rho, jitter, mode="corrected"):
    """ELBO.jl:15-21 as a function of (scale, variance, Z); everything else constant.  lik = (name, params)"""
    m = Z.shape[0]
    K = kernel_matrix(kind, Z, Z, scale, variance) + jitter * torch.eye(m, dtype=torch.float64)  # latentgp.jl:205-207
    Knm = kernel_matrix(kind, X, Z, scale, variance)
    Kinv = torch.linalg.inv(K)
    kappa = Knm @ Kinv                                                                             # :209-211
    Kt = variance + jitter - (kappa * Knm).sum(1)                                                   # :212 (kdiag of a stationary kernel)
    mf = kappa @ mu                                                                                 # :189
    vf = ((kappa @ Sigma) * kappa).sum(1) + Kt                                                      # :179
    name = lik[0]
    if name == "gaussian":
        s2 = lik[1]
        e = -0.5 * (len(y) * math.log(2.0 * math.pi * s2) + (((y - mf) ** 2).sum() + vf.sum()) / s2)
    elif name == "logistic":
        th = local["theta"]
        quad = (th * mf).sum() if mode == "reference" else (th * mf * mf).sum()   # logistic.jl:82 writes dot(theta, mu)
        e = -0.5 * len(y) * math.log(2.0) + 0.5 * ((mf * y).sum() - (th * vf).sum() - quad)
    elif name == "studentt":
        th = local["theta"]
        e = -0.5 * (th * (vf + mf * mf - 2.0 * mf * y + y * y)).sum()  # (+ terms in nu, sigma, c: constants here)
    else:
        raise ValueError(name)
    d = mu - mu0
    kl = 0.5 * (torch.trace(Kinv @ Sigma) + d @ Kinv @ d - m + torch.logdet(K) - torch.logdet(Sigma))
    return rho * e - kl


def autograd_hypergrad(kind, lik, X, y, Z, scale, variance, mu, Sigma, mu0, local, rho, jitter, mode="corrected"):
    """-> (d / d variance, d / d scale (per dimension), d / d Z) by reverse-mode AD of hyper_elbo"""
    t = lambda a: torch.tensor(np.asarray(a, dtype=np.float64))
    D = X.shape[1]
    sc = torch.tensor(np.broadcast_to(np.asarray(scale, dtype=np.float64), (D,)).copy(), requires_grad=True)
    var = torch.tensor(float(variance), dtype=torch.float64, requires_grad=True)
    Zt = t(Z).clone().requires_grad_(True)
    loc = {k: t(v) for k, v in local.items()}
    val = hyper_elbo(kind, lik, t(X), t(y), Zt, sc, var, t(mu), t(Sigma), t(mu0), loc, rho, jitter, mode)
    val.backward()
    return float(var.grad), sc.grad.numpy().copy(), Zt.grad.numpy().copy(), float(val.detach())


# ---------------------------------------------------------------------------------------------------------------------------------
# The augmented logistic-softmax bound, written down from the paper (Galy-Fajou, Wenzel, Donner, Opper: "Multi-Class Gaussian Process
# Classification Made Conjugate", 2019), NOT from src/likelihood/logisticsoftmax.jl or from the oracle:
#   p(y = k | f) = sigma(f_k) / sum_j sigma(f_j)
#   1 / sum_j sigma(f_j) = int_0^inf exp(-lambda sum_j sigma(f_j)) d lambda                       (lambda, improper flat prior)
#   exp(-lambda sigma(f_j)) = exp(-lambda) sum_n lambda^n / n! sigma(-f_j)^n                        (n_j ~ Po(lambda))
#   sigma(f)^y sigma(-f)^n = 2^-(y+n) exp((y - n) f / 2) int exp(-omega f^2 / 2) PG(omega | y + n, 0) d omega
# with q(lambda_i) = Ga(alpha_i, beta_i), q(n_ij) = Po(gamma_ij), q(omega_ij) = PG(y_ij + gamma_ij, c_ij), q(u_j) = N(mu_j, Sigma_j):
#   E log p(y, omega, n | f, lambda) = sum_ij -(y + gamma) log 2 + (y - gamma) E f / 2 - theta E f^2 / 2,  theta = E omega
#                                      + sum_ij gamma (psi(alpha_i) - log beta_i) - alpha_i / beta_i - E log n_ij!
#   - KL(q(omega) || PG(y + gamma, 0)) = -(y + gamma) log cosh(c / 2) + c^2 theta / 2
#   H[Po(gamma)] = gamma - gamma log gamma + E log n!          H[Ga(alpha, beta)] = alpha - log beta + lgamma(alpha) + (1 - alpha) psi(alpha)
# ---------------------------------------------------------------------------------------------------------------------------------
def lsm_augmented_bound(Y, kappa, Kt, Kmats, mus, Ls, gamma, alpha, beta, c):
    """Y (N x K one-hot), kappa (N x m), Kt (N), Kmats: list of K prior covariances; variational parameters: mus[k] (m), Ls[k] (Sigma_k =
    L L'), gamma (N x K), alpha (N), beta (N), c (N x K).  All torch fp64.  Returns the bound (a scalar)."""
    K = Y.shape[1]
    tot = torch.zeros((), dtype=torch.float64)
    psi = torch.digamma(alpha)
    for k in range(K):
        mf = kappa @ mus[k]
        vf = Kt + ((kappa @ Ls[k]) ** 2).sum(1)
        b = Y[:, k] + gamma[:, k]
        th = b * torch.tanh(c[:, k] / 2.0) / (2.0 * c[:, k])
        tot = tot + (-b * math.log(2.0) + (Y[:, k] - gamma[:, k]) * mf / 2.0 - th * (mf * mf + vf) / 2.0).sum()
        tot = tot - (b * torch.log(torch.cosh(c[:, k] / 2.0)) - c[:, k] ** 2 * th / 2.0).sum()
        g = gamma[:, k]
        tot = tot + (g * (psi - torch.log(beta)) - alpha / beta + g - g * torch.log(g)).sum()
        m = len(mus[k])
        Kinv = torch.linalg.inv(Kmats[k])
        Sig = Ls[k] @ Ls[k].T
        kl = 0.5 * (torch.trace(Kinv @ Sig) + mus[k] @ Kinv @ mus[k] - m + torch.logdet(Kmats[k])
                    - 2.0 * torch.log(torch.diagonal(Ls[k]).abs()).sum())
        tot = tot - kl
    tot = tot + (alpha - torch.log(beta) + torch.lgamma(alpha) + (1.0 - alpha) * psi).sum()
    return tot


def lsm_bound_and_gradients(Y, kappa, Kt, Kmats, mus, Sigmas, gamma, alpha, beta, c):
    """numpy in; -> (value, dict of gradients w.r.t. mu_k, L_k (lower Cholesky factor of Sigma_k), gamma, alpha, beta, c)"""
    t = lambda a: torch.tensor(np.asarray(a, dtype=np.float64))
    leaf = lambda a: t(a).clone().requires_grad_(True)
    mus_t = [leaf(mu) for mu in mus]
    Ls_t = [leaf(np.linalg.cholesky(S)) for S in Sigmas]
    g_t, a_t, b_t, c_t = leaf(gamma), leaf(alpha), leaf(beta), leaf(c)
    val = lsm_augmented_bound(t(Y), t(kappa), t(Kt), [t(Km) for Km in Kmats], mus_t, Ls_t, g_t, a_t, b_t, c_t)
    val.backward()
    grads = {"mu": [x.grad.numpy() for x in mus_t], "L": [np.tril(x.grad.numpy()) for x in Ls_t], "gamma": g_t.grad.numpy(),
             "alpha": a_t.grad.numpy(), "beta": b_t.grad.numpy(), "c": c_t.grad.numpy()}
    return float(val.detach()), grads
