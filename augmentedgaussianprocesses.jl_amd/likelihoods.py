"""Likelihood descriptors + label handling (host logic, no numerics).

Mirrors src/likelihood/{gaussian,logistic,classification,studentt,regression,logisticsoftmax,multiclass,laplace,
bayesiansvm,poisson,negativebinomial,event,heteroscedastic}.jl:
constructors, `implemented`, `n_latent`, `treat_labels!`, class mapping / one-hot.  The local updates, gradients,
ELBO terms and compute_proba run on the GPU (csrc/agp_cavi.h).
"""
from __future__ import annotations

import numpy as np

from . import capi


class AbstractLikelihood:
    kind = None
    n_latent = 1

    def lik_desc(self):
        raise NotImplementedError


class GaussianLikelihood(AbstractLikelihood):
    """GaussianLikelihood(σ²=1e-3; opt_noise=false)  src/likelihood/gaussian.jl:10-24.  opt_noise: True -> ADAM(0.05) (:18-21), an
    ADAM object, or False.  With it σ² is state: every local update takes one ADAM ascent step on log σ² before θ = 1/σ² is
    refreshed (:56-72); the value is mirrored back into `sigma2` when training ends."""

    kind = capi.LIK_GAUSSIAN

    def __init__(self, sigma2: float = 1e-3, opt_noise=False):
        if not sigma2 > 0:
            raise ValueError("σ² must be positive")
        self.sigma2 = float(sigma2)
        if isinstance(opt_noise, bool):
            self.noise_eta = 0.05 if opt_noise else 0.0
        else:
            eta = getattr(opt_noise, "eta", None)
            if (eta is None or tuple(getattr(opt_noise, "beta", (0.9, 0.999))) != (0.9, 0.999)
                    or float(getattr(opt_noise, "eps", 1e-8)) != 1e-8):
                # (the device kernel k_noise_finish carries Optimisers.ADAM's default moments and epsilon as constants)
                raise NotImplementedError("opt_noise takes ADAM(eta) with the default moments and epsilon")
            self.noise_eta = float(eta)

    @property
    def lam(self):  # the handle's likelihood-state word (what Poisson / Heteroscedastic call lambda)
        if not self.noise_eta:
            raise AttributeError("lam")
        return self.sigma2

    @lam.setter
    def lam(self, v):
        self.sigma2 = float(v)

    def lik_desc(self):
        return capi.LikDesc(self.kind, 1, self.sigma2, self.noise_eta)

    def __repr__(self):
        return f"Gaussian likelihood (σ² = {self.sigma2})"


class LogisticLikelihood(AbstractLikelihood):
    """LogisticLikelihood() -> BernoulliLikelihood(LogisticLink())  src/likelihood/logistic.jl:19."""

    kind = capi.LIK_LOGISTIC

    def lik_desc(self):
        return capi.LikDesc(self.kind, 1, 0.0, 0.0)

    def __repr__(self):
        return "Bernoulli Likelihood with Logistic Link"


class StudentTLikelihood(AbstractLikelihood):
    """StudentTLikelihood(ν, σ=1)  src/likelihood/studentt.jl:23-35."""

    kind = capi.LIK_STUDENTT

    def __init__(self, nu: float, sigma: float = 1.0):
        if not nu > 0.5:
            raise ValueError("ν should be greater than 0.5")  # studentt.jl:28
        self.nu = float(nu)
        self.sigma = float(sigma)
        self.alpha = (self.nu + 1.0) / 2.0

    def lik_desc(self):
        return capi.LikDesc(self.kind, 1, self.nu, self.sigma)

    def __repr__(self):
        return f"Student-t likelihood (ν={self.nu}, σ={self.sigma})"


class LogisticSoftMaxLikelihood(AbstractLikelihood):
    """LogisticSoftMaxLikelihood(num_class | labels)  src/likelihood/logisticsoftmax.jl:24, multiclass.jl:1-25."""

    kind = capi.LIK_LOGISTICSOFTMAX

    def __init__(self, x):
        if isinstance(x, (int, np.integer)):
            self.n_class = int(x)
            self.class_mapping = None
            self.ind_mapping = None
        else:
            labels = list(x)
            self.n_class = len(labels)
            self.class_mapping = labels
            self.ind_mapping = {v: i + 1 for i, v in enumerate(labels)}
        if self.n_class < 2:
            raise ValueError("need at least two classes")

    @property
    def n_latent(self):  # multiclass.jl:27
        return self.n_class

    def lik_desc(self):
        return capi.LikDesc(self.kind, self.n_class, 0.0, 0.0)

    def __repr__(self):
        return f"Multiclass Likelihood ({self.n_class} classes, Logistic-SoftMax Link )"


class LaplaceLikelihood(AbstractLikelihood):
    """LaplaceLikelihood(β=1.0)  src/likelihood/laplace.jl:17-30 (q(ω) = GIG(a = β⁻², b, p = 1/2))."""

    kind = capi.LIK_LAPLACE

    def __init__(self, beta: float = 1.0):
        if not beta > 0:
            raise ValueError("β must be positive")
        self.beta = float(beta)
        self.a = self.beta ** -2
        self.p = 0.5

    def lik_desc(self):
        return capi.LikDesc(self.kind, 1, self.beta, 0.0)

    def __repr__(self):
        return f"Laplace likelihood (β={self.beta})"


class BayesianSVM(AbstractLikelihood):
    """BayesianSVM() -> BernoulliLikelihood(SVMLink())  src/likelihood/bayesiansvm.jl:19-23."""

    kind = capi.LIK_BAYESIANSVM

    def lik_desc(self):
        return capi.LikDesc(self.kind, 1, 0.0, 0.0)

    def __repr__(self):
        return "Bernoulli Likelihood with SVM Link"


class PoissonLikelihood(AbstractLikelihood):
    """PoissonLikelihood(λ) -> PoissonLikelihood(ScaledLogistic([λ]))  src/likelihood/poisson.jl:16-24.

    λ is state: every local update re-estimates it (poisson.jl:78).  `lam` is the constructor value until a model has
    trained with this likelihood, then the value read back from the device."""

    kind = capi.LIK_POISSON

    def __init__(self, lam: float):
        if not lam > 0:
            raise ValueError("λ must be positive")
        self.lam = float(lam)

    def lik_desc(self):
        return capi.LikDesc(self.kind, 1, self.lam, 0.0)

    def __repr__(self):
        return f"Poisson Likelihood (λ = {self.lam})"


class NegBinomialLikelihood(AbstractLikelihood):
    """NegBinomialLikelihood(r)  src/likelihood/negativebinomial.jl:22-27 (LogisticLink)."""

    kind = capi.LIK_NEGBINOMIAL

    def __init__(self, r):
        if not r > 0:
            raise ValueError("r must be positive")
        self.r = float(r)

    def lik_desc(self):
        return capi.LikDesc(self.kind, 1, self.r, 0.0)

    def __repr__(self):
        return f"Negative Binomial Likelihood (r = {self.r:g})"


class HeteroscedasticLikelihood(AbstractLikelihood):
    """HeteroscedasticLikelihood(λ) -> HeteroscedasticGaussianLikelihood(InvScaledLogistic([λ]))
    src/likelihood/heteroscedastic.jl:17-47 ; two latents (f, g) ; λ is state (heteroscedastic.jl:95)."""

    kind = capi.LIK_HETEROSCEDASTIC
    n_latent = 2

    def __init__(self, lam: float = 1.0):
        if not lam > 0:
            raise ValueError("λ must be positive")
        self.lam = float(lam)

    def lik_desc(self):
        return capi.LikDesc(self.kind, 1, self.lam, 0.0)

    def __repr__(self):
        return "Gaussian likelihood with heteroscedastic noise"


def _unique_in_order(y):
    """unique(y) in first-occurrence order (Julia's `unique`); vectorised for numeric / string arrays"""
    arr = np.asarray(y)
    if arr.dtype != object and arr.ndim == 1 and len(arr) > 0:
        _, first = np.unique(arr, return_index=True)
        return [v.item() if hasattr(v, "item") else v for v in arr[np.sort(first)]]
    seen = []
    for v in y:
        if v not in seen:
            seen.append(v)
    return seen


def create_mapping(l: LogisticSoftMaxLikelihood, y):
    """create_mapping!  src/likelihood/multiclass.jl:60-78."""
    K = l.n_latent
    if l.class_mapping is None:
        cm = _unique_in_order(y)
        if len(cm) <= K and all(isinstance(v, (int, np.integer)) and 1 <= v <= K for v in cm):
            cm = list(range(1, K + 1))
        elif len(cm) > K:
            raise RuntimeError(
                f"The number of unique labels in the data : {cm} is not of the same size then the predefined class "
                f"number ; {K}"
            )
        l.class_mapping = cm
    l.ind_mapping = {v: i + 1 for i, v in enumerate(l.class_mapping)}
    return l.ind_mapping


def create_one_hot(l: LogisticSoftMaxLikelihood, y):
    """create_one_hot  src/likelihood/multiclass.jl:81-94 (bool matrix N x K); O(N log N) instead of the reference's loops."""
    arr = np.asarray(y)
    Y = np.zeros((len(arr), l.n_class), dtype=bool)
    if arr.dtype != object and arr.ndim == 1 and len(arr) > 0:
        uniq, inv = np.unique(arr, return_inverse=True)
        col = np.full(len(uniq), -1, dtype=np.int64)
        for u, v in enumerate(uniq):
            key = v.item() if hasattr(v, "item") else v
            for j, c in enumerate(l.class_mapping):
                if key == c:
                    col[u] = j
                    break
        if np.any(col < 0):
            raise RuntimeError("Some labels of y are not part of the expect labels")
        Y[np.arange(len(arr)), col[inv]] = True
        return Y
    for v in _unique_in_order(y):
        if v not in l.class_mapping:
            raise RuntimeError("Some labels of y are not part of the expect labels")
    for i, v in enumerate(y):
        for j in range(l.n_class):
            if v == l.class_mapping[j]:
                Y[i, j] = True
                break
    return Y


def _as_list(y):
    return y.tolist() if isinstance(y, np.ndarray) else list(y)


def treat_labels(y, l: AbstractLikelihood):
    """treat_labels!  regression.jl:10-15, classification.jl:29-44, multiclass.jl:40-44, event.jl:7-13.

    Returns what view_y hands to the inference: real vector (regression), ±1 vector (Bernoulli), one-hot bool
    matrix (multiclass)."""
    if isinstance(l, (GaussianLikelihood, StudentTLikelihood, LaplaceLikelihood, HeteroscedasticLikelihood)):
        arr = np.asarray(y)
        if not (np.issubdtype(arr.dtype, np.floating) or np.issubdtype(arr.dtype, np.integer)):
            raise ValueError("For regression target(s) should be real valued")
        return arr.astype(np.float64)
    if isinstance(l, (PoissonLikelihood, NegBinomialLikelihood)):  # event.jl:7-13
        arr = np.asarray(y)
        if not np.issubdtype(arr.dtype, np.integer):
            raise TypeError("For event count target(s) should be integers")
        return arr.astype(np.float64)
    if isinstance(l, (LogisticLikelihood, BayesianSVM)):
        arr = np.asarray(y)
        if not (np.issubdtype(arr.dtype, np.floating) or np.issubdtype(arr.dtype, np.integer)
                or arr.dtype == bool):
            raise TypeError("For classification target(s) should be real valued (Bool, Integer or Float)")
        labels = sorted(int(v) for v in np.unique(arr))
        if labels == [0, 1]:
            return np.sign(arr.astype(np.float64) - 0.5)
        if labels == [-1, 1]:
            return arr.astype(np.float64)
        raise ValueError("Labels of y should be binary {-1,1} or {0,1}")
    if isinstance(l, LogisticSoftMaxLikelihood):
        arr = np.asarray(y) if not isinstance(y, list) else None
        if arr is not None and arr.ndim > 1:
            raise ValueError("Target should be a vector of labels")
        yl = _as_list(y)
        create_mapping(l, yl)
        return create_one_hot(l, yl)
    raise TypeError(f"likelihood {l} is not implemented on this path")


def class_indices(Y_onehot: np.ndarray) -> np.ndarray:
    """0-based class index per row of a one-hot matrix (device representation of the BitMatrix)."""
    return np.argmax(Y_onehot, axis=1).astype(np.int32)


# ---- point likelihoods p(y | f): `l(y, f)` and `loglikelihood(l, y, f)` of the reference ------------------------------------------
# Host-side scalars, not on the device path (the CAVI step works with the augmented expectations); kept because the reference's
# likelihood test-sets pin them (test/likelihood/gaussian.jl:9-10 and the definitions cited below).
import math as _math


def _logistic(x):
    return 1.0 / (1.0 + _math.exp(-x)) if x >= 0 else _math.exp(x) / (1.0 + _math.exp(x))


def likelihood_value(l: AbstractLikelihood, y, f) -> float:
    """l(y, f) (for the heteroscedastic model the reference's argument order is l(f, y) with f = (f, g): pass f as a pair)."""
    if isinstance(l, GaussianLikelihood):        # gaussian.jl:27-29: pdf(Normal(y, sqrt(sigma2)), f)
        return _math.exp(-0.5 * (f - y) ** 2 / l.sigma2) / _math.sqrt(2.0 * _math.pi * l.sigma2)
    if isinstance(l, LogisticLikelihood):        # classification.jl:6-8: pdf(Bernoulli(logistic(f)), y), y in {0, 1}
        p = _logistic(f)
        return p if y in (1, True) else (1.0 - p if y in (0, False) else 0.0)
    if isinstance(l, BayesianSVM):               # bayesiansvm.jl:25-38: Bernoulli(pos / (pos + neg)), pseudo-likelihood exp(-2 max(1 - f, 0))
        pos, neg = _math.exp(-2.0 * max(1.0 - f, 0.0)), _math.exp(-2.0 * max(1.0 + f, 0.0))
        p = pos / (pos + neg)
        return p if y in (1, True) else (1.0 - p if y in (0, False) else 0.0)
    if isinstance(l, StudentTLikelihood):        # studentt.jl:43-46 (as written there: no 1/nu inside, no 1/sigma in front)
        return (_math.gamma(l.alpha) / (_math.sqrt(l.nu * _math.pi) * _math.gamma(l.nu / 2.0))
                * (1.0 + ((y - f) / l.sigma) ** 2) ** (-l.alpha))
    if isinstance(l, LaplaceLikelihood):         # laplace.jl:36-38: pdf(Laplace(f, beta), y)
        return _math.exp(-abs(y - f) / l.beta) / (2.0 * l.beta)
    if isinstance(l, PoissonLikelihood):         # poisson.jl:26,34-36: pdf(Poisson(lambda logistic(f)), y)
        mu = l.lam * _logistic(f)
        return _math.exp(y * _math.log(mu) - mu - _math.lgamma(y + 1.0)) if y >= 0 and float(y).is_integer() else 0.0
    if isinstance(l, NegBinomialLikelihood):     # negativebinomial.jl:29,33-35: pdf(NegativeBinomial(r, logistic(-f)), y)
        if y < 0 or not float(y).is_integer():
            return 0.0
        pr = _logistic(-f)
        return _math.exp(_math.lgamma(y + l.r) - _math.lgamma(y + 1.0) - _math.lgamma(l.r) + l.r * _math.log(pr)
                         + y * _math.log1p(-pr))
    if isinstance(l, HeteroscedasticLikelihood):  # heteroscedastic.jl:25,34-36: Normal(f, sqrt(1 / (lambda logistic(g)))), called l(f, y)
        (ff, g), yy = y, f
        var = 1.0 / (l.lam * _logistic(g))
        return _math.exp(-0.5 * (yy - ff) ** 2 / var) / _math.sqrt(2.0 * _math.pi * var)
    if isinstance(l, LogisticSoftMaxLikelihood):  # logisticsoftmax.jl:29-31, multiclass.jl:31-33: normalize(logistic.(f), 1)[y]
        s = [_logistic(v) for v in f]
        return s[int(y) - 1] / sum(s)
    raise TypeError(f"no point likelihood for {l!r}")


def loglikelihood(l: AbstractLikelihood, y, f) -> float:
    """Distributions.loglikelihood(l, y, f) of the reference."""
    if isinstance(l, LogisticLikelihood):        # logistic.jl:28-32: -log(1 + exp(-y f)) with y in {-1, 1}
        z = -float(y) * float(f)
        return -(z + _math.log1p(_math.exp(-z))) if z > 0 else -_math.log1p(_math.exp(z))
    v = likelihood_value(l, y, f)
    return _math.log(v) if v > 0 else -_math.inf


AbstractLikelihood.__call__ = lambda self, y, f: likelihood_value(self, y, f)
