"""Shared likelihood table for the golden fixtures and the parity tests: name -> (oracle ctor, host-mirror ctor, labels)."""
import numpy as np


def oracle_lik(R, name):
    return {
        "gaussian": lambda: R.GaussianLikelihood(0.05),
        "logistic": lambda: R.LogisticLikelihood(),
        "studentt": lambda: R.StudentTLikelihood(3.0, 1.0),
        "logisticsoftmax": lambda: R.LogisticSoftMaxLikelihood(3),
        "laplace": lambda: R.LaplaceLikelihood(0.4),
        "bayesiansvm": lambda: R.BayesianSVM(),
        "poisson": lambda: R.PoissonLikelihood(4.0),
        "negbinomial": lambda: R.NegBinomialLikelihood(6.0),
        "heteroscedastic": lambda: R.HeteroscedasticLikelihood(2.0),
    }[name]()


def agp_lik(AGP, name):
    return {
        "gaussian": lambda: AGP.GaussianLikelihood(0.05),
        "logistic": lambda: AGP.LogisticLikelihood(),
        "studentt": lambda: AGP.StudentTLikelihood(3.0, 1.0),
        "logisticsoftmax": lambda: AGP.LogisticSoftMaxLikelihood(3),
        "laplace": lambda: AGP.LaplaceLikelihood(0.4),
        "bayesiansvm": lambda: AGP.BayesianSVM(),
        "poisson": lambda: AGP.PoissonLikelihood(4.0),
        "negbinomial": lambda: AGP.NegBinomialLikelihood(6.0),
        "heteroscedastic": lambda: AGP.HeteroscedasticLikelihood(2.0),
    }[name]()


def labels(name, f, X, rng):
    """synthetic targets for latent function values f (the generators of the reference's test/likelihood/*.jl)"""
    N = len(f)
    sig = lambda t: 1.0 / (1.0 + np.exp(-t))
    if name == "gaussian":
        return f + 0.2 * rng.standard_normal(N)
    if name in ("logistic", "bayesiansvm"):
        return (f + 0.3 * rng.standard_normal(N) > f.mean()).astype(np.int64)
    if name == "studentt":
        return f + 0.2 * rng.standard_t(3, N)
    if name == "logisticsoftmax":
        return 1 + np.digitize(f, np.quantile(f, [0.33, 0.66]))
    if name == "laplace":
        return f + rng.laplace(0.0, 0.4, N)
    if name == "poisson":
        return rng.poisson(4.0 * sig(f)).astype(np.int64)
    if name == "negbinomial":
        return rng.negative_binomial(6, sig(-f)).astype(np.int64)
    if name == "heteroscedastic":
        g = np.cos(5.0 * X[:, 0]) - 1.0
        return f + rng.standard_normal(N) / np.sqrt(2.0 * sig(g))
    raise ValueError(name)
