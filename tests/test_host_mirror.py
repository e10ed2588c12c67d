"""Host-side logic of the Python mirror that needs no GPU (argument handling of the reference's constructors, the neutral padding
of the streaming model)."""
import numpy as np
import pytest

import agp_amd as AGP
from agp_amd import online


def test_svgp_accepts_the_reference_forms_of_Z():
    """SVGP.jl:36: Z is a vector of m points; the mirror also takes the (m, D) matrix, and a list of matrices as one Z per latent."""
    rng = np.random.default_rng(0)
    Zm = rng.random((7, 3))
    a = AGP.SVGP(AGP.SqExponentialKernel(), AGP.GaussianLikelihood(0.1), AGP.AnalyticVI(), Zm, optimiser=False)
    b = AGP.SVGP(AGP.SqExponentialKernel(), AGP.GaussianLikelihood(0.1), AGP.AnalyticVI(), [z.copy() for z in Zm], optimiser=False)
    assert (a.m, a.D) == (b.m, b.D) == (7, 3) and np.array_equal(a.Zs[0], b.Zs[0])
    c = AGP.SVGP(AGP.SqExponentialKernel(), AGP.LogisticSoftMaxLikelihood(3), AGP.AnalyticVI(), [Zm, Zm + 1, Zm + 2], optimiser=False)
    assert c.n_latent == 3 and np.array_equal(c.Zs[2], Zm + 2)
    with pytest.raises(ValueError):  # latents of one handle share m
        AGP.SVGP(AGP.SqExponentialKernel(), AGP.LogisticSoftMaxLikelihood(3), AGP.AnalyticVI(), [Zm, Zm[:5], Zm], optimiser=False)


def test_inference_objects():
    assert repr(AGP.AnalyticVI()) == "Analytic Variational Inference"
    i = AGP.AnalyticSVI(10)
    assert i.stoch and i.batchsize == 10 and repr(i) == "Analytic Stochastic Variational Inference"
    with pytest.raises(TypeError):
        AGP.SVGP(AGP.SqExponentialKernel(), AGP.GaussianLikelihood(0.1), "AnalyticVI", np.zeros((2, 1)))


def test_neutral_padding_of_the_streaming_model():
    """online._pad_inducing: shorter latents are filled with points that are far from the unit-scale data and from one another, the
    real points stay in front, counts are reported; kernel values that involve a padding point are exactly zero."""
    rng = np.random.default_rng(1)
    Zs = [rng.random((5, 2)), rng.random((3, 2)), rng.random((5, 2))]
    out, m_real = online._pad_inducing(Zs)
    assert m_real == [5, 3, 5] and all(z.shape == (5, 2) for z in out)
    assert np.array_equal(out[1][:3], Zs[1]) and np.array_equal(out[0], Zs[0])
    pad = out[1][3:]
    d2 = ((pad[:, None, :] - np.vstack([Zs[1], rng.random((50, 2))])[None, :, :]) ** 2).sum(-1)
    assert np.all(np.exp(-0.5 * d2 * 0.01 ** 2) == 0.0)           # even with a lengthscale of 100
    assert np.exp(-0.5 * ((pad[0] - pad[1]) ** 2).sum() * 0.01 ** 2) == 0.0


def test_point_likelihoods_like_the_reference_likelihood_testsets():
    """test/likelihood/gaussian.jl:8-13 (`l(y, f) == pdf(Normal(f, sqrt(σ²)), y)`, `loglikelihood`, `repr`, `n_latent`) and the same
    identities for the other likelihoods of the path, each against scipy.stats' density of the distribution the reference builds."""
    import math

    from scipy import stats

    y, f = 0.2, 0.5
    l = AGP.GaussianLikelihood(1e-3)
    assert l.sigma2 == 1e-3 and l.n_latent == 1
    assert l(y, f) == pytest.approx(stats.norm.pdf(y, f, math.sqrt(1e-3)), rel=1e-13)
    assert AGP.loglikelihood(l, y, f) == pytest.approx(stats.norm.logpdf(y, f, math.sqrt(1e-3)), rel=1e-13)
    assert repr(l) == "Gaussian likelihood (σ² = 0.001)"
    sig = 1.0 / (1.0 + math.exp(-f))
    lg = AGP.LogisticLikelihood()
    assert lg(1, f) == pytest.approx(sig) and lg(0, f) == pytest.approx(1 - sig)
    assert AGP.loglikelihood(lg, 1, f) == pytest.approx(math.log(sig)) and AGP.loglikelihood(lg, -1, f) == pytest.approx(math.log(1 - sig))
    assert AGP.loglikelihood(lg, 1, -800.0) == pytest.approx(-800.0)  # no overflow
    st = AGP.StudentTLikelihood(3.0)  # as the reference writes it (studentt.jl:43-46)
    assert st(y, f) == pytest.approx(math.gamma(2.0) / (math.sqrt(3 * math.pi) * math.gamma(1.5)) * (1 + (y - f) ** 2) ** -2.0)
    la = AGP.LaplaceLikelihood(3.0)
    assert la(y, f) == pytest.approx(stats.laplace.pdf(y, f, 3.0)) and AGP.loglikelihood(la, y, f) == pytest.approx(stats.laplace.logpdf(y, f, 3.0))
    po = AGP.PoissonLikelihood(5.0)
    assert po(3, f) == pytest.approx(stats.poisson.pmf(3, 5.0 * sig)) and po(2.5, f) == 0.0
    nb = AGP.NegBinomialLikelihood(10.0)
    assert nb(4, f) == pytest.approx(stats.nbinom.pmf(4, 10.0, 1.0 - sig))  # NegativeBinomial(r, logistic(-f))
    sv = AGP.BayesianSVM()
    pos, neg = math.exp(-2 * max(1 - f, 0)), math.exp(-2 * max(1 + f, 0))
    assert sv(1, f) == pytest.approx(pos / (pos + neg)) and sv(0, f) == pytest.approx(neg / (pos + neg))
    he = AGP.HeteroscedasticLikelihood(2.0)
    assert he((f, -1.0), y) == pytest.approx(stats.norm.pdf(y, f, math.sqrt(1.0 / (2.0 / (1.0 + math.e)))))
    ls = AGP.LogisticSoftMaxLikelihood(3)
    fs = [0.3, -1.0, 2.0]
    s = [1 / (1 + math.exp(-v)) for v in fs]
    assert [ls(k, fs) for k in (1, 2, 3)] == pytest.approx([v / sum(s) for v in s])
    assert sum(ls(k, fs) for k in (1, 2, 3)) == pytest.approx(1.0)
