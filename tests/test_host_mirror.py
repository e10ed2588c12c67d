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
