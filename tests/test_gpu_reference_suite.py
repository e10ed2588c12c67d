"""The reference's own likelihood test-suite, run on the HIP path: for every augmented likelihood of test/likelihood/*.jl the
SVGP rows of `tests_likelihood` (test/testingtools.jl:283-303) with the reference's toy set-ups (N = 20, d = 2, M = 10,
SqExponentialKernel() ∘ ScaleTransform(10.0); N = 100, d = 1, K = 3 for the multi-class model; N = 500, d = 1 for the
heteroscedastic one), the same sequence of calls (`tests`, testingtools.jl:9-19: train 1 -> objective -> train 5 -> testconv ->
proba_y variances > 0, then the second model for 6 iterations) and the same thresholds (`testconv`, testingtools.jl:223-253).
The three models per likelihood are the reference's: AnalyticVI without hyper-optimisation, AnalyticVI with optimiser = true and
Zoptimiser = true, and AnalyticSVI(10).  Threshold tests: they pin behaviour, not digits (parity vs the oracle is elsewhere)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

M_IND = 10


@pytest.fixture(scope="module")
def AGP(built):
    import torch

    assert torch.cuda.is_available()
    import agp_amd as AGP

    return AGP


def generate_f(rng, N, d, scale, variance, X=None):
    """generate_f (testingtools.jl:2-5): a draw from the GP prior with k = variance * SqExponential ∘ ScaleTransform(scale)."""
    if X is None:
        X = rng.random((N, d))
    d2 = ((X[:, None, :] - X[None, :, :]) ** 2).sum(-1) * scale ** 2
    K = variance * np.exp(-0.5 * d2) + 1e-5 * np.eye(N)
    return X, np.linalg.cholesky(K) @ rng.standard_normal(N)


def logistic(x):
    return 1.0 / (1.0 + np.exp(-x))


def conv_ok(AGP, model, problem, X, f, y):
    """`testconv` (testingtools.jl:223-253): every prediction entry point is exercised, then the problem's error threshold."""
    mu, Sig = AGP.predict_f(model, X, cov=True, diag=False)
    mu2, dSig = AGP.predict_f(model, X, cov=True, diag=True)
    first = (lambda v: v[0] if isinstance(v, (tuple, list)) else v)
    S0, d0 = np.asarray(first(Sig)), np.asarray(first(dSig))
    assert S0.shape == (len(X), len(X)) and np.allclose(np.diag(S0), d0, rtol=1e-6, atol=1e-9)
    y_pred = AGP.predict_y(model, X)
    AGP.proba_y(model, X)
    if problem == "Regression":
        return np.mean(np.abs(np.asarray(y_pred) - f)) < 15
    if problem == "Classification":
        return np.mean(np.asarray(y_pred) != y) < 0.5
    if problem == "MultiClass":
        return np.mean(np.asarray(y_pred) != y) < 0.9
    if problem in ("Poisson", "NegBinomial"):
        return np.mean(np.abs(np.asarray(y_pred) - y)) < 20.0
    raise ValueError(problem)


def proba_var_positive(AGP, model, X, problem):
    p = AGP.proba_y(model, X)
    if problem == "MultiClass":  # a table class -> probabilities (predictions.jl:225-247): columns positive, rows sum to one
        P = np.stack([np.asarray(v) for v in p.values()], axis=1)
        return P.shape == (len(X), 3) and np.all(P > 0) and np.allclose(P.sum(axis=1), 1.0, atol=1e-6)
    if isinstance(p, tuple) and len(p) == 2:
        return np.all(np.asarray(p[1]) > 0)
    return np.all(np.asarray(p) >= 0)  # Bernoulli / event likelihoods return the probability (mean) only


def run_tests(AGP, model1, model2, X, f, y, problem):
    """tests (testingtools.jl:9-19)."""
    AGP.train_(model1, X, y, 1)
    assert np.isfinite(AGP.objective(model1))
    AGP.train_(model1, X, y, 5)
    assert conv_ok(AGP, model1, problem, X, f, y)
    assert proba_var_positive(AGP, model1, X, problem)
    AGP.train_(model2, X, y, 6)
    assert conv_ok(AGP, model2, problem, X, f, y)
    assert proba_var_positive(AGP, model2, X, problem)


def make_case(AGP, name, rng):
    """Data and likelihood of test/likelihood/<name>.jl."""
    if name == "logisticsoftmax":
        X, f1 = generate_f(rng, 100, 1, 10.0, 2.0)
        _, f2 = generate_f(rng, 100, 1, 10.0, 2.0, X)
        _, f3 = generate_f(rng, 100, 1, 10.0, 2.0, X)
        y = 1 + np.argmax(np.stack([f1, f2, f3], axis=1), axis=1)
        return X, f1, y, AGP.LogisticSoftMaxLikelihood(3), "MultiClass", 2.0
    if name == "heteroscedastic":
        X, f = generate_f(rng, 500, 1, 10.0, 1.0)
        _, g = generate_f(rng, 500, 1, 10.0, 1.0, X)
        sig = 2.0 * logistic(g - 3.0)
        y = f + rng.standard_normal(500) * np.sqrt(1.0 / sig)
        return X, f, y, AGP.HeteroscedasticLikelihood(2.0), "Regression", 1.0
    var = 2.0 if name == "logistic" else 1.0
    X, f = generate_f(rng, 20, 2, 10.0, var)
    if name == "gaussian":
        return X, f, f + 0.1 * rng.standard_normal(20), AGP.GaussianLikelihood(1e-3), "Regression", var
    if name == "studentt":
        return X, f, f + rng.standard_t(3.0, 20), AGP.StudentTLikelihood(3.0), "Regression", var
    if name == "laplace":
        return X, f, f + rng.laplace(0.0, 3.0, 20), AGP.LaplaceLikelihood(3.0), "Regression", var
    if name == "logistic":
        return X, f, f > 0, AGP.LogisticLikelihood(), "Classification", var
    if name == "bayesiansvm":
        return X, f, f > 0, AGP.BayesianSVM(), "Classification", var
    if name == "poisson":
        return X, f, rng.poisson(5.0 * logistic(f)), AGP.PoissonLikelihood(5.0), "Poisson", var
    if name == "negativebinomial":
        return X, f, rng.negative_binomial(10, 1.0 - logistic(f)), AGP.NegBinomialLikelihood(10.0), "NegBinomial", var
    raise ValueError(name)


NAMES = ["gaussian", "studentt", "laplace", "heteroscedastic", "logistic", "bayesiansvm", "logisticsoftmax", "poisson",
         "negativebinomial"]


@pytest.mark.parametrize("name", NAMES)
def test_svgp_rows_of_the_reference_likelihood_suite(AGP, name):
    rng = np.random.default_rng(42)
    X, f, y, lik, problem, var = make_case(AGP, name, rng)
    Z = AGP.inducingpoints(AGP.KmeansAlg(M_IND), X)
    kern = lambda: var * (AGP.SqExponentialKernel() @ AGP.ScaleTransform(10.0))  # noqa: E731
    import copy

    def svgp(inference, **kw):
        return AGP.SVGP(kern(), copy.deepcopy(lik), inference, [np.array(z) for z in Z], **kw)

    # test_inference_SVGP (testingtools.jl:287-300)
    model = svgp(AGP.AnalyticVI(), optimiser=False)
    assert model.n_latent == (3 if name == "logisticsoftmax" else 2 if name == "heteroscedastic" else 1)
    model_opt = svgp(AGP.AnalyticVI(), optimiser=True, Zoptimiser=True)
    run_tests(AGP, model, model_opt, X, f, y, problem)
    model_svi = svgp(AGP.AnalyticSVI(10), optimiser=False)
    run_tests(AGP, model_svi, model, X, f, y, problem)


class _Online:
    """predict / proba entry points of the streaming model under the names conv_ok uses."""

    def __init__(self, AGP):
        self.predict_f = AGP.online_predict_f
        self.predict_y = AGP.online_predict_y
        self.proba_y = AGP.online_proba_y


OSVGP_NAMES = ["studentt", "laplace", "heteroscedastic", "logistic", "bayesiansvm", "logisticsoftmax", "poisson",
               "negativebinomial"]  # the files whose dict has "OSVGP" => "AVI" => true


@pytest.mark.parametrize("name", OSVGP_NAMES)
def test_osvgp_rows_of_the_reference_likelihood_suite(AGP, name):
    """tests(model1::OnlineSVGP, ...) (testingtools.jl:20-35): the data arrives in batches of 10, five iterations each, with the
    state handed on; first without, then with hyper-parameter optimisation (testingtools.jl:95-116)."""
    import copy

    rng = np.random.default_rng(43)
    X, f, y, lik, problem, var = make_case(AGP, name, rng)
    y = np.asarray(y)
    on = _Online(AGP)
    for optimiser in (False, True):
        model = AGP.OnlineSVGP(var * (AGP.SqExponentialKernel() @ AGP.ScaleTransform(10.0)), copy.deepcopy(lik), AGP.AnalyticVI(),
                               AGP.OIPS(), optimiser=optimiser, seed=7)
        state = None
        for b in range(0, len(X), 10):
            state = AGP.train_online(model, X[b:b + 10], y[b:b + 10], state, iterations=5)
            assert np.isfinite(AGP.online_objective(model))
        assert model.n_latent == (3 if name == "logisticsoftmax" else 2 if name == "heteroscedastic" else 1)
        assert conv_ok(on, model, problem, X, f, y)
        assert proba_var_positive(on, model, X, problem)


def test_mosvgp_testset(AGP):
    """test/models/MOSVGP.jl: two tasks (Logistic, Laplace(2)), one latent per task, N = 20, d = 2, KmeansAlg(5),
    MOSVGP(k, likelihoods, AnalyticVI(), [Z, Z]); train! 10 iterations, then predict_y and proba_y must run."""
    rng = np.random.default_rng(42)
    X, f = generate_f(rng, 20, 2, 10.0, 1.0)
    _, f2 = generate_f(rng, 20, 2, 10.0, 1.0, X)
    ys = [f > 0, f2 + rng.laplace(0.0, 2.0, 20)]
    Z = AGP.inducingpoints(AGP.KmeansAlg(5), X)
    model = AGP.MOSVGP(AGP.SqExponentialKernel() @ AGP.ScaleTransform(10.0), [AGP.LogisticLikelihood(), AGP.LaplaceLikelihood(2.0)],
                       AGP.AnalyticVI(), [Z, Z])
    AGP.train_(model, X, ys, 10)
    yp = AGP.predict_y(model, X)
    pp = AGP.proba_y(model, X)
    assert len(yp) == 2 and len(pp) == 2
    assert np.asarray(yp[0]).shape == (20,) and np.asarray(yp[1]).shape == (20,)
    assert np.all(np.isfinite(np.asarray(yp[1], dtype=float)))
    A = model.get_A()
    assert np.allclose(np.linalg.norm(A, axis=1), 1.0)  # update_A! keeps the rows on the unit sphere (utils :110-112)


def test_analyticvi_object(AGP):
    """test/inference/analyticVI.jl: what the inference object reports before and after set_rho."""
    i = AGP.AnalyticVI()
    assert repr(i) == "Analytic Variational Inference"
    assert i.rho == 1.0 and i.stoch is False
    i = AGP.AnalyticSVI(5)
    assert i.stoch is True and i.batchsize == 5
    i.rho = 20 / 5
    assert i.rho == 4.0
    assert repr(i) == "Analytic Stochastic Variational Inference"


def test_data_wrapping_like_the_reference(AGP):
    """test/data/datacontainer.jl, test/data/utils.jl on the host mirror: a label vector of the wrong length is an error
    (wrap_data), obsdim = 2 takes the observations from the columns (wrap_X), a plain vector is N one-dimensional points, the
    multi-output container takes one label vector per task and rejects a ragged one."""
    rng = np.random.default_rng(3)
    X = rng.random((30, 2))
    y = np.sin(3 * X[:, 0]) + 0.1 * rng.standard_normal(30)
    Z = X[:6].copy()

    def model():
        return AGP.SVGP(AGP.SqExponentialKernel(), AGP.GaussianLikelihood(0.1), AGP.AnalyticVI(), Z, optimiser=False)

    with pytest.raises((ValueError, RuntimeError)):
        AGP.train_(model(), X, np.append(y, 0.0), 2)
    m1, m2 = model(), model()
    AGP.train_(m1, X, y, 3)
    AGP.train_(m2, X.T.copy(), y, 3, obsdim=2)
    assert np.array_equal(m1.get_state(0)[3], m2.get_state(0)[3])
    assert np.allclose(AGP.predict_f(m1, X), AGP.predict_f(m2, X.T.copy(), obsdim=2), rtol=0, atol=0)
    # a vector of N scalars = N one-dimensional points
    x1 = rng.random(25)
    y1 = np.cos(4 * x1)
    mv = AGP.SVGP(AGP.SqExponentialKernel(), AGP.GaussianLikelihood(0.1), AGP.AnalyticVI(), x1[:5, None].copy(), optimiser=False)
    AGP.train_(mv, x1, y1, 3)
    assert AGP.predict_f(mv, x1).shape == (25,)
    # multi-output: one label vector per task, all of the data's length
    ys = [y, (y > 0)]
    mo = AGP.MOSVGP(AGP.SqExponentialKernel(), [AGP.GaussianLikelihood(0.1), AGP.LogisticLikelihood()], AGP.AnalyticVI(), [Z, Z])
    with pytest.raises((ValueError, RuntimeError)):
        AGP.train_(mo, X, [y, np.append(y > 0, True)], 2)
    AGP.train_(mo, X, ys, 2)
