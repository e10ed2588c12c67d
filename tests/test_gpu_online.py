"""GPU: OnlineSVGP streaming (csrc online prior + first-step hand-over between handles, host mirror online.py) against the
oracle restatement of src/training/onlinetraining.jl on identical batches.  fp64 tolerances as in test_gpu_parity.py."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


@pytest.fixture(scope="module")
def env(built):
    import torch

    assert torch.cuda.is_available()
    import agp_amd as AGP
    from oracle import agp_ref as R

    return dict(AGP=AGP, R=R)


def _stream(rng, N=360, D=2):
    X = rng.random((N, D)) * np.array([5.0, 2.0])[:D]
    f = np.sin(2 * X[:, 0]) + 0.5 * np.cos(3 * X[:, -1])
    return X, f


@pytest.mark.parametrize("likname", ["gaussian", "logistic", "studentt", "logisticsoftmax", "poisson", "laplace"])
def test_online_svgp_stream_matches_oracle(env, likname):
    from _liks import agp_lik, labels, oracle_lik

    AGP, R = env["AGP"], env["R"]
    rng = np.random.default_rng(17)
    X, f = _stream(rng)
    y = labels(likname, f, X, rng)
    la, lr = agp_lik(AGP, likname), oracle_lik(R, likname)
    if likname == "logisticsoftmax":
        lr.class_mapping = la.class_mapping = [1, 2, 3]  # every batch must see the same mapping
        la.ind_mapping = {v: i + 1 for i, v in enumerate([1, 2, 3])}
    ka = 1.2 * (AGP.SqExponentialKernel() @ AGP.ScaleTransform(1.5))
    kr = R.Kernel("sqexponential", 1.5, 1.2)
    ma = AGP.OnlineSVGP(ka, la, AGP.AnalyticVI(), AGP.OIPS(0.7), optimiser=False)
    mr = R.OnlineSVGP(kr, lr, R.OIPS(0.7))
    Bs, iters = 60, 3
    for b in range(0, len(X), Bs):
        xb, yb = X[b:b + Bs], y[b:b + Bs]
        ea, er = [], []
        AGP.train_online(ma, xb, yb, iterations=iters, callback=lambda m, cur, i: ea.append(AGP.online_objective(m)))
        ybt = R.treat_labels(yb, lr)
        mr.train(xb, yb, iters, callback=lambda m, it, xx, yy: er.append(m.elbo(ybt)))
        assert len(ma.Zs[0]) == len(mr.latents[0]["Z"])
        for l in range(ma.n_latent):
            assert _rel(ma.Zs[l], mr.latents[l]["Z"]) < 1e-14  # OIPS picked the same points
            mu, Sig, e1, e2 = ma.get_state(l)
            g = mr.latents[l]
            assert _rel(e1, g["eta1"]) < 1e-8, (b, l)
            assert _rel(e2, g["eta2"]) < 1e-8
            assert _rel(mu, g["mu"]) < 1e-7 and _rel(Sig, g["Sigma"]) < 1e-7
        assert np.allclose(ea, er, rtol=1e-7, atol=1e-6), (b, ea, er)
        if hasattr(lr, "lam"):
            assert la.lam == pytest.approx(lr.lam, rel=1e-9)
    assert len(ma.Zs[0]) > 5
    Xt = rng.random((70, X.shape[1])) * np.array([5.0, 2.0])
    mfa = AGP.online_predict_f(ma, Xt, cov=True)
    mfr = mr.predict_f(Xt, cov=True)
    if ma.n_latent == 1:
        assert _rel(mfa[0], mfr[0][0]) < 1e-7 and _rel(mfa[1], mfr[1][0]) < 1e-6
        pa, pr = AGP.online_proba_y(ma, Xt), mr.proba_y(Xt)
        assert _rel(pa[0], pr[0]) < 1e-7
    else:
        for l in range(ma.n_latent):
            assert _rel(mfa[0][l], mfr[0][l]) < 1e-7
        assert np.array_equal(AGP.online_predict_y(ma, Xt), mr.predict_y(Xt))


def test_online_learns_and_rejects_unsupported(env):
    AGP = env["AGP"]
    rng = np.random.default_rng(3)
    X, f = _stream(rng, N=600, D=1)
    y = f + 0.1 * rng.standard_normal(len(f))
    m = AGP.OnlineSVGP(AGP.SqExponentialKernel() @ AGP.ScaleTransform(2.0), AGP.GaussianLikelihood(0.01), AGP.AnalyticVI(),
                       AGP.OIPS(0.8), optimiser=False)
    errs = []
    for b in range(0, 600, 100):
        AGP.train_online(m, X[b:b + 100], y[b:b + 100], iterations=3)
        errs.append(np.mean(np.abs(AGP.online_predict_y(m, X) - f)))
    assert errs[-1] < 0.05 and errs[-1] < errs[0]
    v = AGP.online_proba_y(m, X)[1]
    assert np.all(v > 0)
    with pytest.raises(NotImplementedError):
        AGP.OnlineSVGP(AGP.SqExponentialKernel(), AGP.GaussianLikelihood(0.01), AGP.AnalyticSVI(10))


@pytest.mark.parametrize("likname,ard,zopt", [("gaussian", False, True), ("logistic", True, False)])
def test_online_hyper_steps_match_oracle(env, likname, ard, zopt):
    """OnlineSVGP with its default-on hyper-parameter optimisation: the differentiated ELBO includes extraKL, whose kernel
    matrices (K_ab, kappa_a, K~_a) move with the kernel and with Z (oracle gradient FD-pinned in tests/test_oracle_kat.py)."""
    from _liks import agp_lik, labels, oracle_lik

    AGP, R = env["AGP"], env["R"]
    rng = np.random.default_rng(23)
    X, f = _stream(rng, N=300)
    y = labels(likname, f, X, rng)
    sc = np.array([1.5, 1.0]) if ard else 1.5
    ka = 1.2 * (AGP.SqExponentialKernel() @ (AGP.ARDTransform(sc) if ard else AGP.ScaleTransform(sc)))
    ma = AGP.OnlineSVGP(ka, agp_lik(AGP, likname), AGP.AnalyticVI(), AGP.OIPS(0.7), optimiser=AGP.ADAM(0.01),
                        Zoptimiser=AGP.ADAM(0.001) if zopt else False)
    mr = R.OnlineSVGP(R.Kernel("sqexponential", sc, 1.2), oracle_lik(R, likname), R.OIPS(0.7), k_opt=R.Adam(0.01),
                      z_opt=R.Adam(0.001) if zopt else None)
    for b in range(0, len(X), 60):
        AGP.train_online(ma, X[b:b + 60], y[b:b + 60], iterations=4)
        mr.train(X[b:b + 60], y[b:b + 60], 4)
        g = mr.latents[0]
        k = ma._cur.kernels[0]
        assert k.variance == pytest.approx(g["kernel"].sigma2, rel=1e-7), b
        got = np.asarray(k.transform.v if ard else k.transform.s, dtype=float)
        assert _rel(got, np.asarray(g["kernel"].scale, dtype=float)) < 1e-7
        assert len(ma.Zs[0]) == len(g["Z"]) and _rel(ma.Zs[0], g["Z"]) < 1e-7
        mu, Sig, e1, e2 = ma.get_state(0)
        assert _rel(e2, g["eta2"]) < 1e-6 and _rel(mu, g["mu"]) < 1e-6


def test_online_multilatent_with_different_inducing_counts(env):
    """A multi-class streaming model with hyper-optimisation on: every latent runs OIPS with its own (diverging) kernel, so the
    latents end up with different numbers of inducing points (onlinetraining.jl:153-160).  The device handle shares m; the host
    mirror fills the shorter latents with neutral far-away points (online.py, _pad_inducing), which must not change anything:
    same Z, eta, posterior and predictions as the oracle, whose latents simply have different sizes."""
    from _liks import agp_lik, labels, oracle_lik

    AGP, R = env["AGP"], env["R"]
    rng = np.random.default_rng(5)
    X, f = _stream(rng, N=240)
    y = labels("logisticsoftmax", f, X, rng)
    ka = 1.0 * (AGP.SqExponentialKernel() @ AGP.ScaleTransform(2.0))
    ma = AGP.OnlineSVGP(ka, agp_lik(AGP, "logisticsoftmax"), AGP.AnalyticVI(), AGP.OIPS(0.8), optimiser=AGP.ADAM(0.02))
    mr = R.OnlineSVGP(R.Kernel("sqexponential", 2.0, 1.0), oracle_lik(R, "logisticsoftmax"), R.OIPS(0.8), k_opt=R.Adam(0.02))
    differed = False
    lr_ = mr.likelihood
    for b in range(0, len(X), 40):
        ea, er = [], []
        AGP.train_online(ma, X[b:b + 40], y[b:b + 40], iterations=6, callback=lambda m, cur, i: ea.append(AGP.online_objective(m)))
        ybt = R.treat_labels(y[b:b + 40], lr_)
        mr.train(X[b:b + 40], y[b:b + 40], 6, callback=lambda m, it, xx, yy: er.append(m.elbo(ybt)))
        assert np.allclose(ea, er, rtol=1e-6, atol=1e-5), (b, ea, er)  # ELBO incl. extraKL: the padding adds no constant either
        counts = [len(g["Z"]) for g in mr.latents]
        differed |= len(set(counts)) > 1
        assert [len(z) for z in ma.Zs] == counts, b
        for l, g in enumerate(mr.latents):
            assert _rel(ma.Zs[l], g["Z"]) < 1e-7
            mu, Sig, e1, e2 = ma.get_state(l)
            assert _rel(e2, g["eta2"]) < 1e-8 and _rel(mu, g["mu"]) < 1e-8 and _rel(Sig, g["Sigma"]) < 1e-8, (b, l)
            assert ma._cur.kernels[l].variance == pytest.approx(g["kernel"].sigma2, rel=1e-9)
    assert differed, "the set-up no longer makes the latents' inducing-point counts diverge: the padding path was not exercised"
    Xt = rng.random((50, X.shape[1]))
    pa, pr = AGP.online_predict_f(ma, Xt, cov=True), mr.predict_f(Xt, cov=True)
    for l in range(3):
        assert _rel(pa[0][l], pr[0][l]) < 1e-6 and _rel(pa[1][l], pr[1][l]) < 1e-6
    assert np.array_equal(AGP.online_predict_y(ma, Xt), mr.predict_y(Xt))
