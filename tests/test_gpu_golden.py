"""GPU: the HIP path against the committed golden fixtures (tests/golden/*.npz, produced by make_golden.py)."""
import glob
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLD, "*.npz"))))
def test_hip_matches_golden(built, path):
    import agp_amd as AGP
    from agp_amd import capi

    g = np.load(path, allow_pickle=True)
    name = os.path.basename(path).split("_")[0]
    from _liks import agp_lik

    lik = agp_lik(AGP, name)
    B = int(g["B"])
    inf = AGP.AnalyticSVI(B) if int(g["stochastic"]) else AGP.AnalyticVI()
    k = float(g["variance"]) * (AGP.SqExponentialKernel() @ AGP.ScaleTransform(float(g["scale"])))
    m = AGP.SVGP(k, lik, inf, g["Z"], optimiser=False)
    elbos, snaps, lams = [], {}, []

    def cb(model, state, it):
        elbos.append(AGP.objective(model, state))
        if hasattr(lik, "lam"):
            model._pull_lik_state()
            lams.append(lik.lam)
        if len(elbos) in (1, 2, 10):
            snaps[len(elbos)] = [model.get_state(l) for l in range(model.n_latent)]

    AGP.train_(m, g["X"], g["y"], 10, idx_stream=g["idx"], callback=cb)
    assert np.allclose(elbos, g["elbo"], rtol=1e-8, atol=1e-7)
    if lams:  # lambda of Poisson / Heteroscedastic after every iteration
        assert np.allclose(lams, g["lam"], rtol=1e-10)
    for it in (1, 2, 10):
        for l, (mu, Sig, e1, e2) in enumerate(snaps[it]):
            assert _rel(e1, g[f"eta1_it{it}_l{l}"]) < 1e-9
            assert _rel(e2, g[f"eta2_it{it}_l{l}"]) < 1e-9
            assert _rel(mu, g[f"mu_it{it}_l{l}"]) < 1e-8
            assert _rel(Sig, g[f"Sigma_it{it}_l{l}"]) < 1e-8
    nb = B if int(g["stochastic"]) else len(g["X"])
    for l in range(m.n_latent):
        assert _rel(m.get_matrix(capi.MAT_KAPPA, l, nb), g[f"kappa_l{l}"]) < 1e-9
        assert _rel(m.get_matrix(capi.VEC_KTILDE, l, nb), g[f"Ktilde_l{l}"]) < 1e-8
        assert _rel(m.get_matrix(capi.VEC_THETA, l, nb), g[f"theta_l{l}"]) < 1e-8
        if f"gamma_l{l}" in g.files and name != "logisticsoftmax":
            assert _rel(m.get_matrix(capi.VEC_GAMMA, l, nb), g[f"gamma_l{l}"]) < 1e-8
        if f"c_l{l}" in g.files:
            assert _rel(m.get_matrix(capi.VEC_C, l, nb), g[f"c_l{l}"]) < 1e-9
    if name == "logisticsoftmax":
        assert _rel(m.get_matrix(capi.VEC_ALPHA, 0, nb), g["alpha"]) < 1e-9
    mu, var = AGP.predict_f(m, g["Xt"], cov=True)
    assert _rel(np.stack(mu) if m.n_latent > 1 else mu[None], g["pred_mu"]) < 1e-8
    assert _rel(np.stack(var) if m.n_latent > 1 else var[None], g["pred_var"]) < 1e-7
    pr = AGP.proba_y(m, g["Xt"])
    if name == "logisticsoftmax":
        assert _rel(np.stack([pr[c] for c in (1, 2, 3)], axis=1), g["proba"]) < 1e-8
    else:
        assert _rel(pr[0], g["proba"][0]) < 1e-8 and _rel(pr[1], g["proba"][1]) < 1e-6
    if "pred_y" in g.files:
        py = np.asarray(AGP.predict_y(m, g["Xt"]), dtype=np.float64)
        assert _rel(py, g["pred_y"]) < 1e-8
    assert AGP.ELBO(m, g["X"], g["y"], rho=1.0) == pytest.approx(float(g["elbo_fresh_rho1"]), rel=1e-8)
    if hasattr(lik, "lam"):  # the external ELBO runs a local update, which re-estimates lambda (reference side effect)
        m._pull_lik_state()
        assert lik.lam == pytest.approx(float(g["lam_final"]), rel=1e-10)
