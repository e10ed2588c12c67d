"""Generate the golden fixtures tests/golden/*.npz.

The reference is pure Julia and cannot run in the build container or on the GPU box (no julia binary, no network), so
the vectors come from the line-by-line NumPy restatement oracle/agp_ref.py (seeded numpy.random.default_rng(42)).
Each file holds inputs (X, y, Z, minibatch index stream, kernel/likelihood parameters) and expected outputs after
1, 2 and 10 CAVI steps (eta1, eta2, mu, Sigma, ELBO trace, kappa / K~ / mean_f / var_f / theta of the last step) plus
predictions on held-out points.  Run:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import agp_ref as R  # noqa: E402
from _liks import labels, oracle_lik  # noqa: E402

CASES = [
    # name (likelihood_m{m}_{full|svi}), m, stochastic
    ("gaussian_m8_full", 8, False),
    ("gaussian_m64_svi", 64, True),
    ("logistic_m8_full", 8, False),
    ("logistic_m64_svi", 64, True),
    ("studentt_m8_full", 8, False),
    ("studentt_m64_svi", 64, True),
    ("logisticsoftmax_m8_full", 8, False),
    ("logisticsoftmax_m64_svi", 64, True),
    ("laplace_m8_full", 8, False),
    ("laplace_m64_svi", 64, True),
    ("bayesiansvm_m8_full", 8, False),
    ("bayesiansvm_m64_svi", 64, True),
    ("poisson_m8_full", 8, False),
    ("poisson_m64_svi", 64, True),
    ("negbinomial_m8_full", 8, False),
    ("negbinomial_m64_svi", 64, True),
    ("heteroscedastic_m8_full", 8, False),
    ("heteroscedastic_m64_svi", 64, True),
]
NEW = ("laplace", "bayesiansvm", "poisson", "negbinomial", "heteroscedastic")


def make(name, m, stochastic):
    rng = np.random.default_rng(42)
    N, D, B, iters = 200, 3, 50, 10
    X = rng.random((N, D))
    f = np.sin(4 * X[:, 0]) + X[:, 1] ** 2 - 0.8 * X[:, 2]
    lname = name.split("_")[0]
    lik = oracle_lik(R, lname)
    y = labels(lname, f, X, rng)
    Z = X[rng.permutation(N)[:m]].copy()
    idx = np.stack([rng.choice(N, B, replace=False) for _ in range(iters)])
    kern = R.Kernel("sqexponential", 3.0, 1.2)
    model = R.SVGP(kern, lik, Z, stochastic=stochastic, batchsize=B)
    yt = R.treat_labels(y, lik)
    snaps, elbos, lams = {}, [], []

    def cb(M, it, xb, yb):
        elbos.append(M.elbo(yb))
        lams.append(getattr(M.likelihood, "lam", 0.0))
        if it + 1 in (1, 2, 10):
            for k, g in enumerate(M.latents):
                snaps[f"eta1_it{it + 1}_l{k}"] = g.eta1.copy()
                snaps[f"eta2_it{it + 1}_l{k}"] = g.eta2.copy()
                snaps[f"mu_it{it + 1}_l{k}"] = g.mu.copy()
                snaps[f"Sigma_it{it + 1}_l{k}"] = g.Sigma.copy()

    model.train(X, yt, iters, idx_stream=idx, labels_treated=True, callback=cb)
    out = dict(X=X, y=np.asarray(y), Z=Z, idx=idx, scale=3.0, variance=1.2, stochastic=int(stochastic), B=B,
               elbo=np.array(elbos), lam=np.array(lams), **snaps)
    for k, g in enumerate(model.latents):
        out[f"kappa_l{k}"] = g.kappa
        out[f"Ktilde_l{k}"] = g.Kt
    lv = model.local_vars
    if lik.name == "logisticsoftmax":
        for k in range(lik.n_class):
            out[f"theta_l{k}"] = lv["theta"][k]
            out[f"gamma_l{k}"] = lv["gamma"][k]
        out["alpha"] = lv["alpha"]
    elif lik.name == "heteroscedastic":  # device convention: theta of latent 0 = lambda*sigg (= 2 grad_E_Sigma[0])
        out["theta_l0"] = lik.lam * lv["sigg"]
        out["theta_l1"] = lv["theta"]
        out["gamma_l0"] = lv["gamma"]
        out["gamma_l1"] = lv["sigg"]
        out["c_l0"] = lv["phi"]
        out["c_l1"] = lv["c"]
    else:
        out["theta_l0"] = lv["theta"]
        if lik.name == "poisson":
            out["gamma_l0"] = lv["gamma"]
    Xt = rng.random((40, D))
    out["Xt"] = Xt
    mu, var = model.predict_f(Xt, cov=True)
    out["pred_mu"] = np.stack(mu)
    out["pred_var"] = np.stack(var)
    pr = model.proba_y(Xt)
    out["proba"] = pr if lik.name == "logisticsoftmax" else np.stack(pr)
    if lname in NEW:
        out["pred_y"] = np.asarray(model.predict_y(Xt), dtype=np.float64)
    out["elbo_fresh_rho1"] = model.elbo_fresh(X, yt, 1.0)
    out["lam_final"] = getattr(lik, "lam", 0.0)  # elbo_fresh runs a local update, which re-estimates lambda
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)


if __name__ == "__main__":
    only_new = "--new" in sys.argv  # the original eight fixtures stay byte-identical unless regenerated on purpose
    for c in CASES:
        if only_new and c[0].split("_")[0] not in NEW:
            continue
        make(*c)
        print("wrote", c[0])
