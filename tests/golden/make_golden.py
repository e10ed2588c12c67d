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
from oracle import agp_ref as R  # noqa: E402

CASES = [
    # name, likelihood ctor, m, stochastic
    ("gaussian_m8_full", lambda: R.GaussianLikelihood(0.05), 8, False),
    ("gaussian_m64_svi", lambda: R.GaussianLikelihood(0.05), 64, True),
    ("logistic_m8_full", lambda: R.LogisticLikelihood(), 8, False),
    ("logistic_m64_svi", lambda: R.LogisticLikelihood(), 64, True),
    ("studentt_m8_full", lambda: R.StudentTLikelihood(3.0, 1.0), 8, False),
    ("studentt_m64_svi", lambda: R.StudentTLikelihood(3.0, 1.0), 64, True),
    ("logisticsoftmax_m8_full", lambda: R.LogisticSoftMaxLikelihood(3), 8, False),
    ("logisticsoftmax_m64_svi", lambda: R.LogisticSoftMaxLikelihood(3), 64, True),
]


def make(name, lik_ctor, m, stochastic):
    rng = np.random.default_rng(42)
    N, D, B, iters = 200, 3, 50, 10
    X = rng.random((N, D))
    f = np.sin(4 * X[:, 0]) + X[:, 1] ** 2 - 0.8 * X[:, 2]
    lik = lik_ctor()
    if lik.name == "gaussian":
        y = f + 0.2 * rng.standard_normal(N)
    elif lik.name == "logistic":
        y = (f + 0.3 * rng.standard_normal(N) > f.mean()).astype(np.int64)
    elif lik.name == "studentt":
        y = f + 0.2 * rng.standard_t(3, N)
    else:
        y = 1 + np.digitize(f, np.quantile(f, [0.33, 0.66]))
    Z = X[rng.permutation(N)[:m]].copy()
    idx = np.stack([rng.choice(N, B, replace=False) for _ in range(iters)])
    kern = R.Kernel("sqexponential", 3.0, 1.2)
    model = R.SVGP(kern, lik, Z, stochastic=stochastic, batchsize=B)
    yt = R.treat_labels(y, lik)
    snaps, elbos = {}, []

    def cb(M, it, xb, yb):
        elbos.append(M.elbo(yb))
        if it + 1 in (1, 2, 10):
            for k, g in enumerate(M.latents):
                snaps[f"eta1_it{it + 1}_l{k}"] = g.eta1.copy()
                snaps[f"eta2_it{it + 1}_l{k}"] = g.eta2.copy()
                snaps[f"mu_it{it + 1}_l{k}"] = g.mu.copy()
                snaps[f"Sigma_it{it + 1}_l{k}"] = g.Sigma.copy()

    model.train(X, yt, iters, idx_stream=idx, labels_treated=True, callback=cb)
    out = dict(X=X, y=np.asarray(y), Z=Z, idx=idx, scale=3.0, variance=1.2, stochastic=int(stochastic), B=B,
               elbo=np.array(elbos), **snaps)
    for k, g in enumerate(model.latents):
        out[f"kappa_l{k}"] = g.kappa
        out[f"Ktilde_l{k}"] = g.Kt
    lv = model.local_vars
    if lik.name == "logisticsoftmax":
        for k in range(lik.n_class):
            out[f"theta_l{k}"] = lv["theta"][k]
            out[f"gamma_l{k}"] = lv["gamma"][k]
        out["alpha"] = lv["alpha"]
    else:
        out["theta_l0"] = lv["theta"]
    Xt = rng.random((40, D))
    out["Xt"] = Xt
    mu, var = model.predict_f(Xt, cov=True)
    out["pred_mu"] = np.stack(mu)
    out["pred_var"] = np.stack(var)
    pr = model.proba_y(Xt)
    out["proba"] = pr if lik.name == "logisticsoftmax" else np.stack(pr)
    out["elbo_fresh_rho1"] = model.elbo_fresh(X, yt, 1.0)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)


if __name__ == "__main__":
    for c in CASES:
        make(*c)
        print("wrote", c[0])
