"""save_trained_model(filename, model) / load_trained_model(filename)  -- documented by the reference as "in construction"
(docs/src/userguide.md:207-215) and absent from its src; here they are the host half of the checkpoint / resume story
(SURVEY.md section 5): everything a handle needs is exported through the C ABI (agp_svgp_get_state / get_kernel / get_Z /
get_opt_state / get_lik_param / get_A) into one .npz and pushed back into a fresh handle on load.  Unlike the reference's
note, a reloaded model CAN be trained further (the Robbins-Monro counter travels with it; ADAM moments restart).
"""
from __future__ import annotations

import json

import numpy as np

from . import kernels as K
from . import likelihoods as LK
from .svgp import ADAM, MOSVGP, SVGP, AnalyticSVI, AnalyticVI, Descent, Momentum, RobbinsMonro

_KERNELS = {"SqExponentialKernel": K.SqExponentialKernel, "Matern52Kernel": K.Matern52Kernel,
            "Matern32Kernel": K.Matern32Kernel, "ExponentialKernel": K.ExponentialKernel}


def _kernel_spec(k):
    tr = k.transform
    if type(k).__name__ not in _KERNELS:
        raise TypeError(f"cannot serialise kernel {type(k).__name__}")
    ard = isinstance(tr, K.ARDTransform)
    scale = tr.v.tolist() if ard else (float(tr.s) if tr is not None else 1.0)
    return {"base": type(k).__name__, "variance": float(k.variance), "ard": ard, "scale": scale,
            "has_variance": bool(k.has_variance), "has_transform": tr is not None}


def _kernel_from(spec):
    k = _KERNELS[spec["base"]]()
    if spec.get("has_transform", True):
        k = k @ (K.ARDTransform(np.asarray(spec["scale"])) if spec["ard"] else K.ScaleTransform(spec["scale"]))
    return spec["variance"] * k if spec.get("has_variance", True) else k


def _opt_spec(o):
    if o is None:
        return None
    d = {"rule": type(o).__name__, "eta": o.eta}
    for a in ("rho", "beta", "eps"):
        if hasattr(o, a):
            d[a] = getattr(o, a)
    return d


def _opt_from(d):
    if not d:
        return False
    if not isinstance(d, dict):  # files written before the rule was stored: ADAM(eta)
        return ADAM(d)
    if d["rule"] == "Descent":
        return Descent(d["eta"])
    if d["rule"] == "Momentum":
        return Momentum(d["eta"], d["rho"])
    return ADAM(d["eta"], tuple(d.get("beta", (0.9, 0.999))), d.get("eps", 1e-8))


def _lik_spec(l):
    d = {"type": type(l).__name__}
    for a in ("sigma2", "noise_eta", "nu", "sigma", "beta", "lam", "r", "n_class", "class_mapping"):
        if hasattr(l, a):
            v = getattr(l, a)
            d[a] = v if not isinstance(v, np.generic) else v.item()
    return d


def _lik_from(d):
    t = d["type"]
    if t == "GaussianLikelihood":
        eta = d.get("noise_eta", 0.0)  # opt_noise: the model keeps optimising its noise after a reload (ADAM moments restart)
        return LK.GaussianLikelihood(d["sigma2"], opt_noise=ADAM(eta) if eta else False)
    if t == "LogisticLikelihood":
        return LK.LogisticLikelihood()
    if t == "StudentTLikelihood":
        return LK.StudentTLikelihood(d["nu"], d["sigma"])
    if t == "LogisticSoftMaxLikelihood":
        return LK.LogisticSoftMaxLikelihood(d["class_mapping"] if d.get("class_mapping") else d["n_class"])
    if t == "LaplaceLikelihood":
        return LK.LaplaceLikelihood(d["beta"])
    if t == "BayesianSVM":
        return LK.BayesianSVM()
    if t == "PoissonLikelihood":
        return LK.PoissonLikelihood(d["lam"])
    if t == "NegBinomialLikelihood":
        return LK.NegBinomialLikelihood(d["r"])
    if t == "HeteroscedasticLikelihood":
        return LK.HeteroscedasticLikelihood(d["lam"])
    raise ValueError(f"unknown likelihood {t}")


def save_trained_model(filename: str, model: SVGP) -> None:
    """save_trained_model(filename, model): kernels, inducing points, natural parameters, likelihood state, mixing weights."""
    if model._h is None:
        raise RuntimeError("the model has no device state yet: train it first")
    model._pull_hypers()
    model._pull_lik_state()
    mo = isinstance(model, MOSVGP)
    inf = model.inference
    opt = inf.optimiser or RobbinsMonro()
    meta = {
        "class": "MOSVGP" if mo else "SVGP",
        "kernels": [_kernel_spec(k) for k in model.kernels],
        "likelihood": [_lik_spec(l) for l in model.likelihood.likelihoods] if mo else _lik_spec(model.likelihood),
        "stochastic": bool(inf.stoch), "batchsize": int(inf.batchsize), "n_iter": int(inf.n_iter),
        "rm": [opt.kappa, opt.tau], "T": str(model.T), "elbo_mode": model.elbo_mode,
        "k_opt": _opt_spec(model.k_opt), "z_opt": _opt_spec(model.z_opt),
        "atfrequency": model.atfrequency, "mean": model.mean if np.isscalar(model.mean) or model.mean is None else None,
        "jitter": getattr(model, "jitter", None), "stale_K": bool(getattr(model, "reference_compat_stale_K", False)),
        # the mixing weights' optimiser of a multi-output model (update_A!, single_and_multi_output_utils.jl:87-118): without it a
        # reloaded model silently froze A; its moments restart, like the Z optimiser's
        "a_opt": _opt_spec(getattr(model, "A_opt", None)) if mo else None,
    }
    import ctypes as C

    from . import capi

    n_opt = C.c_int64()
    model._chk(capi.lib().agp_svgp_get_opt_state(model._h, C.byref(n_opt)))
    arrays = {"meta": np.array(json.dumps(meta)), "n_opt": np.array(n_opt.value)}
    for l in range(model.n_latent):
        mu, Sig, e1, e2 = model.get_state(l)
        arrays[f"eta1_{l}"], arrays[f"eta2_{l}"], arrays[f"Z_{l}"] = e1, e2, model.Zs[l]
    if mo:
        arrays["A"] = model.get_A()
    if model.k_opt is not None:
        # moments / velocity and step count of every latent's kernel-parameter optimiser (agp_svgp_hyper_opt_state): a resumed run
        # continues the optimiser instead of taking a bias-corrected first step again
        for l in range(model.n_latent):
            km, kv, ks = (C.c_double * (1 + model.D))(), (C.c_double * (1 + model.D))(), C.c_int32()
            model._chk(capi.lib().agp_svgp_hyper_opt_state(model._h, l, 0, km, kv, C.byref(ks)))
            arrays[f"kopt_m_{l}"], arrays[f"kopt_v_{l}"], arrays[f"kopt_t_{l}"] = np.array(km[:]), np.array(kv[:]), np.array(ks.value)
    if isinstance(model.likelihood, LK.LogisticSoftMaxLikelihood) and inf.batchsize > 0:
        # carried between minibatches; exported by capacity (it is state, not a view of the last batch)
        arrays["lsm_alpha"] = model.get_matrix(capi.VEC_ALPHA, 0, min(int(inf.batchsize), model._max_batch))
    if isinstance(model.mean, (list, np.ndarray)):
        arrays["mean_vec"] = np.asarray(model.mean, dtype=np.float64)
    np.savez_compressed(filename, **arrays)


def load_trained_model(filename: str, *, device=None):
    """load_trained_model(filename) -> the model with its posterior restored on the device (ready to predict or train on).

    What a resumed `train!(model, ...; state=...)` continues and what restarts: the natural parameters, kernels, inducing points,
    the Robbins-Monro counter, likelihood state (lambda, LogisticSoftMax alpha), mixing weights and the moments + step count of
    the kernel-parameter optimisers are restored -- the kernel parameters move on as if the run had never stopped.  The device
    state of the Z optimiser, of `GaussianLikelihood(opt_noise=...)` and of `Aoptimiser` RESTARTS (zero moments, step 0): their
    first step after a reload is a bias-corrected first ADAM step, so the inducing points / noise / mixing weights of a resumed
    run differ from an uninterrupted one at the size of one optimiser step."""
    import ctypes as C

    from . import capi

    g = np.load(filename if str(filename).endswith(".npz") else str(filename) + ".npz", allow_pickle=False)
    meta = json.loads(str(g["meta"]))
    kernels = [_kernel_from(s) for s in meta["kernels"]]
    nl = len(kernels)
    Zs = [g[f"Z_{l}"] for l in range(nl)]
    inf = AnalyticSVI(meta["batchsize"], optimiser=RobbinsMonro(*meta["rm"])) if meta["stochastic"] else AnalyticVI()
    T = np.float64 if "64" in meta["T"] else np.float32
    mean = g["mean_vec"] if "mean_vec" in g.files else meta.get("mean")
    kw = dict(optimiser=_opt_from(meta["k_opt"]), Zoptimiser=_opt_from(meta["z_opt"]),
              atfrequency=meta["atfrequency"], mean=mean, T=T, device=device, elbo_mode=meta["elbo_mode"],
              jitter=meta.get("jitter"), reference_compat_stale_K=meta.get("stale_K", False))
    if meta["class"] == "MOSVGP":
        kw.pop("jitter"), kw.pop("reference_compat_stale_K")  # (not constructor arguments of the multi-output model)
        # (files written before round 6 carry no "a_opt": the model's default ADAM(0.01), MOSVGP.jl:42, as for a fresh model)
        a_opt = _opt_from(meta["a_opt"]) if "a_opt" in meta else None
        model = MOSVGP(kernels, [_lik_from(d) for d in meta["likelihood"]], inf, Zs, A=g["A"], Aoptimiser=a_opt, **kw)
    else:
        model = SVGP(kernels, _lik_from(meta["likelihood"]), inf, Zs, **kw)
    inf.n_iter = meta["n_iter"]
    h = model._ensure_handle(max(meta["batchsize"], 1))
    for l in range(nl):
        model.set_state(l, g[f"eta1_{l}"], g[f"eta2_{l}"])
    model._chk(capi.lib().agp_svgp_set_opt_state(h, int(g["n_opt"])))
    for l in range(nl):
        if f"kopt_m_{l}" in g.files and model.k_opt is not None:
            km = (C.c_double * (1 + model.D))(*[float(v) for v in g[f"kopt_m_{l}"]])
            kv = (C.c_double * (1 + model.D))(*[float(v) for v in g[f"kopt_v_{l}"]])
            ks = C.c_int32(int(g[f"kopt_t_{l}"]))
            model._chk(capi.lib().agp_svgp_hyper_opt_state(h, l, 1, km, kv, C.byref(ks)))
    lik = model.likelihood
    if hasattr(lik, "lam") and not isinstance(lik, list):
        model._chk(capi.lib().agp_svgp_set_lik_param(h, float(lik.lam)))
    if "lsm_alpha" in g.files:
        import torch

        a = torch.as_tensor(g["lsm_alpha"], dtype=model.tdtype, device=model._dev()).contiguous()
        model._chk(capi.lib().agp_svgp_set_lsm_alpha(h, C.c_void_p(a.data_ptr()), a.numel()))
        model._chk(capi.lib().agp_ctx_sync(model._ctx))
    model.trained = True
    return model
