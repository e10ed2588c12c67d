"""Host-side mirror of the reference's SVGP / AnalyticVI / train! / predict API.

Reference                                                         here
  SVGP(kernel, likelihood, inference, Z; ...)   SVGP.jl:33-80       SVGP(kernel, likelihood, inference, Z, ...)
  AnalyticVI(; ϵ) / AnalyticSVI(B; ϵ, optimiser) analyticVI.jl:44-52 AnalyticVI(eps) / AnalyticSVI(B, eps, optimiser)
  train!(model, X, y, iterations; callback, state, obsdim)          train_(model, X, y, iterations, callback=..., state=..., obsdim=1)
  predict_f / predict_y / proba_y                predictions.jl      predict_f / predict_y / proba_y
  ELBO(model, X, y) / objective(model, state, y) ELBO.jl, SVGP.jl:90 ELBO / objective

Everything numeric is a call through the C ABI (capi.py -> libagp_hip.so).  torch supplies device buffers and the
HIP stream.  There is no CPU path: constructing a device handle without a GPU raises.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Callable, Optional, Sequence

import numpy as np

from . import capi
from .kernels import Kernel
from .likelihoods import (
    BayesianSVM,
    HeteroscedasticLikelihood,
    LaplaceLikelihood,
    NegBinomialLikelihood,
    PoissonLikelihood,
    AbstractLikelihood,
    GaussianLikelihood,
    LogisticLikelihood,
    LogisticSoftMaxLikelihood,
    StudentTLikelihood,
    class_indices,
    treat_labels,
)


def _torch():
    import torch

    return torch


class RobbinsMonro:
    """RobbinsMonro(κ=0.51, τ=1)  src/inference/optimisers.jl:1-19."""

    def __init__(self, kappa: float = 0.51, tau: float = 1.0):
        if not (0.5 < kappa <= 1):
            raise ValueError("κ should be in the interval (0.5,1]")
        if not tau > 0:
            raise ValueError("τ should be positive")
        self.kappa = float(kappa)
        self.tau = float(tau)


class ADAM:
    """ADAM(η=0.001, β=(0.9, 0.999)) of Optimisers.jl (bias-corrected, ε = 1e-8); used for the mixing weights A."""

    def __init__(self, eta: float = 0.001, beta=(0.9, 0.999), eps: float = 1e-8):
        self.eta, self.beta, self.eps = float(eta), (float(beta[0]), float(beta[1])), float(eps)


class Descent:
    """Descent(η=0.1) of Optimisers.jl: dx' = η dx (the reference adds it: ascent on the ELBO, autotuning_utils.jl:63-76)."""

    def __init__(self, eta: float = 0.1):
        self.eta = float(eta)


class Momentum:
    """Momentum(η=0.01, ρ=0.9) of Optimisers.jl: vel = ρ vel + η dx ; dx' = vel."""

    def __init__(self, eta: float = 0.01, rho: float = 0.9):
        self.eta, self.rho = float(eta), float(rho)


def _opt_rule(o):
    """(rule, rho) of agp_svgp_hyper_rule for an optimiser object"""
    if isinstance(o, Descent):
        return capi.OPT_DESCENT, 0.0
    if isinstance(o, Momentum):
        return capi.OPT_MOMENTUM, o.rho
    return capi.OPT_ADAM, 0.0


class _MultiOutputLikelihood(AbstractLikelihood):
    """Tuple of task likelihoods of a MOSVGP (nf_per_task = 1 on this path)."""

    kind = capi.LIK_MULTIOUTPUT

    def __init__(self, likelihoods, n_latent):
        for l in likelihoods:
            if not isinstance(l, (GaussianLikelihood, LogisticLikelihood, StudentTLikelihood, LaplaceLikelihood,
                                  BayesianSVM, NegBinomialLikelihood)):
                raise RuntimeError(f"One (or more) of the likelihoods {likelihoods} are not compatible or implemented "
                                   "with the multi-output analytic path")  # MOSVGP.jl:56-58
        self.likelihoods = list(likelihoods)
        self.n_latent = n_latent

    def lik_desc(self):
        return capi.LikDesc(self.kind, 1, 0.0, 0.0)

    def __repr__(self):
        return "(" + ", ".join(repr(l) for l in self.likelihoods) + ")"


class AnalyticVI:
    """AnalyticVI(; ϵ=1e-5) -- full-batch CAVI, Descent(1.0) on the natural parameters (analyticVI.jl:44-46)."""

    def __init__(self, eps: float = 1e-5, *, _stoch=False, _batchsize=0, _optimiser=None):
        self.eps = float(eps)
        self.n_iter = 0
        self.stoch = _stoch
        self.batchsize = int(_batchsize)
        self.rho = 1.0
        self.HyperParametersUpdated = True
        self.optimiser = _optimiser

    def __repr__(self):
        return f"Analytic{' Stochastic' if self.stoch else ''} Variational Inference"


def AnalyticSVI(nMinibatch: int, eps: float = 1e-5, optimiser: Optional[RobbinsMonro] = None) -> AnalyticVI:
    """AnalyticSVI(nMinibatch; ϵ=1e-5, optimiser=RobbinsMonro())  analyticVI.jl:48-52."""
    opt = optimiser if optimiser is not None else RobbinsMonro()
    if not isinstance(opt, RobbinsMonro):
        raise NotImplementedError("only RobbinsMonro is wired on this path (ALRSVI is dead code in the reference)")
    return AnalyticVI(eps, _stoch=True, _batchsize=int(nMinibatch), _optimiser=opt)


class State:
    """What train! returns next to the model (states.jl:1-9): a reference to the device-resident state."""

    def __init__(self, model):
        self.model = model


class SVGP:
    """Sparse Variational GP (src/models/SVGP.jl:22-80).  Z: (m, D) array of inducing points (rows = points).

    Keyword arguments follow the reference (SVGP.jl:33-44): `optimiser` (kernel parameters; default ADAM(0.01), Bool ->
    ADAM(0.001) / off), `Zoptimiser` (inducing points; default off, True -> ADAM(0.001)), `atfrequency`.  The hyper step is
    the hand-derived gradient of libagp_hip (agp_svgp_hyper_step), ADAM ascent with positive parameters in log space.
    """

    def __init__(self, kernel, likelihood, inference, Z, *, verbose: int = 0, optimiser=None, atfrequency: int = 1,
                 mean=None, Zoptimiser=False, T=np.float64, device: Optional[int] = None, seed: Optional[int] = None,
                 elbo_mode: str = "corrected", latent_slice: Optional[tuple] = None,
                 reference_compat_stale_K: bool = False, jitter: Optional[float] = None):
        if not isinstance(inference, AnalyticVI):
            raise TypeError("The inference object should be of type `VariationalInference` : either `AnalyticVI` or "
                            "`NumericalVI`")  # SVGP.jl:45-47 (only AnalyticVI exists on this path)
        # inside one train! the reference keeps the Cholesky of K_ZZ of the first iteration even after
        # hyper-parameter steps (training.jl:187-208).  False (default): K is refreshed; True: mirror the reference.
        self.reference_compat_stale_K = bool(reference_compat_stale_K)
        # jitt of the reference is a constant per float type (src/functions/utils.jl:8-9: 1e-4 for Float64, 1e-3 for Float32): None
        # keeps it; a value overrides it for this model (agp_svgp_desc.jitter) -- used by tests that compare with exact GP regression
        self.jitter = None if jitter is None else float(jitter)
        if not isinstance(likelihood, (GaussianLikelihood, LogisticLikelihood, StudentTLikelihood,
                                       LogisticSoftMaxLikelihood, _MultiOutputLikelihood, LaplaceLikelihood,
                                       BayesianSVM, PoissonLikelihood, NegBinomialLikelihood,
                                       HeteroscedasticLikelihood)):
            raise RuntimeError(f"The {likelihood} is not compatible or implemented with the {inference}")  # :48-49
        if optimiser is None:
            optimiser = ADAM(0.01)                       # SVGP.jl:39
        if isinstance(optimiser, bool):
            optimiser = ADAM(0.001) if optimiser else None  # SVGP.jl:51-53
        if isinstance(Zoptimiser, bool):
            Zoptimiser = ADAM(0.001) if Zoptimiser else None  # SVGP.jl:61-65
        for o in (optimiser, Zoptimiser):
            if o is not None and not isinstance(o, (ADAM, Descent, Momentum)):
                # the reference hands any Optimisers.jl rule to Optimisers.apply (autotuning_utils.jl:47-82); the device carries these
                raise NotImplementedError("hyper-parameter optimisers on the device: ADAM, Descent, Momentum")
        adams = [o for o in (optimiser, Zoptimiser) if isinstance(o, ADAM)]
        if len(adams) == 2 and (adams[0].beta != adams[1].beta or adams[0].eps != adams[1].eps):
            raise NotImplementedError("optimiser and Zoptimiser share one pair of ADAM moments / epsilon on the device")
        self.k_opt, self.z_opt = optimiser, Zoptimiser
        # A non-zero prior mean together with hyper-parameter optimisation: the reference constructs such a model and runs until
        # its first hyper step (n_iter >= 3, training.jl:65-69), where the prior-mean update calls update!(mu0, grad, state) against
        # the method update!(mu0, state, grad) (src/mean/constantmean.jl:31 vs autotuning.jl:104-106) and cannot run.  Mirrored:
        # construction, training up to that point, prediction and load_trained_model work; the first hyper step raises
        # (train_, _hyper_step_guard).
        if mean is not None and not (np.isscalar(mean) or isinstance(mean, (list, np.ndarray))):
            raise TypeError("mean must be None (ZeroMean), a Real (ConstantMean) or a vector (EmpiricalMean)")
        self.likelihood = likelihood
        self.inference = inference
        self.verbose = verbose
        self.atfrequency = atfrequency
        self.trained = False
        self.T = np.dtype(T)
        if self.T not in (np.dtype(np.float64), np.dtype(np.float32)):
            raise TypeError("T must be Float64 or Float32")
        Zlist = None
        if isinstance(Z, (list, tuple)):
            arrs = [np.asarray(z, dtype=np.float64) for z in Z]
            if arrs and all(a.ndim == 1 for a in arrs):  # the reference's own form: a vector of m points (SVGP.jl:36)
                Z = np.stack(arrs)
            else:                                          # one set of inducing points per latent (MOSVGP)
                Zlist = arrs
                Z = Zlist[0]
        Z = np.asarray(Z, dtype=np.float64)
        if Z.ndim != 2:
            raise ValueError("Z must be an (m, D) array of inducing points")
        self.n_latent_total = likelihood.n_latent
        # latent-parallel runs hold a slice [lo, hi) of the latents on this rank
        lo, hi = latent_slice if latent_slice is not None else (0, self.n_latent_total)
        self.latent_offset, self.n_latent = lo, hi - lo
        kernels = kernel if isinstance(kernel, (list, tuple)) else [kernel] * self.n_latent_total
        for k in kernels:
            if not isinstance(k, Kernel):
                raise TypeError("kernel must be a KernelFunctions-style kernel object")
        import copy

        # each latent owns a deep copy of kernel and Z (latentgp.jl:63-68)
        self.kernels = [copy.deepcopy(kernels[lo + i]) for i in range(self.n_latent)]
        self.Zs = [z.copy() for z in Zlist[lo:hi]] if Zlist is not None else [Z.copy() for _ in range(self.n_latent)]
        if any(z.shape != Z.shape for z in self.Zs):
            raise ValueError("all latents must have the same number of inducing points")
        self.mean = mean
        self.m, self.D = Z.shape
        self.elbo_mode = elbo_mode
        self.device = device
        self.rng = np.random.default_rng(seed)
        self._ctx = None
        self._h = None
        self._max_batch = 0
        self._keep = []
        self._data = None  # (X_dev, y_dev, N)

    # ---- device handle management ---------------------------------------------------------------------------
    @property
    def tdtype(self):
        torch = _torch()
        return torch.float64 if self.T == np.dtype(np.float64) else torch.float32

    def _dev(self):
        torch = _torch()
        if not torch.cuda.is_available():
            raise RuntimeError("no HIP device visible: augmentedgaussianprocesses.jl_amd has no CPU fallback")
        d = self.device if self.device is not None else torch.cuda.current_device()
        return torch.device("cuda", d)

    def _ensure_ctx(self):
        if self._ctx is None:
            torch = _torch()
            dev = self._dev()
            L = capi.lib()
            ctx = C.c_void_p()
            stream = torch.cuda.current_stream(dev).cuda_stream
            st = L.agp_ctx_create(dev.index, C.c_void_p(stream), C.byref(ctx))
            if st != capi.AGP_OK:
                raise capi.AGPError(st, "agp_ctx_create failed")
            self._ctx = ctx
        return self._ctx

    def _chk(self, st):
        capi.check(self._ctx, st)

    def _ensure_handle(self, max_batch: int):
        """(Re)create the device handle when a larger batch than ever before is requested; state is carried."""
        torch = _torch()
        L = capi.lib()
        ctx = self._ensure_ctx()
        if self._h is not None and max_batch <= self._max_batch:
            return self._h
        old = None
        carry = {}
        if self._h is not None:
            old = [self.get_state(i) for i in range(self.n_latent)]
            self._pre_destroy()
            n_opt = C.c_int64()
            self._chk(L.agp_svgp_get_opt_state(self._h, C.byref(n_opt)))
            # the rest of the training state travels too: LogisticSoftMax alpha (carried between minibatches) and the ADAM
            # moments of the kernel-parameter optimisers (the Z optimiser's device state restarts, like a new parameter array)
            if isinstance(self.likelihood, LogisticSoftMaxLikelihood):
                carry["alpha"] = self.get_matrix(capi.VEC_ALPHA, 0, self._max_batch)
            if self.k_opt is not None:
                carry["adam"] = []
                for i in range(self.n_latent):
                    km, kv, ks = (C.c_double * (1 + self.D))(), (C.c_double * (1 + self.D))(), C.c_int32()
                    self._chk(L.agp_svgp_hyper_opt_state(self._h, i, 0, km, kv, C.byref(ks)))
                    carry["adam"].append((km, kv, ks))
            L.agp_svgp_destroy(self._h)
            self._h = None
        d = capi.SvgpDesc()
        d.dtype = capi.F64 if self.T == np.dtype(np.float64) else capi.F32
        d.n_latent = self.n_latent
        d.latent_offset = self.latent_offset
        d.stochastic = 1 if self.inference.stoch else 0
        d.m, d.D, d.max_batch = self.m, self.D, int(max_batch)
        d.lik = self.likelihood.lik_desc()
        d.jitter = 0.0 if getattr(self, "jitter", None) is None else self.jitter
        opt = self.inference.optimiser or RobbinsMonro()
        d.rm_kappa, d.rm_tau = opt.kappa, opt.tau
        d.elbo_mode = capi.ELBO_REFERENCE if self.elbo_mode == "reference" else capi.ELBO_CORRECTED
        d.flags = capi.FLAG_STALE_K if self.reference_compat_stale_K else 0
        h = C.c_void_p()
        self._chk(L.agp_svgp_create(ctx, C.byref(d), C.byref(h)))
        self._h = h
        self._max_batch = int(max_batch)
        dev = self._dev()
        for i in range(self.n_latent):
            kd, keep = self.kernels[i].desc(self.D)
            self._chk(L.agp_svgp_set_kernel(h, i, C.byref(kd)))
            z = torch.as_tensor(self.Zs[i], dtype=self.tdtype, device=dev).contiguous()
            self._chk(L.agp_svgp_set_Z(h, i, C.c_void_p(z.data_ptr()), self.D))
            if self.mean is not None:
                mu0 = np.full(self.m, float(self.mean)) if np.isscalar(self.mean) else np.asarray(self.mean, float)
                t = torch.as_tensor(mu0, dtype=self.tdtype, device=dev)
                self._chk(L.agp_svgp_set_prior_mean(h, i, C.c_void_p(t.data_ptr())))
            torch.cuda.synchronize(dev)
        self._post_create(h)
        if old is not None:
            for i, (mu, Sig, e1, e2) in enumerate(old):
                self.set_state(i, e1, e2)
            self._chk(L.agp_svgp_set_opt_state(h, n_opt.value))
            if "alpha" in carry:
                a = torch.as_tensor(carry["alpha"], dtype=self.tdtype, device=dev).contiguous()
                self._chk(L.agp_svgp_set_lsm_alpha(h, C.c_void_p(a.data_ptr()), a.numel()))
                self._chk(L.agp_ctx_sync(ctx))
            for i, (km, kv, ks) in enumerate(carry.get("adam", [])):
                if ks.value > 0:
                    self._chk(L.agp_svgp_hyper_opt_state(h, i, 1, km, kv, C.byref(ks)))
        return h

    def _post_create(self, h):
        if isinstance(self.likelihood, PoissonLikelihood):  # the lambda update integrates logistic by Gauss-Hermite
            nodes, weights = _gauss_hermite()
            self._chk(capi.lib().agp_svgp_set_quadrature(h, nodes.ctypes.data_as(C.POINTER(C.c_double)),
                                                         weights.ctypes.data_as(C.POINTER(C.c_double)), len(nodes)))
        if getattr(self, "_batch_shard", None) is not None:  # a re-created handle keeps its place in a batch-parallel run
            self._chk(capi.lib().agp_svgp_set_batch_shard(h, *self._batch_shard))
        if self.k_opt is not None or self.z_opt is not None:
            o = next((q for q in (self.k_opt, self.z_opt) if isinstance(q, ADAM)), ADAM())
            self._chk(capi.lib().agp_svgp_hyper_configure(
                h, 1 if self.k_opt else 0, self.k_opt.eta if self.k_opt else 0.0, 1 if self.z_opt else 0,
                self.z_opt.eta if self.z_opt else 0.0, o.beta[0], o.beta[1], o.eps))
            (kr, krho), (zr, zrho) = _opt_rule(self.k_opt), _opt_rule(self.z_opt)
            self._chk(capi.lib().agp_svgp_hyper_rule(h, kr, krho, zr, zrho))

    def _pre_destroy(self):
        self._pull_hypers()
        self._pull_lik_state()
        self._close_tickets()

    def _close_tickets(self):
        """objective_enqueue tickets belong to the device handle: before it is re-created (a larger batch than ever before) every
        open one is fetched, so that objective_fetch still returns its value afterwards (found by the ADVICE r04 review: the
        library's slot numbers of a destroyed handle answered AGP_ERR_INVALID)."""
        for tk, slot in list(getattr(self, "_tickets", {}).items()):
            if isinstance(slot, int):
                out, ready = C.c_double(), C.c_int32()
                self._chk(capi.lib().agp_svgp_elbo_fetch(self._h, slot, 1, C.byref(out), C.byref(ready)))
                self._tickets[tk] = float(out.value)

    def _pull_lik_state(self):
        """λ of PoissonLikelihood / HeteroscedasticLikelihood lives on the device while training; mirror it back"""
        if self._h is not None and (isinstance(self.likelihood, (PoissonLikelihood, HeteroscedasticLikelihood)) or
                                    getattr(self.likelihood, "noise_eta", 0.0)):
            v = C.c_double()
            self._chk(capi.lib().agp_svgp_get_lik_param(self._h, C.byref(v)))
            self.likelihood.lam = v.value

    def _pull_hypers(self):
        """copy the (possibly optimised) kernel parameters and inducing points back into the Python objects"""
        if self._h is None or not (self.k_opt or self.z_opt):
            return
        from .kernels import ARDTransform, ScaleTransform

        torch = _torch()
        L = capi.lib()
        for i in range(self.n_latent):
            var = C.c_double()
            sc = (C.c_double * self.D)()
            self._chk(L.agp_svgp_get_kernel(self._h, i, C.byref(var), sc))
            k = self.kernels[i]
            if k.has_variance:  # only parameters that exist in the kernel object are ever stepped (and written back)
                k.variance = var.value
            if isinstance(k.transform, ARDTransform):
                k.transform = ARDTransform(list(sc))
            elif k.transform is not None:
                k.transform = ScaleTransform(sc[0])
            z = torch.empty(self.m, self.D, dtype=self.tdtype, device=self._dev())
            self._chk(L.agp_svgp_get_Z(self._h, i, C.c_void_p(z.data_ptr()), self.D))
            self._chk(L.agp_ctx_sync(self._ctx))
            self.Zs[i] = z.cpu().numpy().astype(np.float64)

    def hypergrad(self, latent: int = 0):
        """(d variance, d scales[D], dZ[m, D]) of the hyper objective on the last minibatch (autotuning.jl:96-98)."""
        torch = _torch()
        dv = C.c_double()
        ds = (C.c_double * self.D)()
        dz = torch.empty(self.m, self.D, dtype=self.tdtype, device=self._dev())
        self._chk(capi.lib().agp_svgp_hypergrad(self._h, latent, C.byref(dv), ds, C.c_void_p(dz.data_ptr())))
        return dv.value, np.array(list(ds)), dz.cpu().numpy()

    @property
    def n_out(self):
        return self.n_latent

    def _treat(self, y):
        return treat_labels(y, self.likelihood)

    def __del__(self):
        try:
            if self._h is not None:
                capi.lib().agp_svgp_destroy(self._h)
            if self._ctx is not None:
                capi.lib().agp_ctx_destroy(self._ctx)
        except Exception:
            pass

    # ---- data ---------------------------------------------------------------------------------------------
    def _upload(self, X, obsdim=1):
        """wrap_X (datacontainer.jl:64-74): N x D (obsdim=1) or D x N (obsdim=2) -> point-major device tensor."""
        torch = _torch()
        dev = self._dev()
        if isinstance(X, torch.Tensor):
            Xt = X.to(device=dev, dtype=self.tdtype)
        else:
            Xt = torch.as_tensor(np.asarray(X), dtype=self.tdtype, device=dev)
        if Xt.ndim == 1:
            Xt = Xt[:, None]
        if obsdim == 2:
            Xt = Xt.t()
        Xt = Xt.contiguous()
        if Xt.shape[1] != self.D:
            raise ValueError(f"data has {Xt.shape[1]} features, inducing points have {self.D}")
        return Xt

    def _upload_y(self, y_treated):
        torch = _torch()
        dev = self._dev()
        if isinstance(self.likelihood, LogisticSoftMaxLikelihood):
            return torch.as_tensor(class_indices(y_treated), dtype=torch.int32, device=dev)
        return torch.as_tensor(y_treated, dtype=self.tdtype, device=dev)

    # ---- state export / import ----------------------------------------------------------------------------
    def get_state(self, latent: int = 0):
        """(μ, Σ, η₁, η₂) of one latent as numpy arrays (VarPosterior, posterior.jl:21-27)."""
        torch = _torch()
        dev = self._dev()
        m = self.m
        mu = torch.empty(m, dtype=self.tdtype, device=dev)
        e1 = torch.empty(m, dtype=self.tdtype, device=dev)
        Sig = torch.empty(m, m, dtype=self.tdtype, device=dev)
        e2 = torch.empty(m, m, dtype=self.tdtype, device=dev)
        self._chk(capi.lib().agp_svgp_get_state(self._h, latent, C.c_void_p(mu.data_ptr()), C.c_void_p(Sig.data_ptr()),
                                                C.c_void_p(e1.data_ptr()), C.c_void_p(e2.data_ptr())))
        self._chk(capi.lib().agp_ctx_sync(self._ctx))
        return mu.cpu().numpy(), Sig.cpu().numpy(), e1.cpu().numpy(), e2.cpu().numpy()

    def set_state(self, latent: int, eta1, eta2):
        torch = _torch()
        dev = self._dev()
        e1 = torch.as_tensor(np.asarray(eta1), dtype=self.tdtype, device=dev).contiguous()
        e2 = torch.as_tensor(np.asarray(eta2), dtype=self.tdtype, device=dev).contiguous()
        self._chk(capi.lib().agp_svgp_set_state(self._h, latent, C.c_void_p(e1.data_ptr()), C.c_void_p(e2.data_ptr())))
        self._chk(capi.lib().agp_svgp_check_status(self._h))

    def get_matrix(self, which: int, latent: int = 0, rows: Optional[int] = None):
        """rows: how many rows / elements of a batch-sized output to fetch (default: the whole last batch).  The library
        refuses a buffer smaller than the batch the handle last saw (an ELBO on a larger set also counts)."""
        torch = _torch()
        dev = self._dev()
        m = self.m
        if rows is None and which not in (capi.MAT_L, capi.MAT_KINV):
            nb = C.c_int64()
            self._chk(capi.lib().agp_svgp_last_batch(self._h, C.byref(nb)))
            rows = int(nb.value) if which != capi.VEC_ALPHA else self._max_batch
        if which in (capi.MAT_L, capi.MAT_KINV):
            out = torch.empty(m, m, dtype=self.tdtype, device=dev)
            ld = m
            rows = m
        elif which in (capi.MAT_KNM, capi.MAT_KAPPA):
            out = torch.empty(rows, m, dtype=self.tdtype, device=dev)
            ld = m
        else:
            out = torch.empty(rows, dtype=self.tdtype, device=dev)
            ld = 1
        self._chk(capi.lib().agp_svgp_get_matrix(self._h, latent, which, C.c_void_p(out.data_ptr()), ld, int(rows)))
        self._chk(capi.lib().agp_ctx_sync(self._ctx))
        return out.cpu().numpy()

    def __repr__(self):
        return f"Sparse Variational Gaussian Process with a {self.likelihood} infered by {self.inference} "


class MOSVGP(SVGP):
    """Multi-Output Sparse Variational GP (src/models/MOSVGP.jl:22-115): Q = len(Zs) latent GPs mixed into
    len(likelihoods) outputs by the weights A[t][q] (random unit vectors by default, MOSVGP.jl:101-104).

    y is a list with one target vector per task.  Aoptimiser: ADAM(...) or False (update_A!,
    single_and_multi_output_utils.jl:87-118).  The reference mixes up n_output and the number of tasks (Appendix A Q7) and
    only works for Q == n_task; this follows the documented intent and accepts any Q."""

    def __init__(self, kernel, likelihoods, inference, Zs, *, Aoptimiser=None, A=None, verbose: int = 0,
                 optimiser=False, atfrequency: int = 1, mean=None, Zoptimiser=False, T=np.float64,
                 device: Optional[int] = None, seed: Optional[int] = None, elbo_mode: str = "corrected",
                 latent_slice: Optional[tuple] = None):
        if not isinstance(inference, AnalyticVI):
            raise TypeError("The inference object should be of type `AnalyticVI`")  # MOSVGP.jl:55
        Zs = [np.asarray(z, dtype=np.float64) for z in Zs]
        Q = len(Zs)
        liks = list(likelihoods)
        kernels = list(kernel) if isinstance(kernel, (list, tuple)) else [kernel]
        kernels = [kernels[i % len(kernels)] for i in range(Q)]  # kernel[mod1(i, n_kernel)]  MOSVGP.jl:96-98
        super().__init__(kernels, _MultiOutputLikelihood(liks, Q), inference, Zs, verbose=verbose, optimiser=optimiser,
                         atfrequency=atfrequency, mean=mean, Zoptimiser=Zoptimiser, T=T, device=device, seed=seed,
                         elbo_mode=elbo_mode, latent_slice=latent_slice)
        # latent_slice=(lo, hi): this rank owns latents [lo, hi) of the Q (parallel.latent_parallel_step exchanges their
        # mean_f / var_f); A stays (n_task, Q) and replicated -- pass the same A (or the same seed) on every rank
        self.sharded = latent_slice is not None
        self.n_task = len(liks)
        if Aoptimiser is None:
            Aoptimiser = ADAM(0.01)  # MOSVGP.jl:42
        self.A_opt = Aoptimiser if isinstance(Aoptimiser, ADAM) else (ADAM(0.01) if Aoptimiser is True else None)
        if A is None:
            A = self.rng.standard_normal((self.n_task, Q))
            A = A / np.linalg.norm(A, axis=1, keepdims=True)
        self.A = np.array(A, dtype=np.float64, order="C", copy=True)
        if self.A.shape != (self.n_task, Q):
            raise ValueError("A must be (n_task, n_latent)")

    @property
    def n_out(self):
        return self.n_task

    def _treat(self, y):
        ys = [treat_labels(yt, l) for yt, l in zip(y, self.likelihood.likelihoods)]
        if len(ys) != self.n_task or len({len(v) for v in ys}) != 1:
            raise ValueError("y must hold one target vector per task, all of the same length")
        return np.stack(ys)

    def _upload_y(self, y_treated):
        """device layout is point-major: y[i * n_task + t]"""
        torch = _torch()
        return torch.as_tensor(np.ascontiguousarray(y_treated.T), dtype=self.tdtype, device=self._dev())

    def _post_create(self, h):
        super()._post_create(h)  # hyper-parameter optimiser configuration
        liks = (capi.LikDesc * self.n_task)(*[l.lik_desc() for l in self.likelihood.likelihoods])
        o = self.A_opt
        if self.sharded:
            self._chk(capi.lib().agp_svgp_mo_shard(h, self.n_latent_total))
        self._chk(capi.lib().agp_svgp_set_multioutput(
            h, self.n_task, liks, self.A.ctypes.data_as(C.POINTER(C.c_double)), o.eta if o else 0.0,
            o.beta[0] if o else 0.9, o.beta[1] if o else 0.999, o.eps if o else 1e-8))

    def _pre_destroy(self):
        super()._pre_destroy()
        self.get_A()  # carry the mixing weights into the re-created handle (the ADAM moments restart)

    def get_A(self):
        if self._h is not None:
            self._chk(capi.lib().agp_svgp_get_A(self._h, self.A.ctypes.data_as(C.POINTER(C.c_double))))
        return self.A.copy()

    def __repr__(self):
        return (f"Multioutput Sparse Variational Gaussian Process with the likelihoods {self.likelihood} "
                f"infered by {self.inference} ")


# ---- training (src/training/training.jl:13-111) ------------------------------------------------------------------
def train_(model: SVGP, X, y, iterations: int = 100, *, callback: Optional[Callable] = None, convergence=None,
           state: Optional[State] = None, obsdim: int = 1, idx_stream: Optional[Sequence] = None):
    """train!(model, X, y, iterations; callback, state, obsdim).  Runs a FIXED number of iterations like the
    reference (ϵ / convergence are never read there, training.jl:48,93-94).

    idx_stream: optional pre-generated minibatch indices (one int array per iteration) replacing
    StatsBase.sample(1:N, B; replace=false) (training.jl:51-53) so runs are reproducible across back-ends.
    """
    torch = _torch()
    L = capi.lib()
    if not iterations > 0:
        raise ValueError("Number of iterations should be positive")
    Xd = model._upload(X, obsdim)
    yt = model._treat(y)
    N = Xd.shape[0]
    if (yt.shape[-1] if isinstance(model, MOSVGP) else len(yt)) != N:
        raise ValueError(f"There is not the same number of samples in X ({N}) and y ({len(yt)})")
    inf = model.inference
    if inf.stoch:
        if not (0 < inf.batchsize <= N):
            raise ValueError(f"The size of mini-batch {inf.batchsize} is incorrect (negative or bigger than number "
                             "of samples), please set `batchsize` correctly in the inference object")
        inf.rho = N / inf.batchsize
    else:
        inf.batchsize = N
        inf.rho = 1.0
    B = inf.batchsize
    yd = model._upload_y(yt)
    h = model._ensure_handle(B)
    model._data = (Xd, yd, N)
    dev = model._dev()
    if state is None:
        inf.HyperParametersUpdated = True
        model._chk(L.agp_svgp_init_state(h))  # init_state(model), training.jl:41-45: counters, local variables, optimiser states
    else:
        model._chk(L.agp_svgp_invalidate_data(h))  # X / y may be the caller's buffers, possibly refilled since the last call
    model._chk(L.agp_svgp_refresh_K(h))
    local_iter = 1

    CH = 64  # minibatch indices are drawn on the host like the reference does, but uploaded 64 iterations at a time: one
    #          blocking host->device copy per iteration would stall the launch queue for longer than the step itself
    chunk = {"base": -1, "dev": None}

    def draw(it):  # StatsBase.sample(1:N, B; replace=false)  training.jl:51-53 (or the caller's stream)
        c0 = (it - 1) // CH * CH
        if chunk["base"] != c0:
            n_here = min(CH, iterations - c0)
            rows = []
            for q in range(c0, c0 + n_here):
                if idx_stream is not None:
                    idx_np = np.asarray(idx_stream[q], dtype=np.int64)
                    if idx_np.shape != (B,):
                        raise ValueError("idx_stream entries must have length batchsize")
                else:
                    idx_np = model.rng.choice(N, B, replace=False).astype(np.int64)
                rows.append(idx_np)
            prev = chunk["dev"]
            chunk["dev"] = torch.as_tensor(np.stack(rows), device=dev)
            chunk["base"] = c0
            model._keep_chunks = [prev, chunk["dev"]]  # the previous chunk may still be referenced by a queued prefetch
        return chunk["dev"][it - 1 - c0]

    nxt = draw(1) if inf.stoch else None
    while True:
        stepped = False  # this iteration's variational update has been enqueued (the device's own counters have moved on)
        try:
            if inf.stoch:
                idx = nxt
                idx_ptr = C.c_void_p(idx.data_ptr())
                model._keep = [idx]
            else:
                idx_ptr = None
            model._chk(L.agp_svgp_cavi_step(h, C.c_void_p(Xd.data_ptr()), Xd.stride(0), C.c_void_p(yd.data_ptr()),
                                            idx_ptr, B, inf.rho))
            stepped = True
            model.trained = True
            model._last_idx = idx_ptr
            hyper_on = model.k_opt is not None or model.z_opt is not None
            if inf.stoch and local_iter < iterations:
                nxt = draw(local_iter + 1)
                model._keep.append(nxt)
                if not hyper_on:  # look-ahead: next minibatch's kappa on the second stream (pointless if K is about to change)
                    model._chk(L.agp_svgp_prefetch(h, C.c_void_p(Xd.data_ptr()), Xd.stride(0),
                                                   C.c_void_p(nxt.data_ptr()), B))
            if callback is not None:
                callback(model, State(model), inf.n_iter)
            # training.jl:65-69 (n_iter is the counter before this iteration's increment)
            if hyper_on and inf.n_iter % model.atfrequency == 0 and inf.n_iter >= 3 and local_iter != iterations:
                _hyper_step_guard(model)
                model._chk(L.agp_svgp_hyper_step(h))
            if model.verbose > 2 or (model.verbose > 1 and local_iter % 10 == 0):
                # (the reference drives a ProgressMeter with the same two values, training.jl:71-90)
                print(f"iter {local_iter}  ELBO {objective(model, State(model), None):.6f}")
            local_iter += 1
            inf.n_iter += 1
            if local_iter > iterations:
                break
        except KeyboardInterrupt:
            # training.jl:95-101: InterruptException -> warn, leave the loop, still finish with compute_Ks.  The work enqueued so far
            # stays valid: a variational update that was already issued counts as done (its Robbins-Monro step has been taken on the
            # device), the pending natural-gradient step is taken by check_status below, and `state=` continues from there.
            import warnings

            warnings.warn(f"Training interrupted by user at iteration {local_iter}")
            if stepped:
                local_iter += 1
                inf.n_iter += 1
            break
    if model.verbose > 0:  # training.jl:103-105
        print(f"Training ended after {local_iter - 1} iterations. Total number of iterations {inf.n_iter}")
    model._chk(L.agp_svgp_check_status(h))
    model._chk(L.agp_svgp_refresh_K(h))  # compute_Ks(model), training.jl:107: final kernel matrices for predictions
    model._pull_hypers()
    model._pull_lik_state()
    return model, State(model)


def _hyper_step_guard(model):
    """update_hyperparameters! with a non-zero prior mean: the reference's prior-mean update cannot run (see SVGP.__init__)"""
    if model.mean is not None:
        raise NotImplementedError("a non-zero prior mean together with hyper-parameter optimisation is not wired: the "
                                  "reference's prior-mean update (autotuning.jl:104-106) is broken; pass optimiser=False, "
                                  "Zoptimiser=False or mean=None")


def objective(model: SVGP, state: Optional[State] = None, y=None) -> float:
    """objective(model, state, y) = ELBO(model, state, y) on the last minibatch (SVGP.jl:90, analyticVI.jl:255-274)."""
    L = capi.lib()
    Xd, yd, N = model._data
    out = C.c_double()
    idx_ptr = getattr(model, "_last_idx", None)
    model._chk(L.agp_svgp_elbo(model._h, C.c_void_p(Xd.data_ptr()), Xd.stride(0), C.c_void_p(yd.data_ptr()), idx_ptr,
                               model.inference.batchsize, model.inference.rho, 0, C.byref(out)))
    return out.value


def objective_enqueue(model: SVGP) -> int:
    """`objective(model, state, y)` put into the stream without waiting for it (agp_svgp_elbo_enqueue): returns a ticket for
    `objective_fetch`.  For convergence monitoring inside a training loop: the next iterations are enqueued while the value is
    on its way (up to 8 tickets in flight)."""
    L = capi.lib()
    Xd, yd, N = model._data
    t = C.c_int32()
    idx_ptr = getattr(model, "_last_idx", None)
    model._chk(L.agp_svgp_elbo_enqueue(model._h, C.c_void_p(Xd.data_ptr()), Xd.stride(0), C.c_void_p(yd.data_ptr()), idx_ptr,
                                       model.inference.batchsize, model.inference.rho, 0, C.byref(t)))
    # tickets are the mirror's own numbers: ticket -> the library's slot of the current handle, or the value itself once the
    # handle it belonged to has been re-created (_close_tickets)
    if not hasattr(model, "_tickets"):
        model._tickets, model._next_ticket = {}, 0
    tk = model._next_ticket
    model._next_ticket += 1
    model._tickets[tk] = int(t.value)
    return tk


def objective_fetch(model: SVGP, ticket: int, wait: bool = True):
    """the value of an `objective_enqueue` ticket; with wait=False: None while it has not arrived yet.  Tickets may be fetched in
    any order; one that outlived its device handle (the model was re-created for a larger batch in between) still has its value."""
    tickets = getattr(model, "_tickets", {})
    if ticket not in tickets:
        raise KeyError(f"objective_fetch: ticket {ticket} is not open (already fetched, or not from this model)")
    slot = tickets[ticket]
    if isinstance(slot, float):
        del tickets[ticket]
        return slot
    out, ready = C.c_double(), C.c_int32()
    model._chk(capi.lib().agp_svgp_elbo_fetch(model._h, slot, 1 if wait else 0, C.byref(out), C.byref(ready)))
    if not ready.value:
        return None
    del tickets[ticket]
    return out.value


class SideObjective:
    """Convergence monitoring NEXT TO the training stream (round 6).

    `objective` / `objective_enqueue` evaluate the ELBO in line on the model's own stream: at C2 a check on a fixed 8192-point
    batch costs the training loop 0.87 ms (kernel matrices of the evaluation batch, local update, the factorisation of -2 eta2 with
    its inverse, the ELBO's reductions) between two 0.31 ms steps.  A check needs nothing of the training state but a SNAPSHOT of
    (eta1, eta2): this object owns a shadow model -- same kernels, inducing points, likelihood and prior mean on a context and
    stream of its own, its K_ZZ factored once -- and a ring of snapshot buffers.  `enqueue` copies (eta1, eta2) out of the training
    handle on the training stream (two device copies; agp_svgp_get_state), lets the side stream wait for them with an event, and
    there installs them (agp_svgp_set_state: the same factorisation launch `objective` uses) and enqueues the evaluation
    (agp_svgp_elbo_enqueue, fresh local variables: ELBO(model, X, y) of src/functions/ELBO.jl:32-47 with rho explicit).  The
    training stream goes straight on to its next steps; the steps are bound by one workgroup's dependent chain while ~250 CUs
    idle, which is where the evaluation runs.  `fetch` returns the value (tickets in any order).
    What makes it pay: the shadow handle does nothing but evaluate, on the same batch every time, with kernels and inducing points
    fixed -- so the library keeps K_nm and kappa = K_nm K^-1 of the evaluation batch between the checks (pointer identity of X / idx,
    the contract of the full-batch kappa cache; 17 GF of 34 per check at 8192 x 1024), which an in-line evaluation cannot: the
    training steps in between own those buffers.  C2: time_to_elbo_tol_reachable 0.674 s against 0.758 s in line (same 1900
    iterations, same ELBO values to rounding).

    The values equal `ELBO(model, X[idx], y[idx], rho=rho)` evaluated in line at the same point of the training sequence to a few
    ulp (the two handles form Sigma = Xa' Xa by different kernels); the training trajectory with snapshots equals the one without to
    rounding (a snapshot takes the pending natural-gradient step with the stand-alone kernel; tests/test_gpu_round6.py).  Models whose kernels / inducing points move (hyper-parameter optimisation on) are
    refused: the shadow's K_ZZ is factored once.  Reference: the monitoring this replaces is `objective(model, state, y)` in
    train!'s progress reporting, src/training/training.jl:71-90, and the ELBO itself, src/inference/analyticVI.jl:255-274."""

    def __init__(self, model: SVGP, max_eval_batch: int, ring: int = 4, priority: Optional[int] = None):
        if isinstance(model, MOSVGP) or model.n_latent != 1:
            raise NotImplementedError("SideObjective: single-latent SVGP models")
        if model.k_opt is not None or model.z_opt is not None:
            raise NotImplementedError("SideObjective: kernels and inducing points must be fixed (optimiser=False, Zoptimiser=False)")
        if model._h is None:
            raise RuntimeError("SideObjective: the model has no device state yet (train it for at least one step, or bind data)")
        torch = _torch()
        self.model = model
        dev = model._dev()
        # priority of the side stream (None: the default): a check is ~34 GF of MFMA work at C2; whether it runs NEXT TO the training
        # steps or between them is the dispatcher's decision, see DESIGN.md section 10
        self.stream = torch.cuda.Stream(device=dev) if priority is None else torch.cuda.Stream(device=dev, priority=int(priority))
        import copy

        with torch.cuda.stream(self.stream):  # (the shadow's context takes torch's current stream: the side stream)
            inf = AnalyticSVI(int(max_eval_batch)) if model.inference.stoch else AnalyticVI()
            self.shadow = SVGP(copy.deepcopy(model.kernels[0]), copy.deepcopy(model.likelihood), inf, model.Zs[0], optimiser=False,
                               Zoptimiser=False, mean=model.mean, T=model.T.type, device=dev.index, elbo_mode=model.elbo_mode,
                               jitter=model.jitter)
            self.h = self.shadow._ensure_handle(int(max_eval_batch))
            self.shadow._chk(capi.lib().agp_svgp_refresh_K(self.h))
        self.stream.synchronize()
        m = model.m
        self._ring = [(torch.empty(m, dtype=model.tdtype, device=dev), torch.empty(m, m, dtype=model.tdtype, device=dev),
                       torch.cuda.Event(), torch.cuda.Event()) for _ in range(int(ring))]
        self._n = 0
        self._tickets = {}

    def enqueue(self, Xd, yd, idx_dev, B: int, rho: float) -> int:
        """Snapshot the training model's (eta1, eta2) now (in stream order) and evaluate ELBO on rows idx_dev[0..B) of the device
        arrays (Xd, yd) on the side stream.  Returns a ticket."""
        torch = _torch()
        L = capi.lib()
        mdl = self.model
        e1, e2, ev_snap, ev_used = self._ring[self._n % len(self._ring)]
        main = torch.cuda.current_stream(mdl._dev())
        if self._n >= len(self._ring):
            main.wait_event(ev_used)  # the side stream has installed the snapshot this slot held (set_state copies it out)
        mdl._chk(L.agp_svgp_get_state(mdl._h, 0, None, None, C.c_void_p(e1.data_ptr()), C.c_void_p(e2.data_ptr())))
        ev_snap.record(main)
        t = C.c_int32()
        with torch.cuda.stream(self.stream):
            self.stream.wait_event(ev_snap)
            self.shadow._chk(L.agp_svgp_set_state(self.h, 0, C.c_void_p(e1.data_ptr()), C.c_void_p(e2.data_ptr())))
            ev_used.record(self.stream)
            self.shadow._chk(L.agp_svgp_elbo_enqueue(self.h, C.c_void_p(Xd.data_ptr()), Xd.stride(0), C.c_void_p(yd.data_ptr()),
                                                     C.c_void_p(idx_dev.data_ptr()) if idx_dev is not None else None, int(B),
                                                     float(rho), 1, C.byref(t)))
        tk = self._n
        self._n += 1
        self._tickets[tk] = int(t.value)
        return tk

    def fetch(self, ticket: int, wait: bool = True):
        slot = self._tickets[ticket]
        out, ready = C.c_double(), C.c_int32()
        self.shadow._chk(capi.lib().agp_svgp_elbo_fetch(self.h, slot, 1 if wait else 0, C.byref(out), C.byref(ready)))
        if not ready.value:
            return None
        del self._tickets[ticket]
        return out.value


def ELBO(model: SVGP, X, y, *, obsdim: int = 1, rho: Optional[float] = None) -> float:
    """External ELBO(model, X, y) (src/functions/ELBO.jl:28-47): fresh local variables on (X, y), one local update.
    rho defaults to the reference's behaviour (the ρ left by the last train!, Appendix A Q13); pass rho=1 for the
    properly scaled full-data ELBO."""
    L = capi.lib()
    Xd = model._upload(X, obsdim)
    yt = model._treat(y)
    yd = model._upload_y(yt)
    n = Xd.shape[0]
    h = model._ensure_handle(max(n, model._max_batch))
    r = model.inference.rho if rho is None else float(rho)
    out = C.c_double()
    model._chk(L.agp_svgp_elbo(h, C.c_void_p(Xd.data_ptr()), Xd.stride(0), C.c_void_p(yd.data_ptr()), None, n, r, 1,
                               C.byref(out)))
    return out.value


# ---- prediction (src/training/predictions.jl) -----------------------------------------------------------------------
def _predict_f(model: SVGP, X_test, cov: bool, obsdim: int = 1):
    torch = _torch()
    L = capi.lib()
    Xd = model._upload(X_test, obsdim)
    nt = Xd.shape[0]
    h = model._ensure_handle(max(model._max_batch, 1))
    dev = model._dev()
    mu = torch.empty(model.n_out, nt, dtype=model.tdtype, device=dev)
    var = torch.empty(model.n_out, nt, dtype=model.tdtype, device=dev) if cov else None
    model._chk(L.agp_svgp_predict_f(h, C.c_void_p(Xd.data_ptr()), Xd.stride(0), nt, C.c_void_p(mu.data_ptr()),
                                    C.c_void_p(var.data_ptr()) if cov else None))
    model._chk(L.agp_ctx_sync(model._ctx))
    return mu, var


def _predict_f_fullcov(model: SVGP, X_test, obsdim: int = 1):
    """predict_f(...; cov=true, diag=false)  predictions.jl:45-49: (mu_f, Sigma_f) with the full n_t x n_t covariance per
    latent (K*m is materialised on the device: small n_t only)."""
    torch = _torch()
    L = capi.lib()
    Xd = model._upload(X_test, obsdim)
    nt = Xd.shape[0]
    h = model._ensure_handle(max(model._max_batch, 1))
    dev = model._dev()
    nout = model.n_out  # latents, or tasks of a multi-output model (mixed: sum_q A[t][q]^2 cov_q, predictions.jl:82-90)
    mu = torch.empty(nout, nt, dtype=model.tdtype, device=dev)
    cov = torch.empty(nout, nt, nt, dtype=model.tdtype, device=dev)
    model._chk(L.agp_svgp_predict_f_cov(h, C.c_void_p(Xd.data_ptr()), Xd.stride(0), nt, C.c_void_p(mu.data_ptr()),
                                        C.c_void_p(cov.data_ptr())))
    model._chk(L.agp_ctx_sync(model._ctx))
    mu_np, cov_np = mu.cpu().numpy(), cov.cpu().numpy()
    if nout > 1 or isinstance(model, MOSVGP):
        return tuple(mu_np), tuple(cov_np)
    return mu_np[0], cov_np[0]


def predict_f(model: SVGP, X_test, state: Optional[State] = None, *, cov: bool = False, diag: bool = True,
              obsdim: int = 1):
    """predict_f(model, X_test; cov=false, diag=true)  predictions.jl:141-164.  diag=False returns the full covariance
    (not streamed: K*m is materialised, n_t <= 8192)."""
    if cov and not diag:
        return _predict_f_fullcov(model, X_test, obsdim)
    mu, var = _predict_f(model, X_test, cov, obsdim)
    mu_np = mu.cpu().numpy()
    if model.n_out > 1 or isinstance(model, MOSVGP):
        m_out = tuple(mu_np[k] for k in range(model.n_out))
        if not cov:
            return m_out
        v_np = var.cpu().numpy()
        return m_out, tuple(v_np[k] for k in range(model.n_out))
    if not cov:
        return mu_np[0]
    return mu_np[0], var.cpu().numpy()[0]


def predict_y(model: SVGP, X_test, state: Optional[State] = None, *, obsdim: int = 1):
    """predict_y  predictions.jl:178-198: regression mean / Bool (μ_f > 0) / most likely class label."""
    torch = _torch()
    L = capi.lib()
    Xd = model._upload(X_test, obsdim)
    nt = Xd.shape[0]
    h = model._ensure_handle(max(model._max_batch, 1))
    dev = model._dev()
    lik = model.likelihood
    if isinstance(model, MOSVGP):
        out = torch.empty(model.n_task, nt, dtype=model.tdtype, device=dev)
    elif isinstance(lik, (GaussianLikelihood, StudentTLikelihood, LaplaceLikelihood, HeteroscedasticLikelihood,
                          PoissonLikelihood, NegBinomialLikelihood)):
        out = torch.empty(nt, dtype=model.tdtype, device=dev)
    else:
        out = torch.empty(nt, dtype=torch.int32, device=dev)
    model._chk(L.agp_svgp_predict_y(h, C.c_void_p(Xd.data_ptr()), Xd.stride(0), nt, C.c_void_p(out.data_ptr())))
    model._chk(L.agp_ctx_sync(model._ctx))
    o = out.cpu().numpy()
    if isinstance(model, MOSVGP):
        return [o[t] > 0.5 if isinstance(l, (LogisticLikelihood, BayesianSVM)) else o[t]
                for t, l in enumerate(lik.likelihoods)]
    if isinstance(lik, (LogisticLikelihood, BayesianSVM)):
        return o.astype(bool)
    if isinstance(lik, LogisticSoftMaxLikelihood):
        cm = lik.class_mapping or list(range(1, lik.n_class + 1))
        return np.array([cm[i] for i in o])
    return o


_GH = None


def _gauss_hermite():
    """pred_nodes, pred_weights = (x*√2, w/√π) of gausshermite(100)   predictions.jl:4."""
    global _GH
    if _GH is None:
        x, w = np.polynomial.hermite.hermgauss(100)
        _GH = (np.ascontiguousarray(x * math.sqrt(2.0)), np.ascontiguousarray(w / math.sqrt(math.pi)))
    return _GH


def proba_y(model: SVGP, X_test, state: Optional[State] = None, *, obsdim: int = 1):
    """proba_y  predictions.jl:225-247: (mean, var) for regression, (p, var) for Bernoulli, dict class -> p for
    multi-class."""
    torch = _torch()
    L = capi.lib()
    Xd = model._upload(X_test, obsdim)
    nt = Xd.shape[0]
    h = model._ensure_handle(max(model._max_batch, 1))
    dev = model._dev()
    lik = model.likelihood
    nodes, weights = _gauss_hermite()
    if isinstance(model, MOSVGP):
        o0 = torch.empty(model.n_task, nt, dtype=model.tdtype, device=dev)
        o1 = torch.empty(model.n_task, nt, dtype=model.tdtype, device=dev)
    elif isinstance(lik, LogisticSoftMaxLikelihood):
        o0 = torch.empty(nt, model.n_latent, dtype=model.tdtype, device=dev)
        o1 = None
    else:
        o0 = torch.empty(nt, dtype=model.tdtype, device=dev)
        o1 = torch.empty(nt, dtype=model.tdtype, device=dev)
    model._chk(L.agp_svgp_proba_y(h, C.c_void_p(Xd.data_ptr()), Xd.stride(0), nt,
                                  nodes.ctypes.data_as(C.POINTER(C.c_double)),
                                  weights.ctypes.data_as(C.POINTER(C.c_double)), len(nodes),
                                  C.c_void_p(o0.data_ptr()), C.c_void_p(o1.data_ptr()) if o1 is not None else None))
    model._chk(L.agp_ctx_sync(model._ctx))
    if isinstance(model, MOSVGP):
        a, b = o0.cpu().numpy(), o1.cpu().numpy()
        return [(a[t], b[t]) for t in range(model.n_task)]
    if isinstance(lik, LogisticSoftMaxLikelihood):
        p = o0.cpu().numpy()
        cm = lik.class_mapping or list(range(1, lik.n_class + 1))
        return {cm[model.latent_offset + k]: p[:, k] for k in range(model.n_latent)}
    return o0.cpu().numpy(), o1.cpu().numpy()
