"""OnlineSVGP (src/models/OnlineSVGP.jl, src/training/onlinetraining.jl) and the OIPS inducing-point rule it streams with
(InducingPoints.jl, unvendored; restated from Galy-Fajou & Opper 2021).

Host logic only.  A streaming model is a chain of device handles: every arriving batch may grow Z, so a fresh handle is
created for the new inducing points and the previous posterior is installed as its prior (`agp_svgp_set_online_prior`);
the old handle serves the first local update (under the old inducing points) and is then dropped.  All kernel values the
OIPS rule looks at come from `agp_kernelmatrix` on the device.
"""
from __future__ import annotations

import ctypes as C
from typing import Callable, Optional

import numpy as np

from . import capi
from .svgp import SVGP, AnalyticVI, _torch
from .svgp import predict_f as _predict_f_fn
from .svgp import predict_y as _predict_y_fn
from .svgp import proba_y as _proba_y_fn


class OIPS:
    """OIPS(ρ_accept=0.8; ρ_remove=1.0, kmin=10): a point joins Z when its largest kernel value with the current Z is below
    ρ_accept; with ρ_remove < 1, `remove_point` drops one of the inducing points that have a neighbour above ρ_remove,
    drawn with weights = number of such neighbours."""

    def __init__(self, rho_accept: float = 0.8, rho_remove: float = 1.0, kmin: int = 10):
        if not 0.0 < rho_accept <= 1.0:
            raise ValueError("ρ_accept should be between 0 and 1")
        self.rho_accept, self.rho_remove, self.kmin = float(rho_accept), float(rho_remove), int(kmin)


def _kmat(model_like, kernel, A, Bm):
    """kernelmatrix(kernel, A, B) on the device (agp_kernelmatrix), returned as a numpy array (host logic reads it)."""
    torch = _torch()
    L = capi.lib()
    dev, td = model_like._dev(), model_like.tdtype
    Ad = torch.as_tensor(A, dtype=td, device=dev).contiguous()
    Bd = torch.as_tensor(Bm, dtype=td, device=dev).contiguous()
    D = Ad.shape[1]
    kd, keep = kernel.desc(D)
    out = torch.empty(Ad.shape[0], Bd.shape[0], dtype=td, device=dev)
    dt = capi.F64 if td == torch.float64 else capi.F32
    model_like._chk(L.agp_kernelmatrix(model_like._ensure_ctx(), dt, C.byref(kd), C.c_void_p(Ad.data_ptr()), Ad.shape[0], D,
                                       None, C.c_void_p(Bd.data_ptr()), Bd.shape[0], D, D, C.c_void_p(out.data_ptr()),
                                       out.stride(0)))
    model_like._chk(L.agp_ctx_sync(model_like._ctx))
    return out.cpu().numpy().astype(np.float64)


def _oips_update(alg: OIPS, host, kernel, Z, X):
    """updateZ(Z, alg, X; kernel): the sequential scan of the reference, on device-computed kernel values: k(X, Z) once,
    k(X_cand, X_cand) for the points that pass the first test (only they can ever be accepted)."""
    X = np.asarray(X, dtype=np.float64)
    if len(X) == 0:
        return Z
    kxz = _kmat(host, kernel, X, Z)
    cand = np.flatnonzero(kxz.max(axis=1) < alg.rho_accept)
    if len(cand) == 0:
        return Z
    kcc = _kmat(host, kernel, X[cand], X[cand])
    acc = []
    for a, _ in enumerate(cand):
        if not acc or np.max(kcc[a, acc]) < alg.rho_accept:
            acc.append(a)
    return np.concatenate([Z, X[cand[acc]]], axis=0)


def _oips_remove(alg: OIPS, rng, Z, Kmat):
    if alg.rho_remove >= 1.0:
        return Z
    overlap = np.sum(Kmat > alg.rho_remove, axis=1) - 1
    removable = np.flatnonzero(overlap > 0)
    if len(removable) > 1 and len(Z) > alg.kmin:
        w = overlap[removable].astype(np.float64)
        gone = removable[rng.choice(len(removable), p=w / w.sum())]
        return np.delete(Z, gone, axis=0)
    return Z


_FAR = 1.0e6  # spacing of the neutral padding points (see _pad_inducing)


def _pad_inducing(Zs):
    """The latents of one device handle share m, the reference's OnlineVarLatents do not (every latent runs its own OIPS with its own
    kernel, onlinetraining.jl:153-160).  Latents with fewer inducing points are filled up with points that are far from the data
    and from one another: every kernel value that involves such a point is exactly 0 in floating point, so K is block diagonal
    with 1x1 blocks for them, their kappa columns vanish, their q(u) stays at the prior (KL contribution 0, no contribution to any
    prediction, statistic or gradient; their own Z-gradient is 0 as well).  The real points come first; the counts are kept."""
    m_real = [len(z) for z in Zs]
    m_max = max(m_real)
    out = []
    for z in Zs:
        z = np.asarray(z, dtype=np.float64)
        if len(z) < m_max:
            pad = np.array([[_FAR * (j + 1)] * z.shape[1] for j in range(m_max - len(z))], dtype=np.float64)
            z = np.vstack([z, pad])
        out.append(z)
    return out, m_real


class OnlineSVGP:
    """OnlineSVGP(kernel, likelihood, AnalyticVI(), Zalg=OIPS(0.9); optimiser=false, T=Float64)  OnlineSVGP.jl:33-72."""

    def __init__(self, kernel, likelihood, inference, Zalg: Optional[OIPS] = None, *, verbose: int = 0, optimiser=False,
                 atfrequency: int = 1, mean=None, Zoptimiser=False, T=np.float64, device: Optional[int] = None,
                 seed: Optional[int] = None, elbo_mode: str = "corrected"):
        if not isinstance(inference, AnalyticVI):
            raise TypeError("The inference object should be of type `AnalyticVI`")  # OnlineSVGP.jl:45
        if inference.stoch:
            raise NotImplementedError("OnlineSVGP streams full batches; the reference's stochastic branch "
                                      "(onlinetraining.jl:48-53) references an undefined variable")
        from .svgp import ADAM

        if optimiser is None:
            optimiser = ADAM(0.01)                             # OnlineSVGP.jl:38
        if isinstance(optimiser, bool):
            optimiser = ADAM(0.01) if optimiser else None      # OnlineSVGP.jl:49-51
        if isinstance(Zoptimiser, bool):
            Zoptimiser = ADAM(0.001) if Zoptimiser else None
        self.k_opt, self.z_opt = optimiser, Zoptimiser
        if mean is not None:
            raise NotImplementedError("only ZeroMean is wired for the online model")
        self.kernel, self.likelihood, self.inference = kernel, likelihood, inference
        self.Zalg = Zalg or OIPS(0.9)
        self.verbose, self.atfrequency, self.T, self.device = verbose, atfrequency, np.dtype(T), device
        self.elbo_mode = elbo_mode
        self.rng = np.random.default_rng(seed)
        self.trained = False
        self._cur: Optional[SVGP] = None   # SVGP wrapper owning the current device handle
        self._data = None
        self._max_batch = 0
        self._m_real: list = []            # inducing points per latent without the neutral padding (_pad_inducing)

    # ---- views the reference exposes -------------------------------------------------------------------------------
    @property
    def Zs(self):
        if self._cur is None:
            return []
        return [np.asarray(z)[:n] for z, n in zip(self._cur.Zs, self._m_real)]

    @property
    def n_latent(self):
        return self.likelihood.n_latent

    def get_state(self, latent: int = 0):
        mu, Sig, e1, e2 = self._cur.get_state(latent)
        n = self._m_real[latent]  # without the neutral padding
        return mu[:n], Sig[:n, :n], e1[:n], e2[:n, :n]

    def _new_svgp(self, Zs, max_batch):
        m = SVGP(self.kernel if self._cur is None else self._cur.kernels, self.likelihood, AnalyticVI(), list(Zs),
                 optimiser=self.k_opt if self.k_opt else False, Zoptimiser=self.z_opt if self.z_opt else False, T=self.T,
                 device=self.device, elbo_mode=self.elbo_mode)
        m._ensure_handle(max_batch)
        return m

    def __repr__(self):
        return f"Online Variational Gaussian Process with a {self.likelihood} infered by {self.inference} "


def train_online(model: OnlineSVGP, X, y, state=None, *, iterations: int = 20, callback: Optional[Callable] = None,
                 obsdim: int = 1):
    """train!(m::OnlineSVGP, X, y, state; iterations, callback)  onlinetraining.jl:17-135: one call per arriving batch."""
    if iterations <= 0:
        raise ValueError("Number of iterations should be positive")
    torch = _torch()
    L = capi.lib()
    first = model._cur is None
    Xh = np.asarray(X.cpu().numpy() if isinstance(X, torch.Tensor) else X, dtype=np.float64)
    if Xh.ndim == 1:
        Xh = Xh[:, None]
    if obsdim == 2:
        Xh = Xh.T
    B = len(Xh)
    mb = max(B, model._max_batch)
    if first:  # init_online_model onlinetraining.jl:182-197
        probe = SVGP(model.kernel, model.likelihood, AnalyticVI(), Xh[:1], optimiser=False, T=model.T, device=model.device)
        Zs = [_oips_update(model.Zalg, probe, k, Xh[:1].copy(), Xh[1:]) for k in probe.kernels]
        Zs, m_real = _pad_inducing(Zs)
        new = model._new_svgp(Zs, mb)
        yt = new._treat(y)
        Xd, yd = new._upload(Xh), new._upload_y(yt)
        dev = new._dev()
        m = new.m
        zero = torch.zeros(m, dtype=new.tdtype, device=dev)
        for l in range(new.n_latent):  # init_opt_state(::OnlineVarLatent) states.jl:85-97 ; Z_a empty
            eye = torch.eye(m, dtype=new.tdtype, device=dev)
            eye[m_real[l]:, m_real[l]:] = 0  # padding points carry no prior term either (they have to stay neutral)
            new._chk(L.agp_svgp_set_online_prior(new._h, l, None, 0, m, C.c_void_p(eye.data_ptr()), m,
                                                 C.c_void_p(zero.data_ptr()), 0.0))
        start = 0
    else:
        old = model._cur
        old._pull_hypers()  # kernels / inducing points as the hyper steps of the previous batch left them
        if mb > old._max_batch:
            old._ensure_handle(mb)
        dev = old._dev()
        snaps = []
        for l in range(old.n_latent):  # save_old_gp! onlinetraining.jl:170-180
            iD = torch.empty(old.m, old.m, dtype=old.tdtype, device=dev)
            e1 = torch.empty(old.m, dtype=old.tdtype, device=dev)
            pl = C.c_double()
            old._chk(L.agp_svgp_online_snapshot(old._h, l, C.c_void_p(iD.data_ptr()), old.m, C.c_void_p(e1.data_ptr()),
                                                C.byref(pl)))
            pla = pl.value
            n = model._m_real[l]
            if n < old.m:
                # The padding points must hand NOTHING on.  Their q(u) sits at the prior of the last CAVI step, the hyper step
                # after it moved K: D_a^-1 = -2 eta2 - K^-1 would carry 1/K_old - 1/K_new on their diagonal and the constant
                # (-logdet Sigma + logdet K)/2 the matching log ratio -- a spurious kernel-variance gradient in extraKL.
                sig_pad = np.diag(np.asarray(old.get_state(l)[1]))[n:]
                jit = 1e-4 if old.T == np.dtype(np.float64) else 1e-3
                k_pad = float(old.kernels[l].variance) + jit
                pla -= 0.5 * float(np.sum(-np.log(sig_pad) + np.log(k_pad)))
                iD[n:, :] = 0
                iD[:, n:] = 0
                e1[n:] = 0
            snaps.append((iD, e1, pla))
        Zs = []
        for l, k in enumerate(old.kernels):  # remove_point (needs K) then updateZ  onlinetraining.jl:153-160,172
            Z = np.asarray(old.Zs[l])[:model._m_real[l]]  # without the padding
            if model.Zalg.rho_remove < 1.0:
                Z = _oips_remove(model.Zalg, model.rng, Z, _kmat(old, k, Z, Z))
            Zs.append(_oips_update(model.Zalg, old, k, Z, Xh))
        Zs, m_real = _pad_inducing(Zs)
        new = model._new_svgp(Zs, mb)
        yt = new._treat(y)
        Xd, yd = new._upload(Xh), new._upload_y(yt)
        for l in range(new.n_latent):
            iD, e1, pl = snaps[l]
            za = torch.as_tensor(old.Zs[l], dtype=new.tdtype, device=dev).contiguous()
            new._chk(L.agp_svgp_set_online_prior(new._h, l, C.c_void_p(za.data_ptr()), za.stride(0), za.shape[0],
                                                 C.c_void_p(iD.data_ptr()), old.m, C.c_void_p(e1.data_ptr()), pl))
        if model.k_opt:  # the kernel-parameter ADAM state lives on (hyperopt_state of the reference's `state`)
            nk = 1 + new.D
            for l in range(new.n_latent):
                km, kv, ks = (C.c_double * nk)(), (C.c_double * nk)(), C.c_int32()
                old._chk(L.agp_svgp_hyper_opt_state(old._h, l, 0, km, kv, C.byref(ks)))
                new._chk(L.agp_svgp_hyper_opt_state(new._h, l, 1, km, kv, C.byref(ks)))
        new._chk(L.agp_ctx_sync(new._ctx))
        # first iteration: local update under the old inducing points, natural gradient under the new ones
        new._chk(L.agp_svgp_online_first_step(new._h, old._h, C.c_void_p(Xd.data_ptr()), Xd.stride(0),
                                              C.c_void_p(yd.data_ptr()), B))
        new._chk(L.agp_ctx_sync(new._ctx))
        start = 1
    model._cur, model._max_batch, model._m_real = new, mb, m_real
    model._data = new._data = (Xd, yd, B)
    new.inference.batchsize, new.inference.rho = B, 1.0
    model.trained = True
    for it in range(iterations):
        if it >= start:
            new._chk(L.agp_svgp_cavi_step(new._h, C.c_void_p(Xd.data_ptr()), Xd.stride(0), C.c_void_p(yd.data_ptr()), None,
                                          B, 1.0))
        if callback is not None:
            callback(model, new, model.inference.n_iter)
        # onlinetraining.jl:112-114 (n_iter is the counter before this iteration's increment)
        if (model.k_opt or model.z_opt) and model.inference.n_iter % model.atfrequency == 0 and model.inference.n_iter >= 3:
            new._chk(L.agp_svgp_hyper_step(new._h))
        model.inference.n_iter += 1
    new._chk(L.agp_svgp_check_status(new._h))
    new._pull_hypers()
    new._pull_lik_state()
    new.trained = True
    return model, new


def online_objective(model: OnlineSVGP) -> float:
    """objective(m::OnlineSVGP, state, y) = ELBO(m, state, y) incl. extraKL on the last batch (OnlineSVGP.jl:79)."""
    cur = model._cur
    Xd, yd, B = model._data
    out = C.c_double()
    cur._chk(capi.lib().agp_svgp_elbo(cur._h, C.c_void_p(Xd.data_ptr()), Xd.stride(0), C.c_void_p(yd.data_ptr()), None, B,
                                      1.0, 0, C.byref(out)))
    return out.value


def online_predict_f(model: OnlineSVGP, X_test, **kw):
    return _predict_f_fn(model._cur, X_test, **kw)


def online_predict_y(model: OnlineSVGP, X_test, **kw):
    return _predict_y_fn(model._cur, X_test, **kw)


def online_proba_y(model: OnlineSVGP, X_test, **kw):
    return _proba_y_fn(model._cur, X_test, **kw)
