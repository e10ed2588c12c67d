"""Inducing-point selection: `inducingpoints(KmeansAlg(m), X)` as the reference's examples and tests call it
(test/testingtools.jl:66, docs/examples/gpclassification.jl:47, docs/src/userguide.md:140-143; the algorithm itself is the
re-exported, unvendored InducingPoints.jl: AFK-MC2 seeding + Clustering.kmeans!).

Host logic only: the random seeding draws (they need the caller's RNG) and the short sequential Metropolis chains run
here on a few thousand gathered candidate points; every O(N m D) distance pass and the Lloyd iterations run on the GPU
through `agp_nearest_center` / `agp_kmeans` (csrc/agp_kmeans.h).
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np

from . import capi


class KmeansAlg:
    """KmeansAlg(m; nMarkov=10, tol=1e-3)  (InducingPoints.jl; metric = SqEuclidean only)."""

    def __init__(self, m: int, nMarkov: int = 10, tol: float = 1e-3, maxiter: int = 100):
        if m <= 0:
            raise ValueError("the number of inducing points must be positive")
        self.m, self.nMarkov, self.tol, self.maxiter = int(m), int(nMarkov), float(tol), int(maxiter)


class RandomSubset:
    """RandomSubset(m): m points of X drawn without replacement (InducingPoints.jl)."""

    def __init__(self, m: int):
        self.m = int(m)


def _seeding_chains(x_first, cand, u):
    Cs = [np.asarray(x_first, dtype=np.float64)]
    for i in range(cand.shape[0]):
        Cm = np.stack(Cs)
        x = cand[i, 0]
        mind = np.min(np.sum((Cm - x) ** 2, axis=1))
        for j in range(1, cand.shape[1]):
            y = cand[i, j]
            dist = np.min(np.sum((Cm - y) ** 2, axis=1))
            if dist > u[i, j - 1] * mind:
                x, mind = y, dist
        Cs.append(np.asarray(x, dtype=np.float64))
    return np.stack(Cs)


def inducingpoints(alg, X, *, rng: Optional[np.random.Generator] = None, obsdim: int = 1, device: int = 0,
                   T=np.float64, return_info: bool = False):
    """inducingpoints(alg, X; obsdim) -> (m, D) array of inducing points.  X: array or CUDA tensor, N x D (obsdim=1)."""
    import torch

    rng = rng or np.random.default_rng()
    dev = torch.device("cuda", device)
    td = torch.float64 if np.dtype(T) == np.float64 else torch.float32
    Xd = X if isinstance(X, torch.Tensor) else torch.as_tensor(np.asarray(X))
    Xd = Xd.to(device=dev, dtype=td)
    if Xd.ndim == 1:
        Xd = Xd[:, None]
    if obsdim == 2:
        Xd = Xd.t()
    Xd = Xd.contiguous()
    N, D = Xd.shape
    if alg.m > N:
        raise ValueError("Input data not big enough given the desired number of inducing points")
    if isinstance(alg, RandomSubset):
        idx = np.sort(rng.choice(N, alg.m, replace=False))
        return Xd[torch.as_tensor(idx, device=dev)].cpu().numpy().astype(np.float64)
    if not isinstance(alg, KmeansAlg):
        raise TypeError(f"{alg} is not an implemented inducing-point selection algorithm")
    L = capi.lib()
    ctx = C.c_void_p()
    st = L.agp_ctx_create(device, C.c_void_p(torch.cuda.current_stream(dev).cuda_stream), C.byref(ctx))
    if st != 0:
        raise capi.AGPError(st, "agp_ctx_create")
    dt = capi.F64 if td == torch.float64 else capi.F32

    def chk(s):
        if s != 0:
            raise capi.AGPError(s, L.agp_last_error(ctx).decode())

    try:
        # ---- AFK-MC2 seeding: q from one device distance pass, chains on the gathered candidates ----
        first = int(rng.integers(N))
        c1 = Xd[first:first + 1].contiguous()
        d1 = torch.empty(N, dtype=td, device=dev)
        chk(L.agp_nearest_center(ctx, dt, C.c_void_p(Xd.data_ptr()), N, Xd.stride(0), D, C.c_void_p(c1.data_ptr()), D, 1, None,
                                 C.c_void_p(d1.data_ptr())))
        q = d1.cpu().numpy().astype(np.float64)
        q = q / np.sum(q) / 2.0 + 1.0 / (2.0 * N)
        q = q / np.sum(q)
        prop = rng.choice(N, size=(max(alg.m - 1, 0), alg.nMarkov), p=q)
        u = rng.random((max(alg.m - 1, 0), max(alg.nMarkov - 1, 0)))
        cand = Xd[torch.as_tensor(prop.ravel(), device=dev)].cpu().numpy().astype(np.float64).reshape(prop.shape + (D,))
        seeds = _seeding_chains(Xd[first].cpu().numpy().astype(np.float64), cand, u)
        # ---- Lloyd iterations on the device ----
        Cd = torch.as_tensor(seeds, dtype=td, device=dev).contiguous()
        iters, conv, obj = C.c_int32(), C.c_int32(), C.c_double()
        chk(L.agp_kmeans(ctx, dt, C.c_void_p(Xd.data_ptr()), N, Xd.stride(0), D, C.c_void_p(Cd.data_ptr()), D, alg.m,
                         alg.maxiter, alg.tol, None, None, C.byref(iters), C.byref(obj), C.byref(conv)))
        Z = Cd.cpu().numpy().astype(np.float64)
    finally:
        L.agp_ctx_destroy(ctx)
    if return_info:
        return Z, dict(seeds=seeds, iterations=iters.value, cost=obj.value, converged=bool(conv.value))
    return Z
