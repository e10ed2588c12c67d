"""Multi-GPU drivers: one process per GPU, torch.distributed (backend "nccl" == RCCL over xGMI; "gloo" in CPU tests).

The path shards in two ways:

* latent-parallel -- the latent GPs of a multi-class / multi-output model are independent between exchanges (each owns
  kernel, Z, K, kappa, eta: deep copies in the reference, src/gpblocks/latentgp.jl:63-68).  Rank r holds the latent slice
  `latent_slice(K, world, r)`; X, y and the minibatch index stream are replicated.  The only data-path collective is the
  LogisticSoftMax fixed point: sum_k gamma_k over ALL latents, a length-B vector all-reduced twice per step
  (src/likelihood/logisticsoftmax.jl:65-72).  Other likelihoods need no collective at all.
  A latent-sharded multi-output model (MOSVGP(..., latent_slice=...)) mixes ALL latents into every task
  (src/models/single_and_multi_output_utils.jl:24-84): the (mean_f, var_f) of the owned latents are exchanged once per step
  (2 Q B numbers, the all-gather of section 8e written as an all-reduce over a zero-padded buffer), after which the task-side
  work and update_A! run redundantly on every rank.
* batch-parallel -- one latent, the minibatch split across ranks; every per-point quantity is row-independent and only the
  batch statistics [kappa'(rho g1) | rho kappa' diag(g2) kappa] couple the shards: one all-reduce per step
  (src/inference/analyticVI.jl:168,179), after which every rank applies the identical global step.

The product path is behind the C ABI: `Comm` wraps an `agp_comm` (RCCL bound directly by libagp_hip.so, or a host-supplied
all-reduce) and `HipEngine.step_multi / elbo_multi / hyper_step_multi / predict_multi` are one `agp_svgp_*_multi` call each --
what a Julia / C host calls too (include/agp_hip.h).  torch.distributed is only the bootstrap channel that
carries the 128-byte RCCL id from rank 0 to the others (`Comm.rccl_from_torch`), or -- for gloo groups in CPU tests and
in-process thread groups -- the transport behind a callback communicator (`Comm.from_group`).

The phase-level drivers further down (latent_parallel_step, batch_parallel_step, ...) spell the same plans out over an
*engine* exposing the phase-split step (step_local / lsm_gamma / lsm_alpha / step_stats / step_global) and the exchange
tensors; they are what the world_size-2 gloo tests run against a CPU reference engine, and they accept a HipEngine too.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import numpy as np

from . import capi


_EXT_STREAMS = {}


def on_stream(stream):
    """torch context for the raw hipStream_t an all-reduce callback is handed (None / 0: the default stream torch is on anyway)"""
    import contextlib

    if not stream:
        return contextlib.nullcontext()
    import torch

    key = int(stream)
    if key not in _EXT_STREAMS:
        _EXT_STREAMS[key] = torch.cuda.ExternalStream(key)
    return torch.cuda.stream(_EXT_STREAMS[key])


class Comm:
    """An `agp_comm` (include/agp_hip.h): sum all-reduce across the ranks of a run, enqueued on the model's HIP stream."""

    def __init__(self, model, handle, rank, world, keep=None):
        self.model, self.h, self.rank, self.world = model, handle, rank, world
        self._keep = keep  # ctypes callback object must outlive the communicator

    @staticmethod
    def unique_id() -> bytes:
        """ncclGetUniqueId through the library (rank 0 calls this; ship the bytes to the other ranks)."""
        buf = (C.c_uint8 * capi.COMM_ID_BYTES)()
        st = capi.lib().agp_comm_unique_id(buf)
        if st != capi.AGP_OK:
            raise capi.AGPError(st, "agp_comm_unique_id failed (librccl not found?)")
        return bytes(buf)

    @classmethod
    def rccl(cls, model, rank: int, world: int, unique_id: bytes) -> "Comm":
        """ncclCommInitRank on the model's device; `unique_id` from rank 0's Comm.unique_id()."""
        ctx = model._ensure_ctx()
        buf = (C.c_uint8 * capi.COMM_ID_BYTES).from_buffer_copy(unique_id)
        h = C.c_void_p()
        model._chk(capi.lib().agp_comm_init(ctx, rank, world, buf, C.byref(h)))
        return cls(model, h, rank, world)

    @classmethod
    def rccl_from_torch(cls, model, group=None) -> "Comm":
        """RCCL communicator bootstrapped over an initialised torch.distributed group (any backend): the group only
        broadcasts the 128-byte id; every collective of the data path is then issued by libagp_hip.so itself."""
        import torch.distributed as dist

        rank, world = dist.get_rank(group), dist.get_world_size(group)
        box = [cls.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        return cls.rccl(model, rank, world, box[0])

    @classmethod
    def from_callback(cls, model, rank: int, world: int, fn) -> "Comm":
        """fn(ptr: int, count: int, dtype: int, stream: int) -> None sums `count` elements in place at the device pointer."""
        def _cb(user, buf, count, dtype, stream):
            try:
                fn(buf, count, dtype, stream)
                return 0
            except BaseException as e:  # never unwind through the C frames
                _cb.error = e
                return 1

        cb = capi.ALLREDUCE_FN(_cb)
        ctx = model._ensure_ctx()
        h = C.c_void_p()
        model._chk(capi.lib().agp_comm_init_callback(ctx, rank, world, cb, None, C.byref(h)))
        c = cls(model, h, rank, world, keep=(cb, _cb))
        return c

    @classmethod
    def from_group(cls, model, group=None, rank: Optional[int] = None, world: Optional[int] = None) -> "Comm":
        """Callback communicator over a torch.distributed group (gloo or nccl) or an in-process group exposing
        all_reduce_sum(tensor) (tests: ranks as threads)."""
        import torch

        dev = model._dev()

        def view(ptr, count, dtype):
            ts = "<f8" if dtype == capi.F64 else "<f4"
            return torch.as_tensor(_DevBuf(ptr, count, ts), device=dev)

        if group is not None and hasattr(group, "all_reduce_sum"):
            def fn(ptr, count, dtype, stream):
                with on_stream(stream):
                    group.all_reduce_sum(view(ptr, count, dtype))
            return cls.from_callback(model, rank, world, fn)
        import torch.distributed as dist

        r, w = dist.get_rank(group), dist.get_world_size(group)
        gloo = dist.get_backend(group) == "gloo"

        def fn(ptr, count, dtype, stream):
            # the library names the stream the sum has to be ordered on: the ctx's, or -- AGP_SPLIT_OVERLAP -- the communicator's own
            with on_stream(stream):
                t = view(ptr, count, dtype)
                if gloo:  # host-staged: gloo reduces host memory
                    hbuf = t.cpu()
                    dist.all_reduce(hbuf, op=dist.ReduceOp.SUM, group=group)
                    t.copy_(hbuf)
                else:
                    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        return cls.from_callback(model, r, w, fn)

    @property
    def is_rccl(self) -> bool:
        f = C.c_int32()
        capi.lib().agp_comm_info(self.h, None, None, C.byref(f))
        return bool(f.value)

    def timing(self, on=True):
        """on: False / True (every collective bracketed by HIP events) / n > 1 (every n-th: the records cost stream time)"""
        self.model._chk(capi.lib().agp_comm_timing(self.h, int(on)))

    def stats(self):
        """(collectives, bytes reduced per rank, summed milliseconds if timing was on) since the last call"""
        n, b, ms = C.c_int64(), C.c_int64(), C.c_double()
        self.model._chk(capi.lib().agp_comm_stats(self.h, C.byref(n), C.byref(b), C.byref(ms)))
        return n.value, b.value, ms.value

    def all_reduce(self, t):
        """in-place sum of a device tensor (float64 / float32) on the model's stream"""
        import torch

        dt = capi.F64 if t.dtype == torch.float64 else capi.F32
        self.model._chk(capi.lib().agp_comm_allreduce(self.h, C.c_void_p(t.data_ptr()), t.numel(), dt))
        return t

    def _raise_pending(self):
        if self._keep is not None and getattr(self._keep[1], "error", None) is not None:
            e = self._keep[1].error
            self._keep[1].error = None
            raise e

    def destroy(self):
        if self.h is not None:
            capi.lib().agp_comm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


def latent_slice(n_latent: int, world: int, rank: int):
    """Contiguous balanced slice [lo, hi) of the latents owned by `rank` (C4: 8 latents / 8 GPUs -> one each;
    C5: 16 latents / 8 GPUs -> two each)."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    base, rem = divmod(n_latent, world)
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


def shard_batch(idx: np.ndarray, world: int, rank: int) -> np.ndarray:
    """Rank r's contiguous share of a minibatch index vector (batch-parallel).  B must divide evenly so that every
    rank carries the same weight rho = N / B_total."""
    idx = np.asarray(idx)
    if len(idx) % world != 0:
        raise ValueError(f"batch size {len(idx)} is not divisible by the world size {world}")
    per = len(idx) // world
    return idx[rank * per:(rank + 1) * per]


def row_slice(n: int, world: int, rank: int):
    """Contiguous balanced slice [lo, hi) of n test points for `rank` (data-parallel prediction: independent units,
    no collective -- SURVEY.md section 8e, third row)."""
    return latent_slice(n, world, rank)


def predict_sharded(predict_fn, X_test, group=None, gather: bool = True):
    """Data-parallel predict_f / proba_y over the rows of X_test: every rank runs `predict_fn` (e.g.
    `lambda X: agp_amd.predict_f(model, X, cov=True)`) on its row slice.  gather=True all-gathers the (numpy) results so
    every rank returns the full arrays; gather=False returns (lo, hi, local result)."""
    import torch.distributed as dist

    on = dist.is_available() and dist.is_initialized()
    world = dist.get_world_size(group) if on else 1
    rank = dist.get_rank(group) if on else 0
    n = len(X_test)
    lo, hi = row_slice(n, world, rank)
    local = predict_fn(X_test[lo:hi])
    if not gather or world == 1:
        return local if world == 1 else (lo, hi, local)
    parts = [None] * world
    dist.all_gather_object(parts, local, group=group)
    if isinstance(local, tuple):
        return tuple(np.concatenate([p[i] for p in parts], axis=-1 if np.ndim(parts[0][i]) > 1 else 0)
                     for i in range(len(local)))
    return np.concatenate(parts, axis=-1 if np.ndim(local) > 1 else 0)


def _all_reduce(t, group=None):
    import torch.distributed as dist

    if group is not None and hasattr(group, "all_reduce_sum"):
        group.all_reduce_sum(t)  # in-process group (tests: several ranks as threads sharing one GPU)
        return t

    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


def latent_parallel_step(engine, idx, rho: float, group=None) -> None:
    """One CAVI step with the latents sharded across ranks.  `idx` is the SAME minibatch on every rank."""
    engine.step_local(idx, rho)
    if engine.is_lsm:
        for _ in range(2):  # the reference iterates the (gamma, alpha) fixed point twice
            engine.lsm_gamma()
            _all_reduce(engine.gsum, group)
            engine.lsm_alpha()
    if getattr(engine, "is_mo_sharded", False):
        _all_reduce(engine.fbuf, group)  # every latent's (mean_f, var_f) on the minibatch
        engine.mo_mix()                  # update_A! + mixed local updates + the owned latents' gradients
    engine.step_stats()
    engine.step_global()


def elbo_parallel(engine, mode: str = "latent", group=None) -> float:
    """ELBO(model, state, y) on the last minibatch of a sharded run (analyticVI.jl:255-297), identical on every rank.
    latent mode: every rank evaluates its latents' share (shared terms are counted by the owner of latent 0), one scalar
    all-reduce; a latent-sharded multi-output model first re-exchanges (mean_f, var_f) under the updated posterior.
    batch mode: the data / augmented-KL sums are per-shard, the Gaussian KL is replicated and counted once."""
    import torch

    if getattr(engine, "is_mo_sharded", False):
        engine.mo_refresh_f()
        _all_reduce(engine.fbuf, group)
    total, (e_data, kl_gauss, kl_aug) = engine.elbo_local()
    if mode == "latent":
        t = torch.tensor([total], dtype=torch.float64, device=engine.reduce_device)
        return float(_all_reduce(t, group)[0])
    t = torch.tensor([e_data, kl_aug], dtype=torch.float64, device=engine.reduce_device)
    _all_reduce(t, group)
    return float(engine.rho * t[0] - kl_gauss - engine.rho * t[1])


def hyper_step_parallel(engine, group=None) -> None:
    """update_hyperparameters! (autotuning.jl:86-140) of a latent-sharded model: every latent optimises its own kernel and Z
    (the reference's deep copies), so the step itself needs no collective; only a sharded multi-output model first
    re-exchanges the latents' mean_f under the updated posterior, which the mixed data term of the gradient reads."""
    if getattr(engine, "is_mo_sharded", False):
        engine.mo_refresh_f()
        _all_reduce(engine.fbuf, group)
    engine.hyper_step()


def predict_mo_sharded(engine, X_test, what: str = "f", group=None):
    """predict_f / predict_y / proba_y of a latent-sharded multi-output model: the partial mixes sum_{q owned} A[t][q]^p f_q
    are all-reduced (n_task x n_t numbers each), the task likelihoods then finish in place on every rank."""
    mu, var = engine.predict_f_partial(X_test, need_var=(what != "y"))
    _all_reduce(mu, group)
    if var is not None:
        _all_reduce(var, group)
    return engine.finish_predict(mu, var, what)


def batch_parallel_step(engine, idx_local, rho: float, group=None) -> None:
    """One CAVI step with the minibatch sharded across ranks (all latents replicated).  `idx_local` is this rank's
    share; `rho` = N / B_total."""
    engine.step_local(idx_local, rho)
    if engine.is_lsm:
        for _ in range(2):
            engine.lsm_gamma()
            engine.lsm_alpha()
    engine.step_stats()
    _all_reduce(engine.stats, group)
    engine.step_global()


def tied_hyper_step(engine, group=None) -> None:
    """Hyper-parameter / inducing-point step of a model whose latents share ONE kernel and ONE set of inducing points
    ("tied Z": an opt-in extension -- the reference gives every latent its own deep copy, latentgp.jl:63-68).  The gradient of
    the shared parameters is the sum over all latents, so every rank sums the gradients of its latent slice, the sums are
    all-reduced (m x D + 1 + D numbers: BASELINE.json config 4's "all-reduce on the Z hyper-grad"), and every latent on every
    rank applies the same ADAM step -- kernels and Z stay identical everywhere without ever being broadcast."""
    g = engine.hyper_gradients()   # flat tensor [dvariance | dscale(D) | dZ(m*D)] summed over the local latents
    _all_reduce(g, group)
    engine.hyper_apply(g)


class _DevBuf:
    """Zero-copy view of a library-owned device buffer for torch (via __cuda_array_interface__)."""

    def __init__(self, ptr: int, n: int, typestr: str):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (ptr, False), "version": 2}


class HipEngine:
    """Phase-split engine over an `SVGP` model's device handle (data must already be resident: `bind_data`)."""

    def __init__(self, model, max_batch: int):
        self.model = model
        self.L = capi.lib()
        self.h = model._ensure_handle(max_batch)
        model._chk(self.L.agp_svgp_refresh_K(self.h))
        self.is_lsm = model.likelihood.kind == capi.LIK_LOGISTICSOFTMAX
        self.is_mo_sharded = bool(getattr(model, "sharded", False))
        self._X = self._y = None
        self._stats = None
        self._gsum = None
        self._fbuf = None
        self._B = 0
        self.rho = 1.0

    @property
    def reduce_device(self):
        return self.model._dev()

    def set_batch_shard(self, rank: int, world: int):
        """this engine sees shard `rank` of `world` of every minibatch (batch-parallel): once-per-evaluation ELBO terms are
        then counted by rank 0 only"""
        self.model._chk(self.L.agp_svgp_set_batch_shard(self.h, rank, world))
        self.model._batch_shard = (int(rank), int(world))  # re-applied whenever the model re-creates its handle
        return self

    def bind_data(self, X, y, obsdim: int = 1):
        self._X = self.model._upload(X, obsdim)
        self._y = self.model._upload_y(self.model._treat(y))
        return self

    def _view(self, ptr_fn):
        import torch

        p, n = C.c_void_p(), C.c_int64()
        self.model._chk(ptr_fn(self.h, C.byref(p), C.byref(n)))
        ts = "<f8" if self.model.T == np.dtype(np.float64) else "<f4"
        return torch.as_tensor(_DevBuf(p.value, n.value, ts), device=self.model._dev())

    def step_local(self, idx, rho: float):
        import torch

        idx_t = torch.as_tensor(np.asarray(idx, dtype=np.int64), device=self.model._dev())
        self._keep = idx_t
        self._B = idx_t.numel()
        self.rho = float(rho)
        self.model._chk(self.L.agp_svgp_step_local(self.h, C.c_void_p(self._X.data_ptr()), self._X.stride(0),
                                                   C.c_void_p(self._y.data_ptr()), C.c_void_p(idx_t.data_ptr()),
                                                   self._B, float(rho)))
        self._gsum = None

    def lsm_gamma(self):
        self.model._chk(self.L.agp_svgp_lsm_gamma(self.h))

    def lsm_alpha(self):
        self.model._chk(self.L.agp_svgp_lsm_alpha(self.h))

    @property
    def gsum(self):
        if self._gsum is None:
            self._gsum = self._view(self.L.agp_svgp_lsm_gsum_ptr)
        return self._gsum

    @property
    def fbuf(self):
        """exchange buffer T[2][Q][Bp] of a latent-sharded multi-output handle (own rows filled, zeros elsewhere)"""
        if self._fbuf is None:
            self._fbuf = self._view(self.L.agp_svgp_mo_fbuf_ptr)
        return self._fbuf

    def mo_mix(self):
        self.model._chk(self.L.agp_svgp_mo_mix(self.h))

    def mo_refresh_f(self):
        self.model._chk(self.L.agp_svgp_mo_refresh_f(self.h))

    def elbo_local(self):
        """(this rank's ELBO share, (data term, Gaussian KL, augmented KL)) on the last minibatch"""
        out = C.c_double()
        parts = (C.c_double * 3)()
        self.model._chk(self.L.agp_svgp_elbo(self.h, C.c_void_p(self._X.data_ptr()), self._X.stride(0),
                                             C.c_void_p(self._y.data_ptr()), C.c_void_p(self._keep.data_ptr()), self._B,
                                             self.rho, 0, C.byref(out)))
        self.model._chk(self.L.agp_svgp_elbo_terms(self.h, parts))
        return out.value, tuple(parts)

    def predict_f_partial(self, X_test, need_var=True):
        import torch

        mdl = self.model
        Xt = mdl._upload(X_test, 1)
        nt = Xt.shape[0]
        mu = torch.empty(mdl.n_task, nt, dtype=mdl.tdtype, device=mdl._dev())
        var = torch.empty_like(mu) if need_var else None
        mdl._chk(self.L.agp_svgp_predict_f(self.h, C.c_void_p(Xt.data_ptr()), Xt.stride(0), nt,
                                           C.c_void_p(mu.data_ptr()), C.c_void_p(var.data_ptr()) if need_var else None))
        return mu, var

    def finish_predict(self, mu, var, what):
        from .svgp import _gauss_hermite as _gh

        mdl = self.model
        nt = mu.shape[1]
        if what == "y":
            mdl._chk(self.L.agp_svgp_mo_predict_from_f(self.h, nt, 0, C.c_void_p(mu.data_ptr()), None, None, None, 0))
        elif what == "proba":
            nodes, weights = _gh()
            mdl._chk(self.L.agp_svgp_mo_predict_from_f(
                self.h, nt, 1, C.c_void_p(mu.data_ptr()), C.c_void_p(var.data_ptr()),
                nodes.ctypes.data_as(C.POINTER(C.c_double)), weights.ctypes.data_as(C.POINTER(C.c_double)), len(nodes)))
        mdl._chk(self.L.agp_ctx_sync(mdl._ctx))
        if what == "y":
            return mu.cpu().numpy()
        return mu.cpu().numpy(), var.cpu().numpy()

    def step_stats(self):
        self.model._chk(self.L.agp_svgp_step_stats(self.h))

    @property
    def stats(self):
        if self._stats is None:
            self._stats = self._view(self.L.agp_svgp_stats_ptr)
        return self._stats

    def step_global(self):
        self.model._chk(self.L.agp_svgp_step_global(self.h))

    def hyper_step(self):
        self.model._chk(self.L.agp_svgp_hyper_step(self.h))

    def hyper_gradients(self):
        """sum over this rank's latents of (dvariance, dscale[D], dZ[m, D]) as one flat device tensor"""
        import torch

        mdl = self.model
        D, m = mdl.D, mdl.m
        tot = torch.zeros(1 + D + m * D, dtype=torch.float64, device=mdl._dev())
        dz = torch.empty(m, D, dtype=mdl.tdtype, device=mdl._dev())
        for l in range(mdl.n_latent):
            dv, ds = C.c_double(), (C.c_double * D)()
            mdl._chk(self.L.agp_svgp_hypergrad(self.h, l, C.byref(dv), ds, C.c_void_p(dz.data_ptr())))
            tot[0] += dv.value
            tot[1:1 + D] += torch.tensor(list(ds), dtype=torch.float64, device=tot.device)
            tot[1 + D:] += dz.reshape(-1).to(torch.float64)
        return tot

    def hyper_apply(self, g):
        mdl = self.model
        D, m = mdl.D, mdl.m
        host = g[:1 + D].cpu().numpy()
        dz = g[1 + D:].reshape(m, D).to(mdl.tdtype).contiguous()
        ds = (C.c_double * D)(*host[1:])
        for l in range(mdl.n_latent):
            dv = C.c_double(float(host[0]))
            mdl._chk(self.L.agp_svgp_hyper_apply(self.h, l, C.byref(dv), ds, C.c_void_p(dz.data_ptr())))

    def check(self):
        self.model._chk(self.L.agp_svgp_check_status(self.h))

    # ---- the sharded step / ELBO / hyper step / prediction as ONE C-ABI call each (collectives inside the library) ----
    def _mchk(self, st, comm):
        if st != capi.AGP_OK and comm is not None:
            comm._raise_pending()  # a Python exception raised inside the all-reduce callback
        self.model._chk(st)

    def step_multi(self, idx, rho: float, mode: int, comm: Optional["Comm"] = None):
        """agp_svgp_cavi_step_multi: mode = capi.SHARD_LATENT (idx = the whole minibatch) or capi.SHARD_BATCH (idx = this
        rank's share, rho = N / B_total)"""
        import torch

        idx_t = idx if isinstance(idx, torch.Tensor) else torch.as_tensor(np.asarray(idx, dtype=np.int64),
                                                                          device=self.model._dev())
        self._keep = idx_t
        self._B = idx_t.numel()
        self.rho = float(rho)
        st = self.L.agp_svgp_cavi_step_multi(self.h, comm.h if comm is not None else None, mode,
                                             C.c_void_p(self._X.data_ptr()), self._X.stride(0),
                                             C.c_void_p(self._y.data_ptr()), C.c_void_p(idx_t.data_ptr()), self._B,
                                             float(rho))
        self._mchk(st, comm)

    def prefetch(self, idx):
        """agp_svgp_prefetch: announce the NEXT minibatch (this rank's share) so that its K_nm / kappa are formed on the look-ahead
        stream while the current step factors.  Returns the device index tensor: hand exactly that object to the next
        step_multi / step_local (the look-ahead is recognised by pointer identity, include/agp_hip.h)."""
        import torch

        idx_t = idx if isinstance(idx, torch.Tensor) else torch.as_tensor(np.asarray(idx, dtype=np.int64),
                                                                          device=self.model._dev())
        self._keep_next = idx_t
        self.model._chk(self.L.agp_svgp_prefetch(self.h, C.c_void_p(self._X.data_ptr()), self._X.stride(0),
                                                 C.c_void_p(idx_t.data_ptr()), idx_t.numel()))
        return idx_t

    def step_counters(self):
        """(steps, steps whose natural-gradient part rode on the following task-graph launch)  -- agp_svgp_step_counters"""
        n, npro = C.c_int64(), C.c_int64()
        self.model._chk(self.L.agp_svgp_step_counters(self.h, C.byref(n), C.byref(npro)))
        return n.value, npro.value

    def elbo_multi(self, mode: int, comm: Optional["Comm"] = None) -> float:
        out = C.c_double()
        self._mchk(self.L.agp_svgp_elbo_multi(self.h, comm.h if comm is not None else None, mode, C.byref(out)), comm)
        return out.value

    def hyper_step_multi(self, comm: Optional["Comm"] = None, tied: bool = False):
        self._mchk(self.L.agp_svgp_hyper_step_multi(self.h, comm.h if comm is not None else None, 1 if tied else 0), comm)

    def predict_multi(self, X_test, what: str = "f", comm: Optional["Comm"] = None):
        """predict_f / predict_y / proba_y of a latent-sharded multi-output model (partial mixes all-reduced inside)"""
        import torch

        from .svgp import _gauss_hermite as _gh

        mdl = self.model
        Xt = mdl._upload(X_test, 1)
        nt = Xt.shape[0]
        mu = torch.empty(mdl.n_task, nt, dtype=mdl.tdtype, device=mdl._dev())
        var = torch.empty_like(mu) if what != "y" else None
        nodes, weights = _gh()
        code = {"f": 0, "y": 1, "proba": 2}[what]
        st = self.L.agp_svgp_predict_multi(self.h, comm.h if comm is not None else None, code, C.c_void_p(Xt.data_ptr()),
                                           Xt.stride(0), nt, C.c_void_p(mu.data_ptr()),
                                           C.c_void_p(var.data_ptr()) if var is not None else None,
                                           nodes.ctypes.data_as(C.POINTER(C.c_double)),
                                           weights.ctypes.data_as(C.POINTER(C.c_double)), len(nodes))
        self._mchk(st, comm)
        mdl._chk(self.L.agp_ctx_sync(mdl._ctx))
        if what == "y":
            return mu.cpu().numpy()
        return mu.cpu().numpy(), var.cpu().numpy()


def train_parallel(model, X, y, iterations: int, idx_stream: Sequence, *, mode: str = "latent", group=None,
                   obsdim: int = 1, comm: Optional[Comm] = None) -> HipEngine:
    """train! for a sharded model.  mode="latent": `model` was built with latent_slice=latent_slice(K, world, rank) and
    every rank passes the same idx_stream.  mode="batch": every rank holds the full model and idx_stream entries are the
    FULL minibatches (each rank takes its share).
    comm: a `Comm` -- every step is then one agp_svgp_cavi_step_multi call with the collectives inside the library (RCCL);
    without it the phase-level drivers below run over `group` (torch.distributed)."""
    import torch.distributed as dist

    if comm is not None:
        world, rank = comm.world, comm.rank
    else:
        world = dist.get_world_size(group) if dist.is_initialized() else 1
        rank = dist.get_rank(group) if dist.is_initialized() else 0
    if world > 1 and type(model.likelihood).__name__ in ("PoissonLikelihood", "HeteroscedasticLikelihood") and \
            not (mode == "batch" and comm is not None):
        # their lambda update (poisson.jl:78, heteroscedastic.jl:94) is a reduction over the whole minibatch: the library
        # all-reduces the two sums inside agp_svgp_cavi_step_multi(AGP_SHARD_BATCH); the phase-level drivers below do not, and
        # the two heteroscedastic latents are coupled point-wise (they stay on one handle: no latent sharding)
        raise NotImplementedError(f"{model.likelihood} shards over the minibatch through a Comm only "
                                  "(train_parallel(..., mode='batch', comm=Comm...))")
    N = np.asarray(X).shape[0] if obsdim == 1 else np.asarray(X).shape[1]
    B_total = len(idx_stream[0])
    B_local = B_total if mode == "latent" else B_total // world
    eng = HipEngine(model, B_local).bind_data(X, y, obsdim)
    if mode == "batch":
        eng.set_batch_shard(rank, world)
    model.inference.rho = N / B_total
    model.inference.batchsize = B_local
    for it in range(iterations):
        idx = np.asarray(idx_stream[it])
        if comm is not None:
            if mode == "latent":
                eng.step_multi(idx, N / B_total, capi.SHARD_LATENT, comm)
            else:
                eng.step_multi(shard_batch(idx, world, rank), N / B_total, capi.SHARD_BATCH, comm)
        elif mode == "latent":
            latent_parallel_step(eng, idx, N / B_total, group)
        else:
            batch_parallel_step(eng, shard_batch(idx, world, rank), N / B_total, group)
        model.inference.n_iter += 1
    eng.check()
    model.trained = True
    return eng
