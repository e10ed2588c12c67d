"""Latent-sharded multi-output model through the C ABI: two ranks simulated as two threads sharing the GPU, each with its
own handle and the real drivers of parallel.py; the collective is an in-process sum.  Checked against the single-handle
model and the oracle (src/models/single_and_multi_output_utils.jl:24-118, src/training/training.jl:153-158)."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


class ThreadGroup:
    """all-reduce(sum) between threads of one process (every thread enqueues on the same HIP stream, so the barriers that
    order the enqueues also order the device work)."""

    def __init__(self, n):
        self.n = n
        self.bar = threading.Barrier(n)
        self.slots = [None] * n
        self.tl = threading.local()

    def all_reduce_sum(self, t):
        self.slots[self.tl.rank] = t
        self.bar.wait()
        s = self.slots[0].clone()
        for o in self.slots[1:]:
            s += o
        self.bar.wait()
        t.copy_(s)
        self.bar.wait()


def _rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


def _build(AGP, R, slice_=None, aopt=True, hyper=False):
    from test_parallel_gloo import _mo_data

    X, ys, liks, Zs, A, idx, N, B, iters, Q = _mo_data()
    la = [AGP.GaussianLikelihood(0.05), AGP.LogisticLikelihood(), AGP.StudentTLikelihood(3.0)]
    k = 1.0 * (AGP.SqExponentialKernel() @ AGP.ScaleTransform(3.0))
    m = AGP.MOSVGP(k, la, AGP.AnalyticSVI(B), Zs, A=A, Aoptimiser=AGP.ADAM(0.01) if aopt else False,
                   optimiser=AGP.ADAM(0.01) if hyper else False, Zoptimiser=AGP.ADAM(0.001) if hyper else False,
                   latent_slice=slice_)
    ysl = [ys[0], (ys[1] > 0).astype(int), ys[2]]
    return m, X, ysl, idx, N, B, iters, Q


def _run_rank(P, group, rank, eng, idx, N, B, iters, hyper, out, Xt):
    try:
        group.tl.rank = rank
        elbos = []
        for it in range(iters):
            P.latent_parallel_step(eng, idx[it], N / B, group)
            elbos.append(P.elbo_parallel(eng, "latent", group))
            if hyper and it >= 1:
                P.hyper_step_parallel(eng, group)
        eng.check()
        out[rank] = {"elbo": elbos, "f": P.predict_mo_sharded(eng, Xt, "f", group),
                     "y": P.predict_mo_sharded(eng, Xt, "y", group), "p": P.predict_mo_sharded(eng, Xt, "proba", group)}
    except BaseException as e:  # surface the failure in the main thread; never leave the peer stuck in a barrier
        out[rank] = e
        group.bar.abort()


@pytest.mark.parametrize("hyper", [False, True])
def test_sharded_multioutput_two_ranks_match_single_handle(hyper):
    import agp_amd as AGP
    from agp_amd import parallel as P
    from oracle import agp_ref as R

    # single handle, same drivers (world 1)
    m1, X, ys, idx, N, B, iters, Q = _build(AGP, R, None, hyper=hyper)
    e1 = P.HipEngine(m1, B).bind_data(X, ys)
    Xt = np.random.default_rng(3).random((37, X.shape[1]))
    el1 = []
    for it in range(iters):
        P.latent_parallel_step(e1, idx[it], N / B)
        el1.append(P.elbo_parallel(e1, "latent"))
        if hyper and it >= 1:
            P.hyper_step_parallel(e1)
    e1.check()
    f1, y1, p1 = AGP.predict_f(m1, Xt, cov=True), AGP.predict_y(m1, Xt), AGP.proba_y(m1, Xt)

    # two ranks = two threads, latents [0, 2) and [2, 4)
    group = ThreadGroup(2)
    models = [_build(AGP, R, P.latent_slice(Q, 2, r), hyper=hyper)[0] for r in range(2)]
    engs = [P.HipEngine(models[r], B).bind_data(X, ys) for r in range(2)]
    out = [None, None]
    th = [threading.Thread(target=_run_rank, args=(P, group, r, engs[r], idx, N, B, iters, hyper, out, Xt)) for r in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=120)
    for o in out:
        if isinstance(o, BaseException):
            raise o
        assert o is not None
    for r in range(2):
        lo, hi = P.latent_slice(Q, 2, r)
        for k in range(hi - lo):
            a, b = models[r].get_state(k), m1.get_state(lo + k)
            assert _rel(a[2], b[2]) < 1e-11 and _rel(a[3], b[3]) < 1e-11
        assert _rel(models[r].get_A(), m1.get_A()) < 1e-12
        assert np.allclose(out[r]["elbo"], el1, rtol=1e-10)
        assert _rel(out[r]["f"][0], np.stack(f1[0])) < 1e-10 and _rel(out[r]["f"][1], np.stack(f1[1])) < 1e-9
        assert _rel(out[r]["y"], np.stack([np.asarray(v, dtype=np.float64) for v in y1])) < 1e-10
        for t in range(3):
            assert _rel(out[r]["p"][0][t], p1[t][0]) < 1e-10 and _rel(out[r]["p"][1][t], p1[t][1]) < 1e-9
    if hyper:
        for r in range(2):
            models[r]._pull_hypers()
        m1._pull_hypers()
        assert _rel(models[1].Zs[0], m1.Zs[2]) < 1e-10 and _rel(models[1].Zs[0], _build(AGP, R)[0].Zs[2]) > 1e-6
        assert models[0].kernels[1].variance == pytest.approx(m1.kernels[1].variance, rel=1e-10)

    # and the single-handle run is the oracle's
    if not hyper:
        from test_parallel_gloo import _mo_data

        Xo, yso, liks, Zs, A, idx, N, B, iters, Q = _mo_data()
        ref = R.MOSVGP(R.Kernel("sqexponential", 3.0, 1.0), liks, Zs, A.copy(), stochastic=True, batchsize=B, A_opt=R.Adam(0.01))
        elr = []
        ref.train(Xo, yso, iters, idx_stream=idx, callback=lambda M, it, xb, yb: elr.append(M.elbo(yb)))
        assert np.allclose(el1, elr, rtol=1e-8)
        assert _rel(m1.get_state(3)[3], ref.latents[3].eta2) < 1e-9


def test_sharded_multioutput_misuse_fails_loudly():
    import ctypes as C

    import agp_amd as AGP
    from agp_amd import parallel as P
    from oracle import agp_ref as R

    m, X, ys, idx, N, B, iters, Q = _build(AGP, R, (0, 2))
    eng = P.HipEngine(m, B).bind_data(X, ys)
    with pytest.raises(Exception, match="mo_mix"):
        eng.mo_mix()  # nothing published yet
    eng.step_local(idx[0], N / B)
    with pytest.raises(Exception, match="mo_refresh_f"):
        eng.elbo_local()  # exchange buffer holds pre-update values
    with pytest.raises(Exception, match="partial mix"):
        AGP.predict_y(m, X[:5])


def _toy(likname):
    from test_parallel_gloo import _data

    return _data(likname)


def _run_generic(P, group, rank, eng, idx, N, B, iters, mode, world, out):
    try:
        group.tl.rank = rank
        elbos = []
        for it in range(iters):
            if mode == "latent":
                P.latent_parallel_step(eng, idx[it], N / B, group)
            else:
                P.batch_parallel_step(eng, P.shard_batch(idx[it], world, rank), N / B, group)
            elbos.append(P.elbo_parallel(eng, mode, group))
        eng.check()
        out[rank] = elbos
    except BaseException as e:
        out[rank] = e
        group.bar.abort()


@pytest.mark.parametrize("mode,likname", [("latent", "logisticsoftmax"), ("batch", "logistic"), ("batch", "logisticsoftmax")])
def test_two_rank_drivers_on_device_match_single_handle(mode, likname):
    """C4's plan (LogisticSoftMax latents over ranks, the sum_k gamma_k all-reduce) and the batch-parallel plan (minibatch over
    ranks, the statistics all-reduce), each with two handles driven by two threads, against one handle; ELBO included."""
    import agp_amd as AGP
    from agp_amd import parallel as P

    X, y, lik, Z, idx, N, B, iters = _toy(likname)
    K = 4
    mk = lambda sl=None: AGP.SVGP(1.0 * (AGP.SqExponentialKernel() @ AGP.ScaleTransform(3.0)),  # noqa: E731
                                  AGP.LogisticSoftMaxLikelihood(K) if likname == "logisticsoftmax" else AGP.LogisticLikelihood(),
                                  AGP.AnalyticSVI(B), Z, optimiser=False, latent_slice=sl)
    m1 = mk()
    e1 = P.HipEngine(m1, B).bind_data(X, y)
    el1 = []
    for it in range(iters):
        P.latent_parallel_step(e1, idx[it], N / B)
        el1.append(P.elbo_parallel(e1, "latent"))
    e1.check()
    group = ThreadGroup(2)
    if mode == "latent":
        models = [mk(P.latent_slice(K, 2, r)) for r in range(2)]
        engs = [P.HipEngine(models[r], B).bind_data(X, y) for r in range(2)]
    else:
        models = [mk() for r in range(2)]
        engs = [P.HipEngine(models[r], B // 2).bind_data(X, y).set_batch_shard(r, 2) for r in range(2)]
    out = [None, None]
    th = [threading.Thread(target=_run_generic, args=(P, group, r, engs[r], idx, N, B, iters, mode, 2, out)) for r in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=120)
    for o in out:
        if isinstance(o, BaseException):
            raise o
        assert o is not None
    for r in range(2):
        lo, hi = P.latent_slice(K, 2, r) if mode == "latent" else (0, m1.n_latent)
        for k in range(hi - lo):
            a, b = models[r].get_state(k), m1.get_state(lo + k)
            assert _rel(a[2], b[2]) < 1e-9 and _rel(a[3], b[3]) < 1e-9
        assert np.allclose(out[r], el1, rtol=1e-9), (out[r], el1)
