"""Multi-GPU entry points of the C ABI (agp_comm_*, agp_svgp_*_multi; include/agp_hip.h) on ONE GPU box:

* the RCCL transport with a world of one (librccl resolved by dlopen, ncclCommInitRank / ncclAllReduce really called);
* two ranks as two threads with two handles and a callback communicator -- batch-parallel logistic (packed-triangle
  statistics), latent-parallel LogisticSoftMax (sum_k gamma_k), the tied-Z hyper step -- against the single-handle model;
* two ranks as two PROCESSES sharing the GPU, the all-reduce done through host shared memory by a ctypes callback: no
  torch.distributed anywhere, which is what a Julia / C host sees.
"""
import ctypes as C
import multiprocessing as mp
import os
import threading

import numpy as np
import pytest

import _knobs as KN

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


class ThreadGroup:
    """sum all-reduce between the threads of one process (same HIP stream: the barriers that order the enqueues also order
    the device work)"""

    def __init__(self, n):
        self.n = n
        self.bar = threading.Barrier(n)
        self.slots = [None] * n
        self.tl = threading.local()

    def all_reduce_sum(self, t):
        import torch

        # (the ranks may sit on different streams -- AGP_SPLIT_OVERLAP hands the communicator's own -- so the host waits for its
        # rank's input, and for its reads of the others', before the barriers let anybody go on)
        torch.cuda.current_stream().synchronize()
        self.slots[self.tl.rank] = t
        self.bar.wait()
        s = self.slots[0].clone()
        for o in self.slots[1:]:
            s += o
        torch.cuda.current_stream().synchronize()
        self.bar.wait()
        t.copy_(s)
        self.bar.wait()


BIG = dict(N=20000, D=8, m=1024)  # C2-sized factorisations (16 block columns, 408-workgroup task graphs)


def _data(rng, N=400, D=3, m=70, K=1):
    X = rng.random((N, D))
    f = np.sin(3 * X[:, 0]) + X[:, 1] ** 2 - 0.7
    Z = X[rng.permutation(N)[:m]].copy()
    if K == 1:
        y = (f + 0.2 * rng.standard_normal(N) > 0).astype(int)
    else:
        y = 1 + np.digitize(f, np.quantile(f, np.linspace(0, 1, K + 1)[1:-1]))
    return X, y, Z


def _kernel(AGP):
    return 1.5 * (AGP.SqExponentialKernel() @ AGP.ScaleTransform(2.0))


def test_rccl_world_of_one_through_the_abi(built):
    """dlopen(librccl) -> ncclGetUniqueId -> ncclCommInitRank(1 rank) -> ncclAllReduce on the ctx stream; a step driven through
    agp_svgp_cavi_step_multi with that communicator equals agp_svgp_cavi_step."""
    import torch

    import agp_amd as AGP
    from agp_amd import capi
    from agp_amd import parallel as P

    rng = np.random.default_rng(1)
    X, y, Z = _data(rng)
    B, iters = 128, 4
    idx = [rng.choice(len(X), B, replace=False) for _ in range(iters)]
    ref = AGP.SVGP(_kernel(AGP), AGP.LogisticLikelihood(), AGP.AnalyticSVI(B), Z, optimiser=False)
    AGP.train_(ref, X, y, iters, idx_stream=idx)
    for mode in ("latent", "batch"):
        m = AGP.SVGP(_kernel(AGP), AGP.LogisticLikelihood(), AGP.AnalyticSVI(B), Z, optimiser=False)
        m._ensure_ctx()
        uid = P.Comm.unique_id()
        assert len(uid) == capi.COMM_ID_BYTES and any(uid)
        comm = P.Comm.rccl(m, 0, 1, uid)
        assert comm.is_rccl and comm.world == 1
        t = torch.arange(1000, dtype=torch.float64, device="cuda")
        comm.timing(True)
        comm.all_reduce(t)
        torch.cuda.synchronize()
        assert torch.equal(t.cpu(), torch.arange(1000, dtype=torch.float64))  # one rank: the sum is the input
        n, nbytes, ms = comm.stats()
        assert n == 1 and nbytes == 8000 and ms >= 0.0
        eng = P.train_parallel(m, X, y, iters, idx, mode=mode, comm=comm)
        for a, b in zip(m.get_state(0), ref.get_state(0)):
            assert _rel(a, b) < 1e-12
        assert abs(eng.elbo_multi(capi.SHARD_LATENT if mode == "latent" else capi.SHARD_BATCH, comm)
                   - AGP.objective(ref)) < 1e-8 * abs(AGP.objective(ref))
        comm.destroy()


def _thread_ranks(world, body):
    """run body(rank, group) on `world` threads; re-raise the first failure"""
    group = ThreadGroup(world)
    out = [None] * world

    def run(r):
        try:
            group.tl.rank = r
            out[r] = body(r, group)
        except BaseException as e:
            out[r] = e
            group.bar.abort()

    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for o in out:
        if isinstance(o, BaseException):
            raise o
    return out


@pytest.mark.parametrize("likname", ["logistic", "logisticsoftmax"])
def test_batch_parallel_two_ranks_match_single_handle(built, likname):
    """C2's sharding: the minibatch split over two ranks, ONE all-reduce of the packed statistics per step
    inside agp_svgp_cavi_step_multi; eta and the ELBO land on the single-handle run."""
    import agp_amd as AGP
    from agp_amd import capi
    from agp_amd import parallel as P

    K = 1 if likname == "logistic" else 3
    rng = np.random.default_rng(2)
    X, y, Z = _data(rng, K=K)
    lik = (lambda: AGP.LogisticLikelihood()) if K == 1 else (lambda: AGP.LogisticSoftMaxLikelihood(3))
    B, iters = 128, 5
    idx = [rng.choice(len(X), B, replace=False) for _ in range(iters)]
    ref = AGP.SVGP(_kernel(AGP), lik(), AGP.AnalyticSVI(B), Z, optimiser=False)
    elbo_ref = []
    AGP.train_(ref, X, y, iters, idx_stream=idx, callback=lambda m, s, i: elbo_ref.append(AGP.objective(m, s)))

    def body(rank, group):
        m = AGP.SVGP(_kernel(AGP), lik(), AGP.AnalyticSVI(B), Z, optimiser=False)
        comm = P.Comm.from_group(m, group, rank, 2)
        eng = P.HipEngine(m, B // 2).bind_data(X, y).set_batch_shard(rank, 2)
        elbos = []
        for it in range(iters):
            eng.step_multi(P.shard_batch(idx[it], 2, rank), len(X) / B, capi.SHARD_BATCH, comm)
            elbos.append(eng.elbo_multi(capi.SHARD_BATCH, comm))
        eng.check()
        n, nbytes, _ = comm.stats()
        nt = (m.m + 63) // 64
        per_step = m.n_latent * (nt * 64 + nt * (nt + 1) // 2 * 4096) * 8  # the packed triangle, not mp x mp
        assert nbytes == iters * (per_step + 16)  # + the two ELBO scalars
        return [m.get_state(k) for k in range(m.n_latent)], elbos

    res = _thread_ranks(2, body)
    for states, elbos in res:
        for k, st in enumerate(states):
            r = ref.get_state(k)
            assert _rel(st[3], r[3]) < 1e-9 and _rel(st[2], r[2]) < 1e-9 and _rel(st[0], r[0]) < 1e-8
        assert np.allclose(elbos, elbo_ref, rtol=1e-8)
    assert res[0][1] == res[1][1]  # identical on every rank


@pytest.mark.parametrize("overlap", [False, True])
def test_batch_parallel_step_rides_on_the_task_graph_launches(built, overlap, monkeypatch):
    """(overlap: AGP_SPLIT_OVERLAP=1 -- the statistics travel in block-column groups on the communicator's own stream and the tile
    workgroups of the next launch wait for their column's group; same assertions.)
    Round 3: with the next minibatch announced (agp_svgp_prefetch) and nobody looking in between, a batch-parallel step over two
    ranks leaves its eta step PENDING on the reduced statistics; the next step's task-graph launch takes it as its prologue (no
    k_eta2_from_packed) and does the row statistics as its epilogue.  Same trajectory as the single handle, ranks identical, and the
    step counters say the scheduling really was in use.  4 block columns, 128 points per rank (two kappa block rows)."""
    import agp_amd as AGP
    from agp_amd import capi
    from agp_amd import parallel as P

    if overlap:
        monkeypatch.setenv("AGP_SPLIT_OVERLAP", "1")
    rng = np.random.default_rng(9)
    N, D, m, B, iters = 3000, 4, 200, 256, 12
    X = rng.random((N, D))
    f = np.sin(4 * X[:, 0]) + X[:, 1] - 1.0
    y = np.sign(f + 0.3 * rng.standard_normal(N))
    y[y == 0] = 1.0
    Z = X[rng.permutation(N)[:m]].copy()
    idx = [rng.choice(N, B, replace=False) for _ in range(iters)]
    kern = lambda: 1.5 * (AGP.SqExponentialKernel() @ AGP.ScaleTransform(2.0))
    ref = AGP.SVGP(kern(), AGP.LogisticLikelihood(), AGP.AnalyticSVI(B), Z, optimiser=False)
    AGP.train_(ref, X, y, iters, idx_stream=idx)

    def body(rank, group):
        mdl = AGP.SVGP(kern(), AGP.LogisticLikelihood(), AGP.AnalyticSVI(B), Z, optimiser=False)
        comm = P.Comm.from_group(mdl, group, rank, 2)
        eng = P.HipEngine(mdl, B // 2).bind_data(X, y).set_batch_shard(rank, 2)
        nxt = P.shard_batch(idx[0], 2, rank)
        for it in range(iters):
            cur = nxt
            eng.step_multi(cur, N / B, capi.SHARD_BATCH, comm)
            if it + 1 < iters:
                nxt = eng.prefetch(P.shard_batch(idx[it + 1], 2, rank))
        eng.check()
        counters = eng.step_counters()
        return mdl.get_state(0), counters

    res = _thread_ranks(2, body)
    r = ref.get_state(0)
    for st, (n, npro) in res:
        if not (KN.no_prologue() or KN.forced("AGP_SPLIT_MERGED")):
            assert npro >= iters - 2, (n, npro)  # every step after the first rode on its successor's launch
        assert _rel(st[3], r[3]) < 1e-9 and _rel(st[2], r[2]) < 1e-9 and _rel(st[0], r[0]) < 1e-8
    for a, b in zip(res[0][0], res[1][0]):
        assert np.array_equal(a, b)  # the replicas stay bitwise together


def test_batch_parallel_merged_step_survives_aborted_launches(built):
    """The same two-rank run in a fresh process with AGP_DAG_TEST_ABORT=1 (read once per process): every task-graph launch is treated
    as having lost a dependency, so every step goes through the in-stream fallback -- which must find eta already stepped by the
    prologue (from the reduced statistics) and redo the rows the epilogue wrote.  Same assertions as above."""
    import subprocess
    import sys

    env = dict(os.environ, AGP_DAG_TEST_ABORT="1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-m", "gpu", "-p", "no:cacheprovider", "-k",
                        "test_batch_parallel_step_rides_on_the_task_graph_launches"], env=env, capture_output=True, text=True,
                       timeout=900, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0 and "2 passed" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]


def test_latent_parallel_lsm_and_tied_hyper_two_ranks(built):
    """C4's sharding: 4-class LogisticSoftMax, two latents per rank, sum_k gamma_k all-reduced twice per step; then the
    tied-Z hyper step (gradient summed over latents and ranks, one all-reduce of 1 + D + m D doubles)."""
    import agp_amd as AGP
    from agp_amd import capi
    from agp_amd import parallel as P

    rng = np.random.default_rng(4)
    X, y, Z = _data(rng, K=4)
    B, iters = 100, 5
    idx = [rng.choice(len(X), B, replace=False) for _ in range(iters)]

    def make(slice_):
        return AGP.SVGP(_kernel(AGP), AGP.LogisticSoftMaxLikelihood(4), AGP.AnalyticSVI(B), Z, optimiser=AGP.ADAM(0.01),
                        Zoptimiser=AGP.ADAM(0.001), latent_slice=slice_)

    def run(m, comm, world, rank):
        eng = P.HipEngine(m, B).bind_data(X, y)
        elbos = []
        for it in range(iters):
            eng.step_multi(idx[it], len(X) / B, capi.SHARD_LATENT, comm)
            elbos.append(eng.elbo_multi(capi.SHARD_LATENT, comm))
            if it >= 2:
                eng.hyper_step_multi(comm, tied=True)
        eng.check()
        m._pull_hypers()
        return [m.get_state(k) for k in range(m.n_latent)], elbos, [(k.variance, k.transform.s) for k in m.kernels], m.Zs

    one = run(make(None), None, 1, 0)

    def body(rank, group):
        m = make(P.latent_slice(4, 2, rank))
        return run(m, P.Comm.from_group(m, group, rank, 2), 2, rank)

    res = _thread_ranks(2, body)
    for rank, (states, elbos, kers, Zs) in enumerate(res):
        lo, hi = P.latent_slice(4, 2, rank)
        for k in range(hi - lo):
            assert _rel(states[k][3], one[0][lo + k][3]) < 1e-9 and _rel(states[k][0], one[0][lo + k][0]) < 1e-8
            assert kers[k][0] == pytest.approx(one[2][lo + k][0], rel=1e-10)
            assert kers[k][1] == pytest.approx(one[2][lo + k][1], rel=1e-10)
            assert _rel(Zs[k], one[3][lo + k]) < 1e-10
        assert np.allclose(elbos, one[1], rtol=1e-9)
    # tied: every latent took the same step
    assert len({round(v, 14) for v, _ in one[2]}) == 1 and abs(one[2][0][0] - 1.5) > 1e-4


# ---- two processes, one GPU, no torch.distributed ---------------------------------------------------------------------
def _proc_rank(rank, world, shm_name, nbytes, bar, q, mode):
    try:
        import sys

        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from multiprocessing import shared_memory

        import torch

        import agp_amd as AGP
        from agp_amd import capi
        from agp_amd import parallel as P

        shm = shared_memory.SharedMemory(name=shm_name)
        slots = np.ndarray((world, nbytes // 8), dtype=np.float64, buffer=shm.buf)
        rng = np.random.default_rng(6)
        big = "-big" in mode
        overlap = mode.endswith("-overlap")  # AGP_SPLIT_OVERLAP=1 (set by the parent) + the look-ahead: the step rides on the launches
        mode = mode.split("-")[0]
        K = 1 if mode == "batch" else 3
        X, y, Z = _data(rng, K=K, **(BIG if big else {}))
        B, iters = (2048, 12) if big else (128, 4)
        idx = [rng.choice(len(X), B, replace=False) for _ in range(iters)]
        lik = AGP.LogisticLikelihood() if K == 1 else AGP.LogisticSoftMaxLikelihood(3)
        sl = None if mode == "batch" else P.latent_slice(3, world, rank)
        m = AGP.SVGP(_kernel(AGP), lik, AGP.AnalyticSVI(B), Z, optimiser=False, latent_slice=sl)
        seen = []

        def allreduce(ptr, count, dtype, stream):
            assert dtype == capi.F64 and count * 8 <= nbytes
            seen.append((count, bool(stream)))
            with P.on_stream(stream):  # the stream the library names: the ctx's, or the communicator's own (column groups)
                t = torch.as_tensor(P._DevBuf(ptr, count, "<f8"), device="cuda")
                slots[rank, :count] = t.cpu().numpy()  # orders after the work already enqueued on that stream
                bar.wait()
                s = slots[:, :count].sum(axis=0)
                bar.wait()
                t.copy_(torch.from_numpy(s))

        comm = P.Comm.from_callback(m, rank, world, allreduce)
        assert not comm.is_rccl
        eng = P.HipEngine(m, B // world if mode == "batch" else B).bind_data(X, y)
        if mode == "batch":
            eng.set_batch_shard(rank, world)
        nxt = P.shard_batch(idx[0], world, rank) if mode == "batch" else None
        for it in range(iters):
            if mode == "batch":
                eng.step_multi(nxt, len(X) / B, capi.SHARD_BATCH, comm)
                if it + 1 < iters:
                    nxt = P.shard_batch(idx[it + 1], world, rank)
                    if overlap:
                        nxt = eng.prefetch(nxt)
            else:
                eng.step_multi(idx[it], len(X) / B, capi.SHARD_LATENT, comm)
        eng.check()
        if overlap and not (KN.no_prologue() or KN.forced("AGP_SPLIT_MERGED")):
            # the statistics really travelled as several ranges on a stream of the communicator's own
            per_step = len(seen) // iters
            assert per_step >= 2 and all(on_side for _, on_side in seen), (per_step, seen[:8])
            assert eng.step_counters()[1] >= iters - 2
        q.put((rank, [m.get_state(k)[3] for k in range(m.n_latent)]))
    except BaseException as e:
        q.put((rank, repr(e)))
        try:
            bar.abort()
        except Exception:
            pass


@pytest.mark.parametrize("mode", ["batch", "latent", "batch-big", "batch-big-overlap", "batch-big-abort-overlap"])
def test_two_processes_share_one_gpu_without_torch_distributed(built, mode, monkeypatch):
    """Two host processes, each with its own ctx / handle on GPU 0, drive a sharded run through the C ABI only; the
    all-reduce is a host callback over POSIX shared memory.  Doubles as the "second process on the same GPU" check: both
    processes factor with the one-launch task graph at the same time.  At the C2 size ("batch-big") the two 408-workgroup task
    graphs can starve each other of a dependency (their deadlock-freedom argument holds for one graph per device): the bounded
    spin then hands the factorisation to the in-stream fallback k_chol_safe, and the run must still land on the
    single-process result -- a Cholesky either succeeds or throws PosDefException, it never stalls or fails the step."""
    from multiprocessing import shared_memory

    import agp_amd as AGP

    big = "-big" in mode
    if mode.endswith("-overlap"):  # the children inherit the environment (spawn)
        monkeypatch.setenv("AGP_SPLIT_OVERLAP", "1")
    if "-abort" in mode:  # ... and every task-graph launch is made to lose a dependency: fallback behind gated prologues
        monkeypatch.setenv("AGP_DAG_TEST_ABORT", "1")
    world, nbytes = 2, 8 * ((1 << 20) if big else (1 << 16))
    ctx = mp.get_context("spawn")
    shm = shared_memory.SharedMemory(create=True, size=world * nbytes)
    try:
        bar, q = ctx.Barrier(world), ctx.Queue()
        ps = [ctx.Process(target=_proc_rank, args=(r, world, shm.name, nbytes, bar, q, mode)) for r in range(world)]
        for p in ps:
            p.start()
        got = dict(q.get(timeout=600) for _ in range(world))
        for p in ps:
            p.join(timeout=60)
    finally:
        shm.close()
        shm.unlink()
    for r in range(world):
        assert not isinstance(got[r], str), got[r]
    rng = np.random.default_rng(6)
    mode = mode.split("-")[0]
    K = 1 if mode == "batch" else 3
    X, y, Z = _data(rng, K=K, **(BIG if big else {}))
    B, iters = (2048, 12) if big else (128, 4)
    idx = [rng.choice(len(X), B, replace=False) for _ in range(iters)]
    lik = AGP.LogisticLikelihood() if K == 1 else AGP.LogisticSoftMaxLikelihood(3)
    ref = AGP.SVGP(_kernel(AGP), lik, AGP.AnalyticSVI(B), Z, optimiser=False)
    AGP.train_(ref, X, y, iters, idx_stream=idx)
    if mode == "batch":
        for r in range(world):
            assert _rel(got[r][0], ref.get_state(0)[3]) < 1e-9
        assert np.array_equal(got[0][0], got[1][0])  # the replicas stay bitwise together
    else:
        from agp_amd import parallel as P

        for r in range(world):
            lo, hi = P.latent_slice(3, world, r)
            for k in range(hi - lo):
                assert _rel(got[r][k], ref.get_state(lo + k)[3]) < 1e-9


def test_sharded_multioutput_through_the_multi_calls(built):
    """C5's sharding: the latents of a multi-output model over two ranks, (mean_f, var_f) exchanged inside
    agp_svgp_cavi_step_multi, predictions through agp_svgp_predict_multi; against the single-handle model."""
    import agp_amd as AGP
    from agp_amd import capi
    from agp_amd import parallel as P
    from test_gpu_sharded import _build

    from oracle import agp_ref as R

    m1, X, ys, idx, N, B, iters, Q = _build(AGP, R, None, hyper=True)
    e1 = P.HipEngine(m1, B).bind_data(X, ys)
    Xt = np.random.default_rng(3).random((37, X.shape[1]))
    el1 = []
    for it in range(iters):
        e1.step_multi(idx[it], N / B, capi.SHARD_LATENT, None)
        el1.append(e1.elbo_multi(capi.SHARD_LATENT, None))
        if it >= 1:
            e1.hyper_step_multi(None, tied=False)
    e1.check()
    f1 = AGP.predict_f(m1, Xt, cov=True)
    p1 = AGP.proba_y(m1, Xt)

    def body(rank, group):
        m, *_ = _build(AGP, R, P.latent_slice(Q, 2, rank), hyper=True)
        comm = P.Comm.from_group(m, group, rank, 2)
        eng = P.HipEngine(m, B).bind_data(X, ys)
        el = []
        for it in range(iters):
            eng.step_multi(idx[it], N / B, capi.SHARD_LATENT, comm)
            el.append(eng.elbo_multi(capi.SHARD_LATENT, comm))
            if it >= 1:
                eng.hyper_step_multi(comm, tied=False)
        eng.check()
        return el, eng.predict_multi(Xt, "f", comm), eng.predict_multi(Xt, "proba", comm), m.get_A()

    res = _thread_ranks(2, body)
    for el, f, p, A in res:
        assert np.allclose(el, el1, rtol=1e-9)
        for t in range(m1.n_task):
            assert _rel(f[0][t], f1[0][t]) < 1e-9 and _rel(f[1][t], f1[1][t]) < 1e-8
            assert _rel(p[0][t], p1[t][0]) < 1e-8
        assert _rel(A, m1.get_A()) < 1e-10
