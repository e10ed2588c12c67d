"""GPU: inducing-point selection (agp_nearest_center / agp_kmeans, csrc/agp_kmeans.h) against the oracle restatement of
`inducingpoints(KmeansAlg(m), X)`; tolerances: labels identical, centres / costs <= 1e-10 relative (fp64)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


@pytest.fixture(scope="module")
def env(built):
    import torch

    import agp_amd as AGP
    from agp_amd import capi
    from oracle import agp_ref as R

    L = capi.lib()
    ctx = C.c_void_p()
    assert L.agp_ctx_create(0, C.c_void_p(torch.cuda.current_stream().cuda_stream), C.byref(ctx)) == 0
    yield dict(torch=torch, AGP=AGP, capi=capi, R=R, L=L, ctx=ctx)
    L.agp_ctx_destroy(ctx)


def _nearest(env, X, Cc, dtype="f64"):
    torch, capi, L, ctx = env["torch"], env["capi"], env["L"], env["ctx"]
    td = torch.float64 if dtype == "f64" else torch.float32
    Xd, Cd = torch.as_tensor(X, dtype=td, device="cuda"), torch.as_tensor(Cc, dtype=td, device="cuda")
    lab = torch.empty(len(X), dtype=torch.int32, device="cuda")
    md = torch.empty(len(X), dtype=td, device="cuda")
    st = L.agp_nearest_center(ctx, capi.F64 if dtype == "f64" else capi.F32, C.c_void_p(Xd.data_ptr()), len(X), Xd.stride(0),
                              X.shape[1], C.c_void_p(Cd.data_ptr()), Cd.stride(0), len(Cc), C.c_void_p(lab.data_ptr()),
                              C.c_void_p(md.data_ptr()))
    assert st == 0
    torch.cuda.synchronize()
    return lab.cpu().numpy(), md.cpu().numpy()


@pytest.mark.parametrize("N,m,D", [(1000, 37, 5), (64, 64, 16), (777, 130, 33), (5, 1, 2), (3000, 200, 128)])
def test_nearest_center_matches_oracle(env, N, m, D):
    R = env["R"]
    rng = np.random.default_rng(N + m)
    X, Cc = rng.random((N, D)), rng.random((m, D))
    lab, md = _nearest(env, X, Cc)
    lr, mr = R.nearest_center(X, Cc)
    assert np.array_equal(lab, lr)
    assert np.allclose(md, mr, rtol=1e-11, atol=1e-13)
    lab32, md32 = _nearest(env, X, Cc, "f32")
    assert np.mean(lab32 == lr) > 0.99 and np.allclose(md32, mr, rtol=2e-3, atol=1e-4)


def test_nearest_center_ties_and_duplicates(env):
    """duplicated centres: the smaller index wins (numpy argmin convention), also across 64-wide centre tiles"""
    rng = np.random.default_rng(2)
    base = rng.random((70, 4))
    Cc = np.concatenate([base, base])  # centre j and j + 70 coincide
    lab, md = _nearest(env, base, Cc)
    assert np.array_equal(lab, np.arange(70)) and np.all(md < 1e-12)


@pytest.mark.parametrize("N,m,D", [(600, 12, 3), (5000, 100, 7), (20000, 64, 32)])
def test_kmeans_lloyd_matches_oracle(env, N, m, D):
    torch, capi, R, L, ctx = env["torch"], env["capi"], env["R"], env["L"], env["ctx"]
    rng = np.random.default_rng(N)
    centers = rng.random((m, D)) * 4
    X = centers[rng.integers(m, size=N)] + 0.15 * rng.standard_normal((N, D))
    seeds = X[rng.choice(N, m, replace=False)].copy()
    Cr, labr, itr, objr, convr = R.kmeans_lloyd(X, seeds, tol=1e-3, maxiter=100)
    Xd = torch.as_tensor(X, device="cuda")
    Cd = torch.as_tensor(seeds, device="cuda").contiguous()
    lab = torch.empty(N, dtype=torch.int32, device="cuda")
    cnt = torch.empty(m, dtype=torch.int32, device="cuda")
    it, conv, obj = C.c_int32(), C.c_int32(), C.c_double()
    st = L.agp_kmeans(ctx, capi.F64, C.c_void_p(Xd.data_ptr()), N, Xd.stride(0), D, C.c_void_p(Cd.data_ptr()), D, m, 100, 1e-3,
                      C.c_void_p(lab.data_ptr()), C.c_void_p(cnt.data_ptr()), C.byref(it), C.byref(obj), C.byref(conv))
    assert st == 0
    assert it.value == itr and bool(conv.value) == convr
    assert np.array_equal(lab.cpu().numpy(), labr)
    assert _rel(Cd.cpu().numpy(), Cr) < 1e-10
    assert obj.value == pytest.approx(objr, rel=1e-10)
    # counts belong to the assignment the last centre update used; they cover every point
    assert int(cnt.sum().item()) == N


def test_inducingpoints_kmeans_end_to_end(env):
    """AGP.inducingpoints(KmeansAlg(m), X; rng) == the oracle's kmeans_inducingpoints with the same generator, then an SVGP
    built on those inducing points trains (the reference's test/testingtools.jl:66 pattern)."""
    AGP, R = env["AGP"], env["R"]
    rng = np.random.default_rng(5)
    N, D, m = 3000, 2, 20
    X = rng.random((N, D))
    Z, info = AGP.inducingpoints(AGP.KmeansAlg(m), X, rng=np.random.default_rng(99), return_info=True)
    Zr = R.kmeans_inducingpoints(X, m, np.random.default_rng(99))
    assert np.array_equal(info["seeds"], R.kmeans_seeding(X, m, 10, np.random.default_rng(99)))
    assert _rel(Z, Zr) < 1e-10 and info["converged"]
    y = np.sin(5 * X[:, 0]) * np.cos(3 * X[:, 1]) + 0.05 * rng.standard_normal(N)
    model = AGP.SVGP(AGP.SqExponentialKernel() @ AGP.ScaleTransform(4.0), AGP.GaussianLikelihood(0.01), AGP.AnalyticVI(), Z,
                     optimiser=False)
    AGP.train_(model, X, y, 3)
    assert np.mean(np.abs(AGP.predict_y(model, X) - y)) < 0.15
    Zs = AGP.inducingpoints(AGP.RandomSubset(m), X, rng=np.random.default_rng(1))
    assert Zs.shape == (m, D) and all(any(np.allclose(z, x) for x in X) for z in Zs[:3])


def test_kmeans_full_size_properties(env):
    """C2-sized selection (N = 1e6, D = 32, m = 1024): cost decreases monotonically from the seeds, every point is assigned,
    the result is a fixed point of one more assignment, and it finishes in well under a second per iteration."""
    import time

    torch, capi, L, ctx = env["torch"], env["capi"], env["L"], env["ctx"]
    N, D, m = 1_000_000, 32, 1024
    g = torch.Generator(device="cuda")
    g.manual_seed(0)
    X = torch.rand(N, D, dtype=torch.float64, device="cuda", generator=g)
    seeds = X[torch.randperm(N, device="cuda", generator=g)[:m]].contiguous()
    costs = []
    Cd = seeds.clone()
    it, conv, obj = C.c_int32(), C.c_int32(), C.c_double()
    for k in range(3):  # one Lloyd iteration per call to watch the cost
        st = L.agp_kmeans(ctx, capi.F64, C.c_void_p(X.data_ptr()), N, X.stride(0), D, C.c_void_p(Cd.data_ptr()), D, m, 1, 0.0,
                          None, None, C.byref(it), C.byref(obj), C.byref(conv))
        assert st == 0
        costs.append(obj.value)
    assert costs[0] > costs[1] > costs[2] > 0
    cnt = torch.empty(m, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    st = L.agp_kmeans(ctx, capi.F64, C.c_void_p(X.data_ptr()), N, X.stride(0), D, C.c_void_p(Cd.data_ptr()), D, m, 5, 0.0, None,
                      C.c_void_p(cnt.data_ptr()), C.byref(it), C.byref(obj), C.byref(conv))
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    assert st == 0 and it.value == 5 and obj.value < costs[2]
    assert int(cnt.sum().item()) == N and torch.isfinite(Cd).all()
    print(f"\nk-means N=1e6 m=1024 D=32 fp64: {dt * 1e3:.1f} ms per Lloyd iteration")
    assert dt < 0.5
