"""Soak / determinism check of the one-launch task-graph Cholesky: the same C2-shaped run twice (same index stream), `steps`
CAVI steps each, final natural parameters compared BITWISE.  Every reduction in the library has a fixed order, so any difference
would mean a tile was read before it was complete (the hand-over uses coherent stores / loads and flags instead of fences)."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import agp_amd as AGP
from agp_amd import capi

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
m = B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024          # 2048 with fp32 = the C3 shape (32 block columns)
TT = np.float32 if (len(sys.argv) > 3 and sys.argv[3] == "f32") else np.float64
if len(sys.argv) > 5:  # m != B (e.g. 20000 1024 f32 cavi 2048: the fp32 launch WITH the prologue that runs split since round 6)
    B = int(sys.argv[5])
PHASES = len(sys.argv) > 4 and sys.argv[4] == "phases"  # step_local / step_stats / prefetch / step_global (the batch-parallel driver's sequence)
D, N = 32, 200000
rng = np.random.default_rng(0)
X = rng.random((N, D))
y = np.sign(np.sin(X @ rng.standard_normal(D)) + 0.1 * rng.standard_normal(N))
Z = X[rng.permutation(N)[:m]].copy()
idx = np.stack([rng.choice(N, B, replace=False) for _ in range(256)])
out = []
for rep in range(2):
    model = AGP.SVGP(AGP.with_lengthscale(AGP.SqExponentialKernel(), np.sqrt(D) / 4), AGP.LogisticLikelihood(), AGP.AnalyticSVI(B), Z, optimiser=False, T=TT)
    AGP.train_(model, X, y, 1, idx_stream=idx[:1])
    L, h = capi.lib(), model._h
    Xd, yd, _ = model._data
    ia = torch.as_tensor(idx, device="cuda")
    for i in range(steps):
        j = i % 256
        if PHASES:
            assert L.agp_svgp_step_local(h, C.c_void_p(Xd.data_ptr()), Xd.stride(0), C.c_void_p(yd.data_ptr()), C.c_void_p(ia[j].data_ptr()), B, N / B) == 0
            assert L.agp_svgp_step_stats(h) == 0
            L.agp_svgp_prefetch(h, C.c_void_p(Xd.data_ptr()), Xd.stride(0), C.c_void_p(ia[(j + 1) % 256].data_ptr()), B)
            assert L.agp_svgp_step_global(h) == 0
            continue
        assert L.agp_svgp_cavi_step(h, C.c_void_p(Xd.data_ptr()), Xd.stride(0), C.c_void_p(yd.data_ptr()), C.c_void_p(ia[j].data_ptr()), B, N / B) == 0
        L.agp_svgp_prefetch(h, C.c_void_p(Xd.data_ptr()), Xd.stride(0), C.c_void_p(ia[(j + 1) % 256].data_ptr()), B)
    model._chk(L.agp_svgp_check_status(h))
    nfb = C.c_int64(-1)
    L.agp_ctx_task_graph_fallbacks(model._ctx, C.byref(nfb))
    mu, Sig, e1, e2 = model.get_state(0)
    out.append((e1.copy(), e2.copy(), mu.copy()))
    import hashlib
    print(f"run {rep}: {steps} steps, task_graph_fallbacks = {nfb.value}, |eta1| = {np.linalg.norm(e1):.6e}, finite = {np.isfinite(e2).all()}, "
          f"sha1(eta2) = {hashlib.sha1(np.ascontiguousarray(e2).tobytes()).hexdigest()[:16]}")
same = all(np.array_equal(a, b) for a, b in zip(out[0], out[1]))
print("bitwise identical:", same, " max |d eta2| =", np.max(np.abs(out[0][1] - out[1][1])))
sys.exit(0 if same else 1)
