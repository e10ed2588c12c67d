"""Soak / determinism check of the interleaved multi-problem task graph: a K-class LogisticSoftMax model (C4 shape, m = B = 1024),
the same run twice, final natural parameters of every latent compared bitwise."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import agp_amd as AGP
from agp_amd import capi

K = int(sys.argv[1]) if len(sys.argv) > 1 else 4
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
m = B = 1024
D, N = 32, 200000
rng = np.random.default_rng(0)
X = rng.random((N, D))
y = 1 + rng.integers(K, size=N)
Z = X[rng.permutation(N)[:m]].copy()
idx = np.stack([rng.choice(N, B, replace=False) for _ in range(128)])
out = []
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
for rep in range(reps):
    model = AGP.SVGP(AGP.with_lengthscale(AGP.SqExponentialKernel(), np.sqrt(D) / 4), AGP.LogisticSoftMaxLikelihood(K), AGP.AnalyticSVI(B), Z, optimiser=False)
    AGP.train_(model, X, y, 1, idx_stream=idx[:1])
    L, h = capi.lib(), model._h
    Xd, yd, _ = model._data
    ia = torch.as_tensor(idx, device="cuda")
    for i in range(steps):
        j = i % 128
        assert L.agp_svgp_cavi_step(h, C.c_void_p(Xd.data_ptr()), Xd.stride(0), C.c_void_p(yd.data_ptr()), C.c_void_p(ia[j].data_ptr()), B, N / B) == 0
        L.agp_svgp_prefetch(h, C.c_void_p(Xd.data_ptr()), Xd.stride(0), C.c_void_p(ia[(j + 1) % 128].data_ptr()), B)
    model._chk(L.agp_svgp_check_status(h))
    out.append([model.get_state(k)[3].copy() for k in range(K)])
    print(f"run {rep}: K = {K}, {steps} steps, finite = {all(np.isfinite(e).all() for e in out[-1])}")
same = True
for r in range(1, reps):
    for k in range(K):
        if not np.array_equal(out[0][k], out[r][k]):
            same = False
            d = np.max(np.abs(out[0][k] - out[r][k])) / np.max(np.abs(out[0][k]))
            print(f"   run {r} latent {k}: max rel diff of eta2 vs run 0 = {d:.3e}")
print("bitwise identical:", same)
sys.exit(0 if same else 1)
