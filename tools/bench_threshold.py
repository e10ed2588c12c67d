"""Task-graph vs per-column Cholesky around the switch point: whole CAVI step, fp64, m = B in {1536, 2048, 2560, 3072}.
Run once with AGP_CHOL_DAG=1 and once with AGP_CHOL_DAG=0 (the library reads the variable once)."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import agp_amd as AGP
from agp_amd import capi

def run(m, B, D=32, N=200000, steps=30):
    rng = np.random.default_rng(0)
    X = rng.random((N, D))
    y = np.sign(np.sin(X @ rng.standard_normal(D)) + 0.1 * rng.standard_normal(N))
    Z = X[rng.permutation(N)[:m]].copy()
    model = AGP.SVGP(AGP.with_lengthscale(AGP.SqExponentialKernel(), np.sqrt(D) / 4), AGP.LogisticLikelihood(), AGP.AnalyticSVI(B), Z, optimiser=False)
    idx = np.stack([rng.choice(N, B, replace=False) for _ in range(steps + 6)])
    AGP.train_(model, X, y, 2, idx_stream=idx[:2])
    L, h = capi.lib(), model._h
    Xd, yd, _ = model._data
    ia = torch.as_tensor(idx, device="cuda")
    def step(i):
        assert L.agp_svgp_cavi_step(h, C.c_void_p(Xd.data_ptr()), Xd.stride(0), C.c_void_p(yd.data_ptr()), C.c_void_p(ia[i].data_ptr()), B, N / B) == 0
        if i + 1 < len(idx): L.agp_svgp_prefetch(h, C.c_void_p(Xd.data_ptr()), Xd.stride(0), C.c_void_p(ia[i + 1].data_ptr()), B)
    for i in range(6): step(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(6, 6 + steps): step(i)
    e1.record(); torch.cuda.synchronize()
    model._chk(L.agp_svgp_check_status(h))
    print(f"AGP_CHOL_DAG={os.environ.get('AGP_CHOL_DAG', 'auto')}  m = B = {m}: {e0.elapsed_time(e1) / steps:.3f} ms/step")

for m in (1536, 2048, 2560, 3072):
    run(m, m)
