"""Development aid: synchronous vs enqueued ELBO (fresh_local = 1 on an evaluation batch) in the bench's setting."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import agp_amd as AGP
from agp_amd import capi
m, B, D, N, EVAL = 1024, 1024, 32, 200000, 8192
rng = np.random.default_rng(0)
X = rng.random((N, D)); y = np.sign(np.sin(X @ rng.standard_normal(D)) + 0.1 * rng.standard_normal(N))
Z = X[rng.permutation(N)[:m]].copy()
idx = np.stack([rng.choice(N, B, replace=False) for _ in range(64)])
ev = torch.as_tensor(rng.choice(N, EVAL, replace=False).astype(np.int64), device="cuda")
model = AGP.SVGP(AGP.with_lengthscale(AGP.SqExponentialKernel(), np.sqrt(D) / 4), AGP.LogisticLikelihood(), AGP.AnalyticSVI(B), Z, optimiser=False)
model.inference.rho = N / B
Xd = model._upload(X, 1); yd = model._upload_y(model._treat(y)); model._data = (Xd, yd, N)
h = model._ensure_handle(EVAL)
L = capi.lib()
model._chk(L.agp_svgp_refresh_K(h))
ia = torch.as_tensor(idx, device="cuda")
xp, yp, ld = C.c_void_p(Xd.data_ptr()), C.c_void_p(yd.data_ptr()), Xd.stride(0)
e, tk, rdy = C.c_double(), C.c_int32(), C.c_int32()
mode = sys.argv[1] if len(sys.argv) > 1 else "pair"
it = 0
for chk in range(6):
    for _ in range(10):
        assert L.agp_svgp_cavi_step(h, xp, ld, yp, C.c_void_p(ia[it % 64].data_ptr()), B, N / B) == 0
        it += 1
        L.agp_svgp_prefetch(h, xp, ld, C.c_void_p(ia[it % 64].data_ptr()), B)
    if mode == "async_late":  # fetched one check later, with ten more steps enqueued behind the evaluation
        prev = tk.value if chk > 0 else None
        model._chk(L.agp_svgp_elbo_enqueue(h, xp, ld, yp, C.c_void_p(ev.data_ptr()), EVAL, N / EVAL, 1, C.byref(tk)))
        if prev is not None:
            model._chk(L.agp_svgp_elbo_fetch(h, prev, 1, C.byref(e), C.byref(rdy))); print(it - 10, "async late", e.value)
    elif mode == "sync":
        model._chk(L.agp_svgp_elbo(h, xp, ld, yp, C.c_void_p(ev.data_ptr()), EVAL, N / EVAL, 1, C.byref(e))); print(it, "sync ", e.value)
    elif mode == "async_wait":
        model._chk(L.agp_svgp_elbo_enqueue(h, xp, ld, yp, C.c_void_p(ev.data_ptr()), EVAL, N / EVAL, 1, C.byref(tk)))
        model._chk(L.agp_svgp_elbo_fetch(h, tk.value, 1, C.byref(e), C.byref(rdy))); print(it, "async+wait", e.value)
    else:  # both on the same state: enqueue, fetch, then the synchronous one
        model._chk(L.agp_svgp_elbo_enqueue(h, xp, ld, yp, C.c_void_p(ev.data_ptr()), EVAL, N / EVAL, 1, C.byref(tk)))
        model._chk(L.agp_svgp_elbo_fetch(h, tk.value, 1, C.byref(e), C.byref(rdy))); a = e.value
        model._chk(L.agp_svgp_elbo(h, xp, ld, yp, C.c_void_p(ev.data_ptr()), EVAL, N / EVAL, 1, C.byref(e)))
        print(it, "async", a, "sync", e.value, "rel", abs(a - e.value) / abs(e.value))
