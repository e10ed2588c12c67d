"""ms per CAVI step of a K-latent LogisticSoftMax model (C4 shape: m = 1024, B = 1024, D = 32) on one GPU:
the Python train_ loop and the raw C-ABI loop (as bench.py drives it)."""
import sys, os, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import agp_amd as AGP
from agp_amd import capi
K = int(sys.argv[1]) if len(sys.argv) > 1 else 8
m, B, D, N = 1024, 1024, 32, 200000
rng = np.random.default_rng(0)
X = rng.random((N, D))
y = 1 + (rng.integers(K, size=N))
Z = X[rng.permutation(N)[:m]].copy()
model = AGP.SVGP(AGP.with_lengthscale(AGP.SqExponentialKernel(), np.sqrt(D) / 4), AGP.LogisticSoftMaxLikelihood(K), AGP.AnalyticSVI(B), Z, optimiser=False)
idx = [rng.choice(N, B, replace=False) for _ in range(60)]
AGP.train_(model, X, y, 10, idx_stream=idx[:10])
torch.cuda.synchronize(); t0 = time.perf_counter()
AGP.train_(model, X, y, 50, idx_stream=idx[10:], state=True)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 50
print(f"K={K} latents, train_ loop : {dt*1e3:.3f} ms/step  ({dt*1e3/K:.3f} ms per latent)")
L = capi.lib(); h = model._h
Xd, yd, _ = model._data
ia = torch.as_tensor(np.stack(idx), device="cuda")
def step(i):
    L.agp_svgp_cavi_step(h, C.c_void_p(Xd.data_ptr()), Xd.stride(0), C.c_void_p(yd.data_ptr()), C.c_void_p(ia[i].data_ptr()), B, N / B)
    if i + 1 < 60: L.agp_svgp_prefetch(h, C.c_void_p(Xd.data_ptr()), Xd.stride(0), C.c_void_p(ia[i + 1].data_ptr()), B)
for i in range(10): step(i)
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(10, 60): step(i)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 50
print(f"K={K} latents, raw ABI loop: {dt*1e3:.3f} ms/step  ({dt*1e3/K:.3f} ms per latent)")
for i in range(3):
    t0 = time.perf_counter(); step(20 + i); t1 = time.perf_counter()
    print(f"   host time of one cavi_step+prefetch call: {(t1-t0)*1e3:.3f} ms")
torch.cuda.synchronize()
