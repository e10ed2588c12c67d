"""Raw C-ABI step time (cavi_step + prefetch, indices resident) for the other BASELINE.json shapes on one GPU:
   C3: Matern52 + StudentT, m = B = 2048, D = 64, fp32      C5 (one latent): SE + Gaussian, m = B = 4096, D = 64, fp64"""
import sys, os, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import agp_amd as AGP
from agp_amd import capi

def run(name, kernel, lik, m, B, D, T, N=300000, steps=60):
    rng = np.random.default_rng(0)
    X = rng.random((N, D))
    y = np.sin(X @ rng.standard_normal(D)) + 0.1 * rng.standard_normal(N)
    Z = X[rng.permutation(N)[:m]].copy()
    model = AGP.SVGP(kernel, lik, AGP.AnalyticSVI(B), Z, optimiser=False, T=T)
    idx = np.stack([rng.choice(N, B, replace=False) for _ in range(steps + 10)])
    AGP.train_(model, X, y, 2, idx_stream=idx[:2])
    L, h = capi.lib(), model._h
    Xd, yd, _ = model._data
    ia = torch.as_tensor(idx, device="cuda")
    def step(i):
        st = L.agp_svgp_cavi_step(h, C.c_void_p(Xd.data_ptr()), Xd.stride(0), C.c_void_p(yd.data_ptr()), C.c_void_p(ia[i].data_ptr()), B, N / B)
        assert st == 0
        if i + 1 < len(idx): L.agp_svgp_prefetch(h, C.c_void_p(Xd.data_ptr()), Xd.stride(0), C.c_void_p(ia[i + 1].data_ptr()), B)
    for i in range(10): step(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(10, 10 + steps): step(i)
    e1.record(); torch.cuda.synchronize()
    dt = e0.elapsed_time(e1) / steps * 1e-3  # stream time (a device-wide sync also waits for deferred frees of earlier models)
    model._chk(L.agp_svgp_check_status(h))
    fl = 6 * B * m * m + m ** 3 + B * m * (3 * D + 12)
    print(f"{name}: {dt*1e3:.3f} ms/step = {1/dt:.0f} iter/s ; algorithmic {fl/dt/1e12:.1f} TFLOP/s")

ell = lambda D: np.sqrt(D) / 4
run("C3 shape (Matern52+StudentT, m=B=2048, D=64, fp32)", AGP.with_lengthscale(AGP.Matern52Kernel(), ell(64)), AGP.StudentTLikelihood(3.0), 2048, 2048, 64, np.float32)
run("C5 shape, one latent (SE+Gaussian, m=B=4096, D=64, fp64)", AGP.with_lengthscale(AGP.SqExponentialKernel(), ell(64)), AGP.GaussianLikelihood(0.01), 4096, 4096, 64, np.float64, steps=20)
run("C2 shape via this loop (SE+StudentT, m=B=1024, D=32, fp64)", AGP.with_lengthscale(AGP.SqExponentialKernel(), ell(32)), AGP.StudentTLikelihood(3.0), 1024, 1024, 32, np.float64)
