"""Quick GPU probe: MFMA ceilings + timing of one C2-shaped CAVI step (used while developing; bench.py is the contract)."""
import ctypes as C
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import __graft_entry__ as g
g.build()
import agp_amd as AGP
from agp_amd import capi

L = capi.lib()
ctx = C.c_void_p()
assert L.agp_ctx_create(0, C.c_void_p(torch.cuda.current_stream().cuda_stream), C.byref(ctx)) == 0
for dt, name in ((0, "f64"), (1, "f32")):
    t = C.c_double()
    assert L.agp_mfma_peak(ctx, dt, C.byref(t)) == 0
    print(f"mfma_peak {name}: {t.value:.1f} TFLOP/s")

def run(m, B, D, N, T, steps=30):
    rng = np.random.default_rng(0)
    X = rng.random((N, D)); w = rng.standard_normal(D)
    y = np.sign(np.sin(X @ w) + 0.1 * rng.standard_normal(N))
    Z = X[rng.permutation(N)[:m]].copy()
    ell = np.sqrt(D) / 4
    model = AGP.SVGP(AGP.with_lengthscale(AGP.SqExponentialKernel(), ell), AGP.LogisticLikelihood(), AGP.AnalyticSVI(B), Z, optimiser=False, T=T)
    idx = [rng.choice(N, B, replace=False) for _ in range(steps + 5)]
    AGP.train_(model, X, y, 5, idx_stream=idx[:5])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    AGP.train_(model, X, y, steps, idx_stream=idx[5:], state=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    flops = 6 * B * m * m + m ** 3 + B * m * (3 * D + 12)
    print(f"m={m} B={B} D={D} {np.dtype(T).name}: {dt*1e3:.3f} ms/iter  {1/dt:.1f} iter/s  {flops/dt/1e12:.2f} TFLOP/s(alg)")

run(1024, 1024, 32, 100000, np.float64)
run(2048, 2048, 64, 100000, np.float32)
run(4096, 4096, 64, 100000, np.float64, steps=5)
