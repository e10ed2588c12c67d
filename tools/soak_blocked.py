"""Soak of the blocked factorisation (diagonal block / panel / trailing launches with the look-ahead on a side stream): the same
matrix factored `reps` times, every factor compared BITWISE with the first -- a race between the two streams would show up as a
difference.  usage (GPU box): python tools/soak_blocked.py [n] [reps]"""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as g
g.build()
from agp_amd import capi
L = capi.lib()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 6144
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
ctx = C.c_void_p()
assert L.agp_ctx_create(0, C.c_void_p(torch.cuda.current_stream().cuda_stream), C.byref(ctx)) == 0
for dt, tdt in ((0, torch.float64), (1, torch.float32)):
    G = torch.randn(n, n + 64, dtype=tdt, device="cuda")
    A = G @ G.T / n + 0.5 * torch.eye(n, dtype=tdt, device="cuda")
    del G
    first, bad = None, 0
    info = C.c_int32(-1)
    for it in range(reps):
        a = A.clone()
        assert L.agp_potrf_jitter(ctx, dt, a.data_ptr(), n, n, 1e-4, C.byref(info)) == 0 and info.value == 0
        if first is None:
            first = a
        elif not torch.equal(first, a):
            bad += 1
    print(f"{'f64' if dt == 0 else 'f32'} n = {n}: {reps} factorisations, {bad} differ from the first")
    assert bad == 0

# the CAVI step on the blocked path (m = B = 4096: 64 + 65 block rows), with the look-ahead stream: two identical runs, bitwise
import numpy as np
import agp_amd as AGP
m = B = 4096
D, N, steps = 16, 60000, int(sys.argv[3]) if len(sys.argv) > 3 else 12
rng = np.random.default_rng(0)
X = rng.random((N, D))
y = np.sign(np.sin(X @ rng.standard_normal(D)) + 0.1 * rng.standard_normal(N))
Z = X[rng.permutation(N)[:m]].copy()
idx = np.stack([rng.choice(N, B, replace=False) for _ in range(steps)])
outs = []
for rep in range(2):
    model = AGP.SVGP(AGP.with_lengthscale(AGP.SqExponentialKernel(), np.sqrt(D) / 4), AGP.LogisticLikelihood(), AGP.AnalyticSVI(B), Z, optimiser=False)
    AGP.train_(model, X, y, steps, idx_stream=list(idx))
    outs.append(model.get_state(0)[3])
print("blocked CAVI step, two runs of", steps, "steps bitwise identical:", bool(np.array_equal(outs[0], outs[1])), "finite:", bool(np.isfinite(outs[0]).all()))
assert np.array_equal(outs[0], outs[1])
