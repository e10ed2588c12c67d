"""Development aid behind docs/DESIGN_LOG.md section 14 (defect 5, "which wait expires"): stress B until a launch loses a dependency, then
the first bounded wait that ran out -- which flag (tile / role), in which workgroup.  Needs a library built with -DAGP_DEBUG_PTRS:

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -DAGP_DEBUG_PTRS -o /tmp/libagp_dbg.so augmentedgaussianprocesses.jl_amd/csrc/agp_capi.hip
    AGP_HIP_LIB=/tmp/libagp_dbg.so AGP_CHAIN_SPLIT=1 AGP_STEP_PROLOGUE=0 python tools/stress/split_which_wait.py 350"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import __graft_entry__ as g  # noqa: E402,F401
import agp_amd as AGP  # noqa: E402
from agp_amd import capi  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
m, B, D, N, steps = 1024, 2048, 16, 50000, 200
if len(sys.argv) > 2:  # round 6: any shape, e.g. `... split_which_wait.py 10 2048 2048 32 20000` = the C3 soak's shape
    m, B, D, steps = (int(v) for v in sys.argv[2:6])
    N = 200000
rng = np.random.default_rng(0)
X = rng.random((N, D))
y = np.sin(X @ rng.standard_normal(D)) + 0.1 * rng.standard_normal(N)
Z = X[rng.permutation(N)[:m]].copy()
idx = np.stack([rng.choice(N, B, replace=False) for _ in range(32)])
L = capi.lib()
dbg = C.CDLL(capi.LIB_PATH)
dbg.agp_debug_dag_diag.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
nt, ne = m // 64, B // 64 + 1
for rep in range(reps):
    model = AGP.SVGP(AGP.with_lengthscale(AGP.Matern52Kernel(), 1.0), AGP.StudentTLikelihood(3.0), AGP.AnalyticSVI(B), Z, optimiser=False, T=np.float32)
    AGP.train_(model, X, y, 1, idx_stream=idx[:1])
    h = model._h
    Xd, yd, _ = model._data
    ia = torch.as_tensor(idx, device="cuda")
    for i in range(steps):
        j = i % 32
        assert L.agp_svgp_cavi_step(h, C.c_void_p(Xd.data_ptr()), Xd.stride(0), C.c_void_p(yd.data_ptr()), C.c_void_p(ia[j].data_ptr()), B, N / B) == 0
        L.agp_svgp_prefetch(h, C.c_void_p(Xd.data_ptr()), Xd.stride(0), C.c_void_p(ia[(j + 1) % 32].data_ptr()), B)
    out = (C.c_ulonglong * 8)()
    assert dbg.agp_debug_dag_diag(model._ctx, out) == 0
    nfb = C.c_int64(0)
    L.agp_ctx_task_graph_fallbacks(model._ctx, C.byref(nfb))
    print(f"run {rep}: {steps} steps, task_graph_fallbacks {nfb.value}, expired waits {int(out[0])}, chain release never came {int(out[5])}", flush=True)
    if out[0] or out[5]:
        fi = (int(out[1]) - int(out[6])) // 4 // int(out[7]) if out[1] else -1
        ntile = (nt + ne) * nt
        if 0 <= fi < ntile:
            what = f"ready flag of tile (R = {fi // nt}, c = {fi % nt})  [R < {nt}: matrix row, else extension row]"
        elif fi >= ntile:
            k = fi - ntile
            what = ["X_k ready (chain)", "tile (k, k-1) parked for the chain", "diagonal tile (k, k) parked for the chain", "beyond"][min(k // nt, 3)] + f", k = {k % nt}"
        else:
            what = "no flag wait expired"
        print(f"run {rep}: expired waits {int(out[0])}, first: flag index {fi} = {what}; epoch {int(out[2])}, workgroup {int(out[3])}; "
              f"chain kernels whose release word never came: {int(out[5])}")
        break
    del model
else:
    print("no wait expired in", reps, "runs")
