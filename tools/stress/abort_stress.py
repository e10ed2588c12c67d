"""Stress A of docs/DESIGN_LOG.md section 14: N models of 6 CAVI steps (fp64, m = B = 1024) in one process, meant to be run with the test
hook and a forced launch form, e.g.

    AGP_CHAIN_SPLIT=1 AGP_DAG_TEST_ABORT=1 timeout 100 python tools/stress/abort_stress.py 30

Every model must land on the first model's natural parameters (the fallback is deterministic); prints one line per model with its
wall time (a stall shows as a model that never prints) and the count of models that failed or deviated.  Keep N small: a process
that creates many contexts slows down."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import __graft_entry__ as g  # noqa: E402

g.build()
import agp_amd as AGP  # noqa: E402

rng = np.random.default_rng(6)
N, D, m, B, iters = 6000, 6, 1024, 1024, 6
X = rng.random((N, D))
f = np.sin(4 * X[:, 0]) + X[:, 1] - 0.8
y = np.sign(f + 0.3 * rng.standard_normal(N))
Z = X[rng.permutation(N)[:m]].copy()
idx = [rng.choice(N, B, replace=False) for _ in range(iters)]
ref, bad = None, 0
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 30):
    t0 = time.time()
    try:
        ma = AGP.SVGP(AGP.SqExponentialKernel() @ AGP.ScaleTransform(3.0), AGP.LogisticLikelihood(), AGP.AnalyticSVI(B), Z, optimiser=False)
        if os.environ.get("AGP_STRESS_NO_LOOKAHEAD") == "1":
            # the same six steps through the raw ABI WITHOUT agp_svgp_prefetch: no look-ahead stream, kappa computed in line
            import ctypes as C

            import torch

            from agp_amd import capi

            L = capi.lib()
            AGP.train_(ma, X, y, 1, idx_stream=idx[:1])
            Xd, yd, _ = ma._data
            ia = torch.as_tensor(np.stack(idx), device="cuda")
            for i in range(1, iters):
                assert L.agp_svgp_cavi_step(ma._h, C.c_void_p(Xd.data_ptr()), Xd.stride(0), C.c_void_p(yd.data_ptr()),
                                            C.c_void_p(ia[i].data_ptr()), B, N / B) == 0
            ma._chk(L.agp_svgp_check_status(ma._h))
        else:
            AGP.train_(ma, X, y, iters, idx_stream=idx)
        e2 = ma.get_state(0)[3]
        if ref is None:
            ref = e2
        d = float(np.max(np.abs(e2 - ref)) / np.max(np.abs(ref)))
        note = "" if d <= 1e-9 else f"  DEVIATES {d:.2e}"
        bad += d > 1e-9
    except Exception as e:  # noqa: BLE001
        bad += 1
        note = "  FAILED " + str(e)[:100]
    extra = ""
    if os.environ.get("AGP_STRESS_DIAG") == "1":  # a -DAGP_DEBUG_PTRS build: which bounded wait of the task graph ran out, if any
        import ctypes as C

        from agp_amd import capi

        dbg = C.CDLL(capi.LIB_PATH)
        dbg.agp_debug_dag_diag.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
        out = (C.c_ulonglong * 8)()
        dbg.agp_debug_dag_diag(ma._ctx, out)
        extra = f"  [diag: expired waits {int(out[0])}, kind {int(out[4])}, workgroup {int(out[3])}, chain release never came {int(out[5])}]"
    print(f"model {rep}: {time.time() - t0:.2f} s{note}{extra}", flush=True)
print("bad", bad)
