"""Stress B of docs/DESIGN_LOG.md section 14: N runs of 200 CAVI steps (fp32, m = 1024, B = 2048, look-ahead on, no synchronisation inside a
run) in one process; prints the SHA-256 prefixes of the final eta2 and, for runs that leave the first run's trajectory, the size of the
deviation.  e.g.  AGP_CHAIN_SPLIT=1 python tools/stress/split_hash_stress.py 300   (merged launch: AGP_CHAIN_SPLIT=0)"""
import sys, ctypes as C, hashlib, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import __graft_entry__ as g; g.build()
import agp_amd as AGP
from agp_amd import capi
reps = int(sys.argv[1])
m, B, D, N, steps = 1024, 2048, 16, 50000, 200
rng = np.random.default_rng(0)
X = rng.random((N, D)); y = np.sin(X @ rng.standard_normal(D)) + 0.1 * rng.standard_normal(N)
Z = X[rng.permutation(N)[:m]].copy()
idx = np.stack([rng.choice(N, B, replace=False) for _ in range(32)])
L = capi.lib()
from collections import Counter
cnt = Counter()
ref = None
diffs = []
fallbacks = 0
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
    torch.cuda.synchronize()
    nfb = C.c_int64(0)
    L.agp_ctx_task_graph_fallbacks(model._ctx, C.byref(nfb))
    fallbacks += nfb.value
    try:
        e2 = model.get_state(0)[3]
        cnt[hashlib.sha256(np.ascontiguousarray(e2).tobytes()).hexdigest()[:10]] += 1
        if ref is None: ref = e2.copy()
        d = np.abs(e2.astype(np.float64) - ref)
        if d.max() > 0:
            i, j = np.unravel_index(np.argmax(d), d.shape)
            diffs.append((rep, float(d.max() / np.abs(ref).max()), int((d > 0).sum()), int(i), int(j)))
    except Exception as e:
        cnt["EXC " + str(e)[:60]] += 1
    del model
print(os.environ.get("AGP_HIP_LIB", "default")[-12:], "AGP_CHAIN_SPLIT", os.environ.get("AGP_CHAIN_SPLIT"), "AGP_STEP_PROLOGUE",
      os.environ.get("AGP_STEP_PROLOGUE"), "runs", reps, "launches", reps * steps, "task_graph_fallbacks", fallbacks, dict(cnt), diffs[:12])
