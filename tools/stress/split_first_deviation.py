"""Development aid behind docs/DESIGN_LOG.md section 14 (defect 5): the first step at which a run of stress B leaves the first run's
trajectory, and which quantity (eta, r, w, theta) differs there.  Needs a library built with -DAGP_DEBUG_PTRS (agp_debug_ptr):

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -DAGP_DEBUG_PTRS -o /tmp/libagp_dbg.so augmentedgaussianprocesses.jl_amd/csrc/agp_capi.hip
    AGP_HIP_LIB=/tmp/libagp_dbg.so AGP_CHAIN_SPLIT=1 python tools/stress/split_first_deviation.py 400"""
import sys, ctypes as C, hashlib, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import __graft_entry__ as g
import agp_amd as AGP
from agp_amd import capi
reps = int(sys.argv[1])
m, B, D, N, steps = 1024, 2048, 16, 50000, 200
rng = np.random.default_rng(0)
X = rng.random((N, D)); y = np.sin(X @ rng.standard_normal(D)) + 0.1 * rng.standard_normal(N)
Z = X[rng.permutation(N)[:m]].copy()
idx = np.stack([rng.choice(N, B, replace=False) for _ in range(32)])
L = capi.lib()
dbg = C.CDLL(capi.LIB_PATH)
dbg.agp_debug_ptr.restype = C.c_void_p
dbg.agp_debug_ptr.argtypes = [C.c_void_p, C.c_int]
hip = C.CDLL("libamdhip64.so")
hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
names = {0: ("eta2", m * m), 1: ("eta1", m), 2: ("w_a", B), 3: ("r_a", B), 4: ("w_b", B), 5: ("r_b", B), 8: ("theta", B)}
def run():
    model = AGP.SVGP(AGP.with_lengthscale(AGP.Matern52Kernel(), 1.0), AGP.StudentTLikelihood(3.0), AGP.AnalyticSVI(B), Z, optimiser=False, T=np.float32)
    AGP.train_(model, X, y, 1, idx_stream=idx[:1])
    h = model._h
    Xd, yd, _ = model._data
    ia = torch.as_tensor(idx, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    hist = {k: torch.zeros((steps, n), dtype=torch.float32, device="cuda") for k, (nm, n) in names.items()}
    ptr = {k: dbg.agp_debug_ptr(h, k) for k in names}
    for i in range(steps):
        j = i % 32
        assert L.agp_svgp_cavi_step(h, C.c_void_p(Xd.data_ptr()), Xd.stride(0), C.c_void_p(yd.data_ptr()), C.c_void_p(ia[j].data_ptr()), B, N / B) == 0
        L.agp_svgp_prefetch(h, C.c_void_p(Xd.data_ptr()), Xd.stride(0), C.c_void_p(ia[(j + 1) % 32].data_ptr()), B)
        for k, (nm, n) in names.items():
            assert hip.hipMemcpyAsync(C.c_void_p(hist[k][i].data_ptr()), C.c_void_p(ptr[k]), n * 4, 3, C.c_void_p(st)) == 0
    torch.cuda.synchronize()
    e2 = model.get_state(0)[3]
    return hashlib.sha256(np.ascontiguousarray(e2).tobytes()).hexdigest()[:10], hist
ref_h, ref = run()
print("reference", ref_h)
found = 0
for rep in range(reps):
    hh, hist = run()
    if hh == ref_h:
        continue
    found += 1
    print("rep", rep, "hash", hh)
    first = {}
    for k, (nm, n) in names.items():
        neq = (hist[k] != ref[k]).any(dim=1).cpu().numpy()
        first[nm] = int(np.argmax(neq)) if neq.any() else -1
    print("  first differing snapshot per quantity:", first)
    s = min(v for v in first.values() if v >= 0)
    for k, (nm, n) in names.items():
        a, b = hist[k][s].double(), ref[k][s].double()
        d = (a - b).abs()
        if float(d.max()) > 0:
            nz = int((d > 0).sum())
            print(f"  snapshot {s} {nm}: {nz} elements differ, max abs {float(d.max()):.3e} (max |ref| {float(b.abs().max()):.3e})")
            if nm == "eta2":
                dd = d.reshape(m, m).cpu().numpy()
                t = dd.reshape(m // 64, 64, m // 64, 64).max(axis=(1, 3))
                tiles = [(int(i), int(j), float(t[i, j])) for i, j in zip(*np.nonzero(t))]
                print("   tiles with differences (row, col, max):", tiles[:20], "n =", len(tiles))
                ii, jj = np.nonzero(dd)
                print("   elements:", [(int(a_), int(b_), float(hist[k][s][a_ * m + b_]), float(ref[k][s][a_ * m + b_])) for a_, b_ in list(zip(ii, jj))[:8]])
            else:
                ii = torch.nonzero(d > 0).flatten()[:8].cpu().numpy()
                print("   elements:", [(int(q), float(a[q]), float(b[q])) for q in ii])
    if found >= 3:
        break
print("deviating runs:", found, "of", rep + 1)
