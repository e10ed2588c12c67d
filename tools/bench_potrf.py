"""Times the Cholesky building block (agp_potrf_jitter) alone on large matrices: TFLOP/s = n^3/3 / time.
Settings come from the environment (AGP_CHOL_GROUP, AGP_CHOL_LOOKAHEAD); run through gpurun."""
import ctypes as C, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as g
g.build()
from agp_amd import capi
L = capi.lib()
ctx = C.c_void_p()
assert L.agp_ctx_create(0, C.c_void_p(torch.cuda.current_stream().cuda_stream), C.byref(ctx)) == 0
for dt, tdt in ((0, torch.float64), (1, torch.float32)):
    for n in (2048, 4096, 8192, 16384):
        G = torch.randn(n, n + 64, dtype=tdt, device="cuda")
        A = G @ G.T / n + 0.5 * torch.eye(n, dtype=tdt, device="cuda")
        del G
        info = C.c_int32(-1)
        ts = []
        for it in range(4):
            a = A.clone()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            st = L.agp_potrf_jitter(ctx, dt, a.data_ptr(), n, n, 1e-4, C.byref(info))  # synchronises (info comes back)
            t1 = time.perf_counter()
            assert st == 0 and info.value == 0
            ts.append(t1 - t0)
        t = min(ts[1:])
        # residual on a slice
        Lf = torch.tril(a)
        r = (Lf[:512] @ Lf[:512].T - (A[:512, :512] + 1e-4 * torch.eye(512, dtype=tdt, device="cuda"))).abs().max().item()
        print(f"{'f64' if dt == 0 else 'f32'} n={n:6d}  {t*1e3:8.3f} ms  {n**3/3/t/1e12:6.2f} TFLOP/s  resid {r:.1e}", flush=True)
        del A, a, Lf
