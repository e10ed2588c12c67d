import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as g
g.build()
from agp_amd import capi
L = capi.lib()
f = L.agp_dev_diag_bench
f.restype = C.c_int32
f.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_double)]
ctx = C.c_void_p()
assert L.agp_ctx_create(0, C.c_void_p(torch.cuda.current_stream().cuda_stream), C.byref(ctx)) == 0
names = {0: "1-level", 1: "2-level", 2: "2-level+minors", 3: "[expt] no pivot LDL", 4: "[expt] no transforms/updates", 5: "[expt] barriers only", 6: "[expt] publish+barrier", 7: "[expt] harness only", 8: "panel-16", 9: "[expt] panel: no inverse", 10: "[expt] panel: no elimination, no inverse", 11: "[expt] panel: up to substitutions", 12: "[expt] panel: no final scaling"}
for dt, dn in ((0, "f64"), (1, "f32")):
    for blocks in (1, 16):
        for v in ((0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12) if dt == 0 else (0, 1, 2, 8)):
            us = C.c_double()
            rc = f(ctx, dt, v, blocks, 50, C.byref(us))
            if rc and (v < 3 or v == 8): print("FAILED residual check", dn, names[v])
            print(f"{dn} blocks={blocks:3d} {names[v]:42s} {us.value:8.2f} us/tile  ({us.value*2.1e3/64:6.0f} cyc/col @2.1GHz)")
