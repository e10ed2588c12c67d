"""Five passes of the streaming predictor (means) over N = 1e6 points at the C2 shape -- the workload of `predict_roofline` in the
bench line, alone, for `rocprofv3 --kernel-trace --stats` (tools/collect_round6.sh predict -> profiles/r06_predict_kernel_stats.csv)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import __graft_entry__ as g

g.build()
import agp_amd as AGP
from agp_amd import capi

N, D, m, B = 1000000, 32, 1024, 1024
X = torch.rand(N, D, dtype=torch.float64, device="cuda")
y = torch.sign(torch.randn(N, dtype=torch.float64, device="cuda"))
Z = X[:m].cpu().numpy()
model = AGP.SVGP(AGP.with_lengthscale(AGP.SqExponentialKernel(), 1.4), AGP.LogisticLikelihood(), AGP.AnalyticSVI(B), Z, optimiser=False)
AGP.train_(model, X.cpu().numpy(), y.cpu().numpy(), 3)
L, h = capi.lib(), model._h
Xd, yd, _ = model._data
out = torch.empty(1, N, dtype=torch.float64, device="cuda")
for _ in range(5):
    model._chk(L.agp_svgp_predict_f(h, C.c_void_p(Xd.data_ptr()), Xd.stride(0), N, C.c_void_p(out.data_ptr()), None))
torch.cuda.synchronize()
