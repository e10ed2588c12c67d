"""Development aid: the batch-parallel step on ONE GPU with a one-rank callback communicator whose "all-reduce" is an asynchronous
sleep kernel on the stream it is handed (AGP_FORCE_SPLIT=1): with AGP_SPLIT_OVERLAP=1 the tile workgroups of the next task-graph
launch really wait at their arrival gates.  Prints a hash of the final state and the step time; the hash must not depend on the flag.
usage: AGP_FORCE_SPLIT=1 [AGP_SPLIT_OVERLAP=1] python tools/dbg_overlap.py [us_total] [iters] [m] [B]"""
import hashlib
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g  # noqa: E402

g.build()
import torch  # noqa: E402

import agp_amd as AGP  # noqa: E402
from agp_amd import capi  # noqa: E402
from agp_amd import parallel as P  # noqa: E402

us_total = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 200
m = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
B = int(sys.argv[4]) if len(sys.argv) > 4 else 1024
rng = np.random.default_rng(11)
N, D = 20000, 8
X = rng.random((N, D))
f = np.sin(4 * X[:, 0]) + X[:, 1] - 0.8
y = np.sign(f + 0.3 * rng.standard_normal(N))
y[y == 0] = 1.0
Z = X[rng.permutation(N)[:m]].copy()
idx = [rng.choice(N, B, replace=False) for _ in range(iters)]
mdl = AGP.SVGP(AGP.SqExponentialKernel() @ AGP.ScaleTransform(3.0), AGP.LogisticLikelihood(), AGP.AnalyticSVI(B), Z, optimiser=False)
mp_ = (m + 63) // 64 * 64
full = mp_ + (mp_ // 64) * (mp_ // 64 + 1) // 2 * 4096
calls = []


def fake(ptr, count, dtype, stream):
    us = us_total if count >= full else max(15.0, us_total * count / full)
    calls.append(count)
    with P.on_stream(stream):
        torch.cuda._sleep(int(us * 1700))


comm = P.Comm.from_callback(mdl, 0, 1, fake)
eng = P.HipEngine(mdl, B).bind_data(X, y)
nxt = eng.prefetch(idx[0])
torch.cuda.synchronize()
t0 = time.perf_counter()
for it in range(iters):
    cur = nxt
    eng.step_multi(cur, N / B, capi.SHARD_BATCH, comm)
    if it + 1 < iters:
        nxt = eng.prefetch(idx[it + 1])
eng.check()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
st = mdl.get_state(0)
h = hashlib.sha256(b"".join(np.ascontiguousarray(a).tobytes() for a in st)).hexdigest()[:16]
print(f"overlap={os.environ.get('AGP_SPLIT_OVERLAP', '0')} calls/step={len(calls) / iters:.1f} ms/step={dt * 1e3 / iters:.4f} "
      f"prologue steps={eng.step_counters()[1]} state={h}")
