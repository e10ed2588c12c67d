"""Development aid: C2-shaped training WITH the hyper-parameter / inducing-point step every iteration (the reference's default),
for rocprofv3 --kernel-trace --stats."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import agp_amd as AGP
m, B, D, N = 1024, 1024, 32, 100000
rng = np.random.default_rng(0)
X = rng.random((N, D)); w = rng.standard_normal(D)
y = np.sign(np.sin(X @ w) + 0.1 * rng.standard_normal(N))
Z = X[rng.permutation(N)[:m]].copy()
model = AGP.SVGP(AGP.with_lengthscale(AGP.SqExponentialKernel(), np.sqrt(D) / 4), AGP.LogisticLikelihood(), AGP.AnalyticSVI(B), Z,
                 optimiser=AGP.ADAM(0.01), Zoptimiser=AGP.ADAM(0.001))
it = 60
idx = [rng.choice(N, B, replace=False) for _ in range(it)]
AGP.train_(model, X, y, it, idx_stream=idx)
torch.cuda.synchronize()
