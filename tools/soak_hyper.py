"""Determinism soak of the hyper-on iteration (the reference's default mode): the C2 shape with the hyper-parameter / inducing-point
step every iteration, run twice -- eta2, Z and the kernel parameters must agree bit for bit (balanced X'X in unit order, the
reductions of the gradient in fixed order, no atomics).   usage: python tools/soak_hyper.py [iterations]"""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import agp_amd as AGP

it = int(sys.argv[1]) if len(sys.argv) > 1 else 400
m, B, D, N = 1024, 1024, 32, 100000
res = []
for run in range(2):
    rng = np.random.default_rng(0)
    X = rng.random((N, D)); w = rng.standard_normal(D)
    y = np.sign(np.sin(X @ w) + 0.1 * rng.standard_normal(N))
    Z = X[rng.permutation(N)[:m]].copy()
    idx = [rng.choice(N, B, replace=False) for _ in range(it)]
    model = AGP.SVGP(AGP.with_lengthscale(AGP.SqExponentialKernel(), np.sqrt(D) / 4), AGP.LogisticLikelihood(), AGP.AnalyticSVI(B), Z,
                     optimiser=AGP.ADAM(0.01), Zoptimiser=AGP.ADAM(0.001))
    AGP.train_(model, X, y, it, idx_stream=idx)
    mu, Sig, e1, e2 = model.get_state(0)
    h = hashlib.sha1(np.ascontiguousarray(e2).tobytes() + np.ascontiguousarray(model.Zs[0]).tobytes()
                     + np.asarray(model.kernels[0].scales(D)).tobytes()).hexdigest()[:16]
    print(f"run {run}: {it} iterations, |eta1| = {np.linalg.norm(e1):.6e}, scale[0] = {model.kernels[0].scales(D)[0]:.12f}, finite = "
          f"{bool(np.isfinite(e2).all())}, sha1(eta2, Z, scales) = {h}")
    res.append((e2, model.Zs[0].copy()))
print("bitwise identical:", bool(np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])))
