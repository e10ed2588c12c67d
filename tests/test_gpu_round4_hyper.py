"""GPU tests of the round-4 rework of the hyper-parameter iteration (fewer products per iteration).

What changed on the device (agp_hyper.h, hypergrad() in agp_capi.hip):
* H = G_kappa K^-1 from ONE product, kappa (Sigma K^-1) (was kappa Sigma, then (.) K^-1);
* G_K from ONE product, C (Sigma K^-1) with C = kappa' diag(w) kappa + K^-1 / 4 left behind by the prologue of the factorisation
  launch (was kappa' H and K^-1 Sigma K^-1) -- only taken from 2 block columns on (the prologue needs a task graph), which the small
  fixtures of tests/test_gpu_parity.py never reach;
* mu = Sigma eta1 (was Xa' (Xa eta1)); the kernel backward pass in product form; K^-1 = X'X / Sigma = Xa'Xa as a balanced product
  (k_xtx_bal, from 8 block rows on).
All of it against the NumPy oracle's training loop (update_hyperparameters!, autotuning.jl:86-140) at sizes where the new paths run,
with the counters of agp_svgp_hyper_counters saying that they did; plus the pre-round-4 paths (environment switches) on the same
trajectory, and the balanced product against numpy.linalg through agp_spd_inverse.
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

import _knobs as KN

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


@pytest.fixture(scope="module")
def mods(built):
    import torch

    assert torch.cuda.is_available()
    import agp_amd as AGP
    from agp_amd import capi

    from oracle import agp_ref as R

    return AGP, R, capi, torch


def _problem(rng, N, D, m):
    X = rng.random((N, D))
    w = rng.standard_normal(D)
    f = np.sin(3.0 * X @ w / np.sqrt(D)) + X[:, 0] - 0.5
    Z = X[rng.permutation(N)[:m]].copy()
    return X, f, Z


@pytest.mark.parametrize("likname,ard,m,B", [("logistic", False, 512, 512), ("gaussian", True, 576, 448),
                                             ("studentt", False, 1024, 1024)])
def test_hyper_iteration_with_fused_products_matches_oracle(mods, likname, ard, m, B):
    """train! with optimiser and Zoptimiser at 8, 9 and 16 block columns: kernel parameters, Z, eta and mu after eight iterations
    against the oracle (<= 1e-8 / 1e-7), and every hyper-gradient formed G_K from one product."""
    AGP, R, capi, torch = mods
    rng = np.random.default_rng(1000 + m)
    N, D, iters = 4000, 4, 8  # (the reference takes hyper steps from its third iteration on and not after the last one)
    X, f, Z = _problem(rng, N, D, m)
    la, lr, y = {
        "gaussian": (AGP.GaussianLikelihood(0.1), R.GaussianLikelihood(0.1), f + 0.2 * rng.standard_normal(N)),
        "logistic": (AGP.LogisticLikelihood(), R.LogisticLikelihood(), (f + 0.3 * rng.standard_normal(N) > 0).astype(int)),
        "studentt": (AGP.StudentTLikelihood(4.0), R.StudentTLikelihood(4.0), f + 0.2 * rng.standard_t(4.0, N)),
    }[likname]
    sc = np.array([2.0, 1.5, 2.5, 1.0]) if ard else 2.0
    ka = 1.3 * (AGP.SqExponentialKernel() @ (AGP.ARDTransform(sc) if ard else AGP.ScaleTransform(sc)))
    idx = [rng.choice(N, B, replace=False) for _ in range(iters)]
    ma = AGP.SVGP(ka, la, AGP.AnalyticSVI(B), Z, optimiser=AGP.ADAM(0.02), Zoptimiser=AGP.ADAM(0.002))
    AGP.train_(ma, X, y, iters, idx_stream=idx)
    mr = R.SVGP(R.Kernel("sqexponential", sc, 1.3), lr, Z, stochastic=True, batchsize=B, k_opt=R.Adam(0.02), z_opt=R.Adam(0.002),
                ard=ard)
    mr.train(X, y, iters, idx_stream=idx)
    g = mr.latents[0]
    assert abs(ma.kernels[0].variance - 1.3) > 1e-3 and np.max(np.abs(ma.Zs[0] - Z)) > 1e-4  # the hypers and Z really moved
    assert ma.kernels[0].variance == pytest.approx(g.kernel.sigma2, rel=1e-8)
    assert _rel(ma.kernels[0].scales(D), np.broadcast_to(g.kernel.scale, (D,))) < 1e-8
    assert _rel(ma.Zs[0], g.Z) < 1e-8
    mu, Sig, e1, e2 = ma.get_state(0)
    assert _rel(e2, g.eta2) < 1e-7 and _rel(e1, g.eta1) < 1e-7
    assert _rel(mu, g.mu) < 1e-7 and _rel(Sig, g.Sigma) < 1e-7
    ng, nf = C.c_int64(), C.c_int64()
    ma._chk(capi.lib().agp_svgp_hyper_counters(ma._ensure_handle(B), C.byref(ng), C.byref(nf)))
    assert ng.value >= iters - 4, (ng.value, nf.value)
    if not (KN.no_prologue() or KN.forced("AGP_HYPER_GK_FUSED")):
        assert nf.value == ng.value, (ng.value, nf.value)


_AB_SCRIPT = r"""
import sys, numpy as np
sys.path.insert(0, {root!r})
import agp_amd as AGP
rng = np.random.default_rng(5)
N, D, m, B, iters = 3000, 3, 512, 512, 7
X = rng.random((N, D)); f = np.sin(4 * X[:, 0]) + X[:, 1] - 0.8 * X[:, 2]
y = (f + 0.3 * rng.standard_normal(N) > 0.2).astype(int)
Z = X[rng.permutation(N)[:m]].copy()
idx = [rng.choice(N, B, replace=False) for _ in range(iters)]
k = 1.2 * (AGP.SqExponentialKernel() @ AGP.ARDTransform(np.array([2.0, 3.0, 1.5])))
ma = AGP.SVGP(k, AGP.LogisticLikelihood(), AGP.AnalyticSVI(B), Z, optimiser=AGP.ADAM(0.02), Zoptimiser=AGP.ADAM(0.002))
AGP.train_(ma, X, y, iters, idx_stream=idx)
mu, Sig, e1, e2 = ma.get_state(0)
np.savez(sys.argv[1], var=ma.kernels[0].variance, sc=ma.kernels[0].scales(D), Z=ma.Zs[0], mu=mu, e2=e2, pred=AGP.predict_f(ma, X[:200], cov=True)[1])
"""


def test_pre_round4_hyper_paths_stay_on_the_same_trajectory(mods, tmp_path):
    """AGP_HYPER_GK_FUSED=0 forces the gradient form that handles without a prologue launch take (several latents, batch-sharded,
    online, stale-K: kappa' H and Apred instead of the one product C (Sigma K^-1)).  Fresh processes (the switch is read once), same
    data: the trajectories agree to rounding -- also when every task-graph launch is aborted behind its prologue and redone by the
    fallback (AGP_DAG_TEST_ABORT=1).  (The round-4 A/B switches AGP_HYPER_TWO_PRODUCTS / AGP_XTX_BALANCED / AGP_HYPER_SIDE are
    gone: the two-product form is what the heteroscedastic model runs, the one-workgroup-per-tile X'X what matrices below 8 block
    rows run -- both covered by their own tests.)"""
    script = tmp_path / "ab.py"
    script.write_text(_AB_SCRIPT.format(root=ROOT))
    outs = {}
    for name, env in [("new", {}), ("old", {"AGP_HYPER_GK_FUSED": "0"}), ("abort", {"AGP_DAG_TEST_ABORT": "1"})]:
        out = tmp_path / f"{name}.npz"
        r = subprocess.run([sys.executable, str(script), str(out)], env={**os.environ, **env}, capture_output=True, text=True,
                           timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs[name] = np.load(out)
    # "abort": every task-graph launch of the run is treated as having lost a dependency AFTER its prologue (which has taken the
    # natural-gradient step and stored C = kappa' diag(w) kappa + K^-1 / 4 by then) and is redone by the in-stream fallback, another
    # factorisation algorithm: the fused G_K must still find a complete C
    for other in ("old", "abort"):
        for key in ("var", "sc", "Z", "mu", "e2"):
            # (3e-9: the two gradient forms differ by rounding, and after 30 optimiser steps Z sits 1.2e-9 apart when the kernel
            #  matrices come from the direct-difference VALU kernel -- AGP_KERNELMATRIX_VALU=1 --; with the default MFMA kernel the old limit of 1e-9 holds)
            assert _rel(outs[other][key], outs["new"][key]) < (3e-9 if other != "abort" else 1e-8), (other, key)
        # predictive variances k** - k*' (K^-1 - K^-1 Sigma K^-1) k* cancel against K^-1 of a kernel matrix with jitter 1e-8: a
        # different summation order inside X'X shows at 1e-6 of the variance (measured 9.6e-7)
        assert _rel(outs[other]["pred"], outs["new"]["pred"]) < 1e-5, other
    assert abs(float(outs["new"]["var"]) - 1.2) > 1e-3


@pytest.mark.parametrize("n,dtype", [(512, 0), (1000, 0), (2048, 0), (1024, 1), (1500, 1)])
def test_balanced_xtx_through_spd_inverse(mods, n, dtype):
    """agp_spd_inverse = task-graph factorisation with L^-1, then A^-1 = X'X as the balanced product (k_xtx_bal: units of at most
    ch k-blocks, partial tiles added in unit order by a second launch) at 8 .. 32 block rows, ragged sizes included, fp64 and
    fp32, against numpy -- and twice, bit for bit."""
    AGP, R, capi, torch = mods
    L = capi.lib()
    ctx = C.c_void_p()
    assert L.agp_ctx_create(0, C.c_void_p(torch.cuda.current_stream().cuda_stream), C.byref(ctx)) == 0
    try:
        rng = np.random.default_rng(n)
        G = rng.standard_normal((n, n + 5))
        A = G @ G.T / n + 0.3 * np.eye(n)
        tdt = torch.float64 if dtype == 0 else torch.float32
        res = []
        for _ in range(2):
            ad = torch.tensor(A, dtype=tdt, device="cuda")
            inv = torch.empty(n, n, dtype=tdt, device="cuda")
            ld, info = C.c_double(), C.c_int32(-1)
            st = L.agp_spd_inverse(ctx, dtype, ad.data_ptr(), n, n, inv.data_ptr(), n, C.byref(ld), C.byref(info))
            assert st == 0 and info.value == 0
            res.append(inv.cpu().numpy().astype(np.float64))
        ref = np.linalg.inv(A)
        assert _rel(res[0], ref) < (1e-10 if dtype == 0 else 2e-3)
        assert np.array_equal(res[0], res[1])
        assert np.array_equal(res[0], res[0].T)  # mirrored tiles
        assert abs(ld.value - np.linalg.slogdet(A)[1]) < (1e-9 if dtype == 0 else 1e-3) * max(1.0, abs(ld.value))
    finally:
        L.agp_ctx_destroy(ctx)
