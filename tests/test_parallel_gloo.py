"""world_size-2 `gloo` tests (CPU) of the multi-GPU drivers in augmentedgaussianprocesses.jl_amd/parallel.py.

The drivers are backend-agnostic; here they drive an oracle-backed engine (tests may use the oracle) so that the sharding
plan and the collectives are checked against the single-process oracle: latent-parallel LogisticSoftMax (B-vector
all-reduce inside the fixed point) and batch-parallel logistic (statistics all-reduce)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import agp_ref as R


class OracleEngine:
    """The phase-split step of include/agp_hip.h restated with the oracle, for a latent slice [lo, hi)."""

    def __init__(self, kern, lik, Z, X, y_treated, lo, hi, stochastic=True, batchsize=0):
        self.M = R.SVGP(kern, lik, Z, stochastic=stochastic, batchsize=batchsize)
        self.M.latents = self.M.latents[lo:hi]
        self.lo, self.hi = lo, hi
        self.lik = lik
        self.X, self.y = X, y_treated
        self.is_lsm = lik.name == "logisticsoftmax"
        self.alpha = None
        self.gsum = None
        self.stats = None
        self.n = 1

    def step_local(self, idx, rho):
        self.rho = rho
        self.idx = np.asarray(idx)
        xb = self.X[self.idx]
        self.yb = self.y[self.idx]
        self.M.hp_updated = self.M.hp_updated
        self.M.compute_kernel_matrices(xb)
        self.muf = [R.mean_f(g.mu, g.kappa) for g in self.M.latents]
        self.varf = [R.var_f(g.Sigma, g.kappa, g.Kt) for g in self.M.latents]
        B = len(self.idx)
        if self.is_lsm:
            K = self.lik.n_class
            if self.alpha is None:
                self.alpha = K * np.ones(B)
            self.beta = K * np.ones(B)
            self.c = [R.sqrt_expec_square(m, v) for m, v in zip(self.muf, self.varf)]
            self.gsum = torch.zeros(B, dtype=torch.float64)

    def lsm_gamma(self):
        from scipy.special import digamma

        psi = digamma(self.alpha)
        self.gamma = [np.exp(psi) * R.safe_expcosh(-m / 2, c / 2) / (2 * self.beta) for m, c in zip(self.muf, self.c)]
        self.gsum.copy_(torch.from_numpy(sum(self.gamma) if self.gamma else np.zeros_like(self.alpha)))

    def lsm_alpha(self):
        self.alpha = 1.0 + self.gsum.numpy().copy()

    def step_stats(self):
        nl = len(self.M.latents)
        if self.is_lsm:
            Y = self.yb[:, self.lo:self.hi].astype(np.float64)
            theta = [(Y[:, k] + self.gamma[k]) * R.theta_pg(self.c[k]) for k in range(nl)]
            g1 = [(Y[:, k] - self.gamma[k]) / 2 for k in range(nl)]
            g2 = [t / 2 for t in theta]
        else:
            lv = R.local_updates(R.init_local_vars(self.lik, len(self.idx)), self.lik, self.yb, tuple(self.muf),
                                 tuple(self.varf))
            g1 = list(R.grad_E_mu(self.lik, self.yb, lv))
            g2 = list(R.grad_E_Sigma(self.lik, self.yb, lv))
        m = len(self.M.latents[0].Z)
        out = np.zeros(nl * (m + m * m))
        for k, g in enumerate(self.M.latents):
            t = g.kappa.T @ (self.rho * g1[k])
            S = R.rho_kappa_diag_theta_kappa(self.rho, g.kappa, g2[k])
            out[k * (m + m * m):k * (m + m * m) + m] = t
            out[k * (m + m * m) + m:(k + 1) * (m + m * m)] = S.ravel()
        self.stats = torch.from_numpy(out)

    def step_global(self):
        m = len(self.M.latents[0].Z)
        lr = R.robbins_monro_lr(self.n)
        st = self.stats.numpy()
        for k, g in enumerate(self.M.latents):
            t = st[k * (m + m * m):k * (m + m * m) + m]
            S = st[k * (m + m * m) + m:(k + 1) * (m + m * m)].reshape(m, m)
            d1 = t - g.eta1  # ZeroMean
            d2 = -(S + g.Kinv / 2) - g.eta2
            g.eta1 = g.eta1 + lr * d1
            g.eta2 = g.eta2 + lr * d2
            g.eta2 = (g.eta2 + g.eta2.T) / 2
            g.mu, g.Sigma = R.natural_to_standard(g.eta1, g.eta2)
        self.n += 1


class TiedOracleEngine(OracleEngine):
    """OracleEngine + the tied-Z hyper step (gradients through the oracle's hand backward, ADAM like agp_svgp_hyper_apply)."""

    def step_stats(self):
        super().step_stats()
        nl = len(self.M.latents)
        Y = self.yb[:, self.lo:self.hi].astype(np.float64)
        self._theta = [(Y[:, k] + self.gamma[k]) * R.theta_pg(self.c[k]) for k in range(nl)]
        self._g1 = [(Y[:, k] - self.gamma[k]) / 2 for k in range(nl)]

    def hyper_gradients(self):
        xb = self.X[self.idx]
        tot = None
        for k, g in enumerate(self.M.latents):
            muf = R.mean_f(g.mu, g.kappa)
            gr = R.hyper_gradient_core(g, xb, self._g1[k] - self._theta[k] * muf, -self._theta[k] / 2.0, self.rho, self.M.jitter)
            v = np.concatenate([[gr["dvariance"]], gr["dscale"], gr["dZ"].ravel()])
            tot = v if tot is None else tot + v
        return torch.from_numpy(tot)

    def hyper_apply(self, gt):
        g = gt.numpy()
        D = self.X.shape[1]
        for lat in self.M.latents:
            if not hasattr(lat, "_adam"):
                lat._adam = [R.Adam(0.01).init(np.zeros(1)), R.Adam(0.01).init(np.zeros(1)), R.Adam(0.001).init(np.zeros_like(lat.Z))]
            ak, az = R.Adam(0.01), R.Adam(0.001)
            v = np.array([lat.kernel.sigma2])
            lat._adam[0], dv = ak.apply(lat._adam[0], v * g[:1])
            lat.kernel.sigma2 = float(np.exp(np.log(v) + dv)[0])
            s0 = np.array([float(lat.kernel.scale)])
            lat._adam[1], ds = ak.apply(lat._adam[1], s0 * np.array([np.sum(g[1:1 + D])]))
            lat.kernel.scale = float(np.exp(np.log(s0) + ds)[0])
            lat._adam[2], dz = az.apply(lat._adam[2], g[1 + D:].reshape(lat.Z.shape))
            lat.Z = lat.Z + dz
        self.M.hp_updated = True


def _tied_worker(rank, world, port, q):
    import sys

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import agp_amd  # noqa: F401
    from agp_amd import parallel as P

    if world > 1:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
    X, y, lik, Z, idx, N, B, iters = _data("logisticsoftmax")
    yt = R.treat_labels(y, lik)
    lo, hi = P.latent_slice(lik.n_latent, world, rank)
    eng = TiedOracleEngine(R.Kernel("sqexponential", 3.0, 1.0), lik, Z, X, yt, lo, hi, batchsize=B)
    for it in range(iters):
        P.latent_parallel_step(eng, idx[it], N / B)
        P.tied_hyper_step(eng)
    q.put((rank, lo, hi, [g.Z for g in eng.M.latents], [g.kernel.sigma2 for g in eng.M.latents],
           [g.kernel.scale for g in eng.M.latents], [g.eta1 for g in eng.M.latents]))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def test_tied_z_hyper_step_all_reduce_two_ranks():
    """tied-Z mode: the Z / kernel hyper-gradient summed over latents and all-reduced over ranks keeps one shared kernel and
    one shared Z on every latent of every rank; the 2-rank run equals the single-process run of the same driver."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_tied_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    q1 = ctx.Queue()
    p1 = ctx.Process(target=_tied_worker, args=(0, 1, port, q1))
    p1.start()
    ref = q1.get(timeout=180)
    p1.join(timeout=60)
    _, _, _, Zr, vr, sr, e1r = ref
    for rank, lo, hi, Zs, vs, ss, e1s in res:
        for k in range(hi - lo):
            assert np.allclose(Zs[k], Zr[lo + k], rtol=1e-9, atol=1e-12)
            assert vs[k] == pytest.approx(vr[lo + k], rel=1e-9) and ss[k] == pytest.approx(sr[lo + k], rel=1e-9)
            assert np.allclose(e1s[k], e1r[lo + k], rtol=1e-8, atol=1e-10)
            assert np.allclose(Zs[k], Zs[0]) and np.allclose(Zs[0], res[0][3][0])  # one Z everywhere
    assert not np.allclose(Zr[0], _data("logisticsoftmax")[3])  # and it moved


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _data(likname):
    rng = np.random.default_rng(11)
    N, D, m, B, iters = 160, 2, 12, 40, 5
    X = rng.random((N, D))
    f = np.sin(5 * X[:, 0]) + X[:, 1]
    if likname == "logisticsoftmax":
        lik = R.LogisticSoftMaxLikelihood(4)
        y = 1 + np.digitize(f, np.quantile(f, [0.25, 0.5, 0.75]))
    else:
        lik = R.LogisticLikelihood()
        y = (f > f.mean()).astype(int)
    Z = X[rng.permutation(N)[:m]].copy()
    idx = [rng.choice(N, B, replace=False) for _ in range(iters)]
    return X, y, lik, Z, idx, N, B, iters


def _worker(rank, world, port, likname, mode, q):
    import sys

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import agp_amd  # noqa: F401  (the product package provides the drivers)
    from agp_amd import parallel as P

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    X, y, lik, Z, idx, N, B, iters = _data(likname)
    yt = R.treat_labels(y, lik)
    kern = R.Kernel("sqexponential", 3.0, 1.0)
    if mode == "latent":
        lo, hi = P.latent_slice(lik.n_latent, world, rank)
        eng = OracleEngine(kern, lik, Z, X, yt, lo, hi, batchsize=B)
        for it in range(iters):
            P.latent_parallel_step(eng, idx[it], N / B)
    else:
        lo, hi = 0, lik.n_latent
        eng = OracleEngine(kern, lik, Z, X, yt, lo, hi, batchsize=B // world)
        for it in range(iters):
            P.batch_parallel_step(eng, P.shard_batch(idx[it], world, rank), N / B)
    q.put((rank, lo, hi, [g.eta1 for g in eng.M.latents], [g.eta2 for g in eng.M.latents]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("likname,mode", [("logisticsoftmax", "latent"), ("logistic", "batch")])
def test_two_rank_drivers_match_single_process(likname, mode):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, likname, mode, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    X, y, lik, Z, idx, N, B, iters = _data(likname)
    ref = R.SVGP(R.Kernel("sqexponential", 3.0, 1.0), lik, Z, stochastic=True, batchsize=B)
    ref.train(X, y, iters, idx_stream=idx)
    for rank, lo, hi, e1s, e2s in res:
        for k, (e1, e2) in enumerate(zip(e1s, e2s)):
            g = ref.latents[lo + k]
            assert np.allclose(e1, g.eta1, rtol=1e-9, atol=1e-11)
            assert np.allclose(e2, g.eta2, rtol=1e-9, atol=1e-11)


def _pred_worker(rank, world, port, q):
    import torch.distributed as dist

    import agp_amd  # noqa: F401
    from agp_amd import parallel as P

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    Xt = np.arange(37 * 3, dtype=np.float64).reshape(37, 3)
    fn = lambda X: (X.sum(axis=1), (X ** 2).sum(axis=1))  # stands in for predict_f(model, X, cov=True)
    mu, var = P.predict_sharded(fn, Xt)
    lo, hi, loc = P.predict_sharded(fn, Xt, gather=False)
    q.put((rank, mu, var, lo, hi, len(loc[0])))
    dist.barrier()
    dist.destroy_process_group()


def test_row_sharded_prediction_two_ranks():
    """data-parallel predict over test rows: independent units, all-gather only to hand every rank the full result"""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_pred_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    Xt = np.arange(37 * 3, dtype=np.float64).reshape(37, 3)
    spans = []
    for rank, mu, var, lo, hi, nloc in res:
        assert np.array_equal(mu, Xt.sum(axis=1)) and np.array_equal(var, (Xt ** 2).sum(axis=1))
        assert nloc == hi - lo
        spans.append((lo, hi))
    assert sorted(spans) == [(0, 19), (19, 37)]


def test_sharding_helpers():
    import agp_amd  # noqa: F401
    from agp_amd import parallel as P

    assert [P.latent_slice(8, 8, r) for r in range(8)] == [(r, r + 1) for r in range(8)]
    assert [P.latent_slice(16, 8, r) for r in range(8)] == [(2 * r, 2 * r + 2) for r in range(8)]
    assert [P.latent_slice(3, 2, r) for r in range(2)] == [(0, 2), (2, 3)]
    idx = np.arange(12)
    assert np.array_equal(np.concatenate([P.shard_batch(idx, 4, r) for r in range(4)]), idx)
    with pytest.raises(ValueError):
        P.shard_batch(np.arange(10), 4, 0)
