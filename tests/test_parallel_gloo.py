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


class MOOracleEngine:
    """Latent-sharded multi-output model restated with the oracle: this rank owns latents [lo, hi) of Q; everything on the
    task side (mixing, local variables, update_A!) is replicated and only sees the exchanged buffer."""

    is_lsm = False
    is_mo_sharded = True
    reduce_device = "cpu"

    def __init__(self, kern, liks, Zs, A, X, ys, lo, hi, batchsize, a_opt):
        self.M = R.MOSVGP(kern, liks, Zs, A.copy(), stochastic=True, batchsize=batchsize, A_opt=a_opt)
        self.lo, self.hi = lo, hi
        self.own = self.M.latents[lo:hi]
        for q, g in enumerate(self.M.latents):  # poison what this rank does not own: it must never be read
            if not (lo <= q < hi):
                g.mu = np.full_like(g.mu, np.nan)
        self.X, self.ys = X, ys
        self.M.local_vars = [R.init_local_vars_single(l, batchsize) for l in liks]
        self.fbuf = None

    def _publish(self):
        Q, B = self.M.Q, len(self.idx)
        f = np.zeros((2, Q, B))
        for q in range(self.lo, self.hi):
            g = self.M.latents[q]
            f[0, q], f[1, q] = R.mean_f(g.mu, g.kappa), R.var_f(g.Sigma, g.kappa, g.Kt)
        self.fbuf = torch.from_numpy(f)

    def step_local(self, idx, rho):
        self.rho = self.M.rho = rho
        self.idx = np.asarray(idx)
        self.yb = [y[self.idx] for y in self.ys]
        helper = R.SVGP.__new__(R.SVGP)
        helper.latents, helper.jitter, helper.hp_updated, helper.stochastic = self.own, self.M.jitter, self.M.hp_updated, True
        R.SVGP.compute_kernel_matrices(helper, self.X[self.idx])
        self.M.hp_updated = False
        self._publish()

    def _mixed(self):
        f, A = self.fbuf.numpy(), self.M.A
        return ([A[t] @ f[0] for t in range(self.M.n_task)], [(A[t] ** 2) @ f[1] for t in range(self.M.n_task)])

    def mo_mix(self):
        M, f = self.M, self.fbuf.numpy()
        if M.A_opt is not None:  # update_A! on the exchanged values (replicated)
            for t, lik in enumerate(M.likelihoods):
                gmu = R.grad_E_mu(lik, self.yb[t], M.local_vars[t])[0]
                gS = R.grad_E_Sigma(lik, self.yb[t], M.local_vars[t])[0]
                mt = M.A[t] @ f[0]
                dA = np.zeros(M.Q)
                for q in range(M.Q):
                    others = mt - M.A[t, q] * f[0, q]
                    x1 = np.dot(gmu, f[0, q]) - 2.0 * np.dot(gS, f[0, q] * others)
                    x2 = np.dot(gS, f[0, q] ** 2 + f[1, q])
                    dA[q] = x1 - 2.0 * M.A[t, q] * x2
                M.A_state[t], delta = M.A_opt.apply(M.A_state[t], dA)
                M.A[t] = M.A[t] + delta
                M.A[t] = M.A[t] / np.sqrt(np.sum(M.A[t] ** 2))
        mu_t, var_t = self._mixed()
        for t, lik in enumerate(M.likelihoods):
            M.local_vars[t] = R.local_updates(M.local_vars[t], lik, self.yb[t], (mu_t[t],), (var_t[t],))
        gmu = [R.grad_E_mu(l, self.yb[t], M.local_vars[t])[0] for t, l in enumerate(M.likelihoods)]
        gS = [R.grad_E_Sigma(l, self.yb[t], M.local_vars[t])[0] for t, l in enumerate(M.likelihoods)]
        self.g1 = {q: sum(M.A[t, q] * (gmu[t] - 2.0 * gS[t] * (mu_t[t] - M.A[t, q] * f[0, q])) for t in range(M.n_task))
                   for q in range(self.lo, self.hi)}
        self.g2 = {q: sum(M.A[t, q] ** 2 * gS[t] for t in range(M.n_task)) for q in range(self.lo, self.hi)}

    def step_stats(self):
        pass

    def step_global(self):
        M = self.M
        for q in range(self.lo, self.hi):
            gp = M.latents[q]
            d1 = R.grad_eta1(self.g1[q], self.rho, gp.kappa, gp.L, gp.mu0, gp.eta1)
            d2 = R.grad_eta2(self.g2[q], self.rho, gp.kappa, gp.Kinv, gp.eta2)
            lr = R.robbins_monro_lr(gp.n_eta1, M.kappa_rm, M.tau_rm)
            gp.n_eta1 += 1
            gp.eta1 = gp.eta1 + lr * d1
            gp.eta2 = gp.eta2 + lr * d2
            gp.eta2 = (gp.eta2 + gp.eta2.T) / 2.0
            gp.mu, gp.Sigma = R.natural_to_standard(gp.eta1, gp.eta2)

    def mo_refresh_f(self):
        self._publish()

    def elbo_local(self):
        M = self.M
        mu_t, var_t = self._mixed()
        e = sum(R.expec_loglikelihood(l, self.yb[t], (mu_t[t],), (var_t[t],), M.local_vars[t], M.elbo_mode)
                for t, l in enumerate(M.likelihoods))
        ka = sum(R.augmented_kl(l, M.local_vars[t], self.yb[t], M.elbo_mode) for t, l in enumerate(M.likelihoods))
        if self.lo != 0:
            e = ka = 0.0
        kg = sum(R.gaussian_kl(g.mu, g.mu0, g.Sigma, g.L) for g in self.own)
        return self.rho * e - kg - self.rho * ka, (e, kg, ka)


def _mo_data():
    rng = np.random.default_rng(23)
    N, D, m, B, iters, Q = 150, 2, 10, 50, 5, 4
    X = rng.random((N, D))
    f = np.sin(4 * X[:, 0]) + X[:, 1]
    liks = [R.GaussianLikelihood(0.05), R.LogisticLikelihood(), R.StudentTLikelihood(3.0)]
    ys = [f + 0.1 * rng.standard_normal(N), np.where(f > f.mean(), 1.0, -1.0), f ** 2 + 0.1 * rng.standard_normal(N)]
    Zs = [X[rng.permutation(N)[:m]].copy() for _ in range(Q)]
    A = rng.standard_normal((3, Q))
    A /= np.linalg.norm(A, axis=1, keepdims=True)
    idx = [rng.choice(N, B, replace=False) for _ in range(iters)]
    return X, ys, liks, Zs, A, idx, N, B, iters, Q


def _mo_worker(rank, world, port, q):
    import sys

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import agp_amd  # noqa: F401
    from agp_amd import parallel as P

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    X, ys, liks, Zs, A, idx, N, B, iters, Q = _mo_data()
    lo, hi = P.latent_slice(Q, world, rank)
    eng = MOOracleEngine(R.Kernel("sqexponential", 3.0, 1.0), liks, Zs, A, X, ys, lo, hi, B, R.Adam(0.01))
    elbos = []
    for it in range(iters):
        P.latent_parallel_step(eng, idx[it], N / B)
        elbos.append(P.elbo_parallel(eng, "latent"))
    q.put((rank, lo, hi, [(g.eta1, g.eta2) for g in eng.own], eng.M.A.copy(), elbos))
    dist.barrier()
    dist.destroy_process_group()


def test_multioutput_latent_sharded_two_ranks():
    """MOSVGP with its 4 latents over 2 ranks: one exchange of (mean_f, var_f) per step (+ one per ELBO), update_A! replicated.
    Equals the single-process oracle model (training.jl:153-158) step for step."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_mo_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    X, ys, liks, Zs, A, idx, N, B, iters, Q = _mo_data()
    ref = R.MOSVGP(R.Kernel("sqexponential", 3.0, 1.0), liks, Zs, A.copy(), stochastic=True, batchsize=B, A_opt=R.Adam(0.01))
    elbos = []
    ref.train(X, ys, iters, idx_stream=idx, callback=lambda M, it, xb, yb: elbos.append(M.elbo(yb)))
    for rank, lo, hi, st, A2, el in res:
        for k in range(hi - lo):
            assert np.allclose(st[k][0], ref.latents[lo + k].eta1, rtol=1e-9, atol=1e-11)
            assert np.allclose(st[k][1], ref.latents[lo + k].eta2, rtol=1e-9, atol=1e-11)
        assert np.allclose(A2, ref.A, rtol=1e-10, atol=1e-12)
        assert np.allclose(el, elbos, rtol=1e-9)


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
