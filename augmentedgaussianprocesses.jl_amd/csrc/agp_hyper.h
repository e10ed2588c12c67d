// agp_hyper.h -- hand-derived reverse mode of the hyper-parameter objective for gfx950.
//
// The reference differentiates ELBO(model, x, y, mu0, kernels, Zs, state) (src/functions/ELBO.jl:15-21) with Zygote
// (src/hyperparameter/autotuning.jl:86-140; custom rule for A / ::Cholesky in zygote_rules.jl:1-8) holding (mu, Sigma,
// local variables) fixed and ignoring AugmentedKL (analyticVI.jl:269-271).  With g_mu = dE/dmu_f, g_s = dE/dsigma2_f:
//   G_kappa = rho [ g_mu mu' + 2 diag(g_s) kappa Sigma - diag(g_s) Knm ]
//   H       = G_kappa K^-1 ;  G_Knm = H - rho diag(g_s) kappa ;  G_kdiag = rho g_s
//   G_K     = -sym(kappa' H) - 1/2 [ K^-1 - K^-1 Sigma K^-1 ] + 1/2 a a' ,  a = K^-1 (mu - mu0)
// followed by the chain rule through k(x, z) = s2 * phi(|| s .* (x - z) ||^2)  (k_kernel_backward).
// The dense contractions reuse the MFMA GEMM kernels; this file holds the element-wise glue and the kernel backward.
#pragma once
#include "agp_cavi.h"

namespace agp {

// g_mu = r/rho - theta mu_f , g_s = -theta/2   (every augmented likelihood's expec_loglikelihood is quadratic in f)
// mode 1: the reference's dot(theta, mu) variant (logistic.jl:82, negativebinomial.jl:125): g_mu = r/rho - theta/2
// mode 2: the reference's BayesianSVM expression (bayesiansvm.jl:81): g_mu = y - 2 theta (1 - y mu) y
// mode 3: heteroscedastic latent f: the data term is -PoissonKL(gamma, lam0, log lam0), lam0 = lam ((y-mu)^2 + var)/2 at
//         the CURRENT (mu, var) => t = lam - 2 gamma / ((y-mu)^2 + var), g_mu = -t (mu - y), g_s = -t/2
template <typename T>
__global__ void k_hyper_gvec(int64_t B, T rho, int mode, const T* __restrict__ r, const T* __restrict__ theta,
                             const T* __restrict__ muf, const T* __restrict__ y, const int64_t* __restrict__ idx,
                             const T* __restrict__ varf, const T* __restrict__ gam, const T* __restrict__ lam,
                             T* __restrict__ gmu, T* __restrict__ gs) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= B) return;
  T th = theta[i];
  if (mode == 3) {
    T yi = y[idx ? idx[i] : i];
    T dm = muf[i] - yi;
    T t = lam[0] - T(2) * gam[i] / (dm * dm + varf[i]);
    gmu[i] = -t * (muf[i] - yi);
    gs[i] = -t / T(2);
    return;
  }
  if (mode == 2) {
    T yi = y[idx ? idx[i] : i];
    gmu[i] = yi - T(2) * th * (T(1) - yi * muf[i]) * yi;
  } else {
    gmu[i] = r[i] / rho - (mode == 1 ? th / T(2) : th * muf[i]);
  }
  gs[i] = -th / T(2);
}

// var_f = rowdot(T1, kappa) + K~ with T1 = kappa*Sigma (latentgp.jl:189) ; one wave per row
template <typename T>
__global__ void k_hyper_varf(int64_t B, int64_t cols, int64_t ld, const T* __restrict__ T1, const T* __restrict__ kappa,
                             const T* __restrict__ Kt, T* __restrict__ out) {
  const int64_t i = blockIdx.x * (int64_t)(blockDim.x >> 6) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (i >= B) return;
  T s = T(0);
  for (int64_t j = lane; j < cols; j += 64) s += T1[i * ld + j] * kappa[i * ld + j];
  for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o);
  if (lane == 0) out[i] = s + Kt[i];
}

// in place on T1 = kappa*Sigma :  Gk = rho ( g_mu mu' + 2 g_s T1 - g_s Knm ) ; rows >= B are zero
template <typename T>
__global__ void k_hyper_gkappa(int64_t B, int64_t rows, int64_t cols, int64_t ld, T rho, const T* __restrict__ gmu,
                               const T* __restrict__ gs, const T* __restrict__ mu, const T* __restrict__ Knm,
                               T* __restrict__ T1) {
  int64_t i = blockIdx.y * (int64_t)blockDim.y + threadIdx.y;
  int64_t j = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= rows || j >= cols) return;
  T v = T(0);
  if (i < B) v = rho * (gmu[i] * mu[j] + T(2) * gs[i] * T1[i * ld + j] - gs[i] * Knm[i * ld + j]);
  T1[i * ld + j] = v;
}

// G_Knm = H - rho g_s kappa   (rows >= B zero)
template <typename T>
__global__ void k_hyper_gknm(int64_t B, int64_t rows, int64_t cols, int64_t ld, T rho, const T* __restrict__ gs,
                             const T* __restrict__ H, const T* __restrict__ kappa, T* __restrict__ out) {
  int64_t i = blockIdx.y * (int64_t)blockDim.y + threadIdx.y;
  int64_t j = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= rows || j >= cols) return;
  out[i * ld + j] = (i < B) ? H[i * ld + j] - rho * gs[i] * kappa[i * ld + j] : T(0);
}

// Round 4 (one GEMM and two element-wise launches fewer): with T1 = kappa (Sigma K^-1) -- ONE product, the intermediate
// K^-1 Sigma of Apred is reused -- and a = K^-1 mu,
//   H     = G_kappa K^-1 = rho ( g_mu a' + g_s (2 T1 - kappa) )        (Knm K^-1 IS kappa)
//   G_Knm = H - rho g_s kappa
// rows >= B are zero.
template <typename T>
__global__ void k_hyper_hk(int64_t B, int64_t rows, int64_t cols, int64_t ld, T rho, const T* __restrict__ gmu,
                           const T* __restrict__ gs, const T* __restrict__ a, const T* __restrict__ T1,
                           const T* __restrict__ kappa, T* __restrict__ H, T* __restrict__ Gknm) {
  int64_t i = blockIdx.y * (int64_t)blockDim.y + threadIdx.y;
  int64_t j = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= rows || j >= cols) return;
  T h = T(0), g = T(0);
  if (i < B) {
    const T k = kappa[i * ld + j], s = rho * gs[i];
    h = rho * gmu[i] * a[j] + s * (T(2) * T1[i * ld + j] - k);
    g = h - s * k;
  }
  H[i * ld + j] = h;
  Gknm[i * ld + j] = g;
}

// (round 5: the tile form of this pass, k_hyper_hk_tile, which also left the partial column sums of u = kappa' g_mu for the fused G_K,
//  is the epilogue of the product in front of it now: k_gemm_nt<EPI_HK>, agp_linalg.h)

// G_K from ONE m^3 product (round 4).  With S = kappa' diag(w) kappa of the step (w = rho grad_E_Sigma = -rho g_s), C = S + K^-1/4
// (left by the prologue of the factorisation launch, ProArgs::Cout), M = Sigma K^-1, a = K^-1 mu, at = a - K^-1 mu0, u = kappa' g_mu:
//   kappa' H = rho u a' - 2 S M + S         K^-1 - K^-1 Sigma K^-1 = K^-1 - K^-1 M
//   G_K = -1/2 sym2(kappa' H) - 1/2 (K^-1 - K^-1 M) + 1/2 at at'
//       = TM + TM' - C - K^-1/4 - rho/2 (u a' + a u') + 1/2 at at' ,   TM = C M
// instead of the two products kappa' H (2 B m^2) and K^-1 (Sigma K^-1).  Valid m x m block; zero in the padding.
// 32 x 32 output tiles, 256 threads (32 x 8): the transposed tile of TM travels through LDS (read directly, TM[j][i] cost the
// launch 4 us of 32-byte segments); grid = (mp / 32, mp / 32).
template <typename T>
__global__ __launch_bounds__(256) void k_hyper_gK_fused(int64_t m, int64_t mp, const T* __restrict__ TM, const T* __restrict__ C,
                                                        const T* __restrict__ Kinv, const T* __restrict__ a,
                                                        const T* __restrict__ a2, const T* __restrict__ upart, int nchunk,
                                                        int64_t ldu, T rho, T* __restrict__ out) {
  __shared__ T tt[32][33];
  __shared__ T ui[32], uj[32], up[4][64];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5, t = threadIdx.x;
  const int64_t i0 = blockIdx.y * (int64_t)32, j0 = blockIdx.x * (int64_t)32;
  {  // u of the tile's 32 rows and 32 columns: four threads per entry fetch the tile-row partials side by side, fixed order
    const int el = t & 63, cg = t >> 6;
    const int64_t e = el < 32 ? i0 + el : j0 + (el - 32);
    T s = T(0);
    for (int c = cg; c < nchunk; c += 4) s += upart[c * ldu + e];
    up[cg][el] = s;
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) tt[ty + 8 * q][tx] = TM[(j0 + ty + 8 * q) * mp + i0 + tx];  // tile (j0.., i0..): TM[j][i]
  __syncthreads();
  if (t < 64) {
    const T s = (up[0][t] + up[1][t]) + (up[2][t] + up[3][t]);
    (t < 32 ? ui[t] : uj[t - 32]) = s;
  }
  __syncthreads();
  const int64_t j = j0 + tx;
  const T aj = a[j], atj = aj - (a2 ? a2[j] : T(0)), ujv = uj[tx];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int r = ty + 8 * q;
    const int64_t i = i0 + r;
    T v = T(0);
    if (i < m && j < m) {
      const T ai = a[i], ati = ai - (a2 ? a2[i] : T(0));
      v = TM[i * mp + j] + tt[tx][r] - C[i * mp + j] - T(0.25) * Kinv[i * mp + j] - T(0.5) * rho * (ui[r] * aj + ai * ujv) +
          T(0.5) * ati * atj;
    }
    out[i * mp + j] = v;
  }
}

// mean_f = kappa mu (one wave per row) and, in the same launch, k_hyper_gvec's modes 0 - 2 from it
// (Kinv != nullptr: waves B .. B + cols - 1 also form a = K^-1 mu, the symmetric mat-vec that was a launch of its own in front)
template <typename T>
__global__ void k_hyper_muf_gvec(int64_t B, int64_t cols, int64_t ld, T rho, int mode, const T* __restrict__ kappa,
                                 const T* __restrict__ mu, const T* __restrict__ r, const T* __restrict__ theta,
                                 const T* __restrict__ y, const int64_t* __restrict__ idx, T* __restrict__ muf,
                                 T* __restrict__ gmu, T* __restrict__ gs, const T* __restrict__ Kinv = nullptr,
                                 T* __restrict__ a_out = nullptr) {
  const int64_t i = blockIdx.x * (int64_t)(blockDim.x >> 6) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (i >= B) {
    const int64_t q = i - B;
    if (!Kinv || q >= cols) return;
    T s = T(0);
    for (int64_t k = lane; k < cols; k += 64) s += Kinv[q * ld + k] * mu[k];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o);
    if (lane == 0) a_out[q] = s;
    return;
  }
  T s = T(0);
  for (int64_t k = lane; k < cols; k += 64) s += kappa[i * ld + k] * mu[k];
  for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o);
  if (lane != 0) return;
  muf[i] = s;
  const T th = theta[i];
  if (mode == 2) {
    const T yi = y[idx ? idx[i] : i];
    gmu[i] = yi - T(2) * th * (T(1) - yi * s) * yi;
  } else {
    gmu[i] = r[i] / rho - (mode == 1 ? th / T(2) : th * s);
  }
  gs[i] = -th / T(2);
}

// G_K = -1/2 (M1 + M1') - 1/2 Apred + 1/2 a a'   on the valid m x m block, zero in the padding
// klw weighs the Gaussian-KL part (-1/2 Apred + 1/2 a a'): 1 normally, 1 / world on a batch-sharded handle, where that part is
// replicated on every rank while the data part (M1) is a sum over the ranks' shards -- the all-reduced gradient then counts it once
// (a = a1 - a2 when a2 is given: K^-1 mu - K^-1 mu0 without a kernel of its own in front)
template <typename T>
__global__ void k_hyper_gK(int64_t m, int64_t mp, const T* __restrict__ M1, const T* __restrict__ Apred,
                           const T* __restrict__ a, T* __restrict__ out, T klw, const T* __restrict__ a2 = nullptr) {
  int64_t i = blockIdx.y * (int64_t)blockDim.y + threadIdx.y;
  int64_t j = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= mp || j >= mp) return;
  T v = T(0);
  if (i < m && j < m) {
    const T ai = a[i] - (a2 ? a2[i] : T(0)), aj = a[j] - (a2 ? a2[j] : T(0));
    v = T(-0.5) * (M1[i * mp + j] + M1[j * mp + i]) + klw * (T(-0.5) * Apred[i * mp + j] + T(0.5) * ai * aj);
  }
  out[i * mp + j] = v;
}

// The optimiser rules the reference may be handed for its hyper-parameters (any Optimisers.jl rule, autotuning_utils.jl:47-82;
// unvendored, restated from the package's documented update rules): dx' = apply(rule, state, dx), which the reference ADDS
//   rule 0  ADAM(eta, (b1, b2), eps)   m = b1 m + (1 - b1) dx ; v = b2 v + (1 - b2) dx^2 ; dx' = eta mhat / (sqrt(vhat) + eps)
//   rule 1  Descent(eta)               dx' = eta dx
//   rule 2  Momentum(eta, rho)         vel = rho vel + eta dx ; dx' = vel          (the velocity lives in the first-moment slot)
__device__ __forceinline__ double opt_rule_delta(int rule, double g, double* am, double* av, int step, double eta, double b1,
                                                 double b2, double eps, double rho) {
  if (rule == 1) return eta * g;
  if (rule == 2) {
    const double vel = rho * am[0] + eta * g;
    am[0] = vel;
    return vel;
  }
  const double m = b1 * am[0] + (1.0 - b1) * g, v = b2 * av[0] + (1.0 - b2) * g * g;
  am[0] = m;
  av[0] = v;
  const double mh = m / (1.0 - pow(b1, (double)step)), vh = v / (1.0 - pow(b2, (double)step));
  return eta * mh / (sqrt(vh) + eps);
}

// update_kernel! (autotuning_utils.jl:47-67) on the device: ADAM ASCENT on [variance | scale(s)] in log space,
//   x <- exp(log x + ADAM(x .* g)) ,
// from the gradient g = [dvariance, dscale_0 .. dscale_{D-1}] (doubles, w.r.t. the parameters themselves) left by the backward pass.
// params = the kernel's device parameter array [scale_0 .. scale_{D-1} | variance]; am / av: ADAM moments [1 + (ard ? D : 1)];
// a ScaleTransform has ONE scale (its gradient is the sum over dimensions, all D entries carry the same value).  Structural like
// the reference's Zygote gradient (autotuning.jl:99-118): a parameter that does not exist in the kernel object (has_variance /
// has_transform = 0) is never stepped and keeps zero moments.  One workgroup; thread j steps parameter j.
template <typename T>
__global__ void k_adam_kernel_params(int D, int ard, int has_variance, int has_transform, const double* __restrict__ g,
                                     T* __restrict__ params, double* __restrict__ am, double* __restrict__ av, int step, double eta,
                                     double b1, double b2, double eps, int rule = 0, double rho = 0.0) {
  const int np = 1 + (ard ? D : 1);
  const int j = threadIdx.x;
  if (j >= np) return;
  const double p = j == 0 ? (double)params[D] : (double)params[ard ? j - 1 : 0];
  double gl;
  if (j == 0) {
    gl = has_variance ? p * g[0] : 0.0;
  } else if (ard) {
    gl = has_transform ? p * g[j] : 0.0;
  } else {
    double sgm = 0.0;
    for (int d = 0; d < D; ++d) sgm += g[1 + d];
    gl = has_transform ? p * sgm : 0.0;
  }
  const double np_ = exp(log(p) + opt_rule_delta(rule, gl, am + j, av + j, step, eta, b1, b2, eps, rho));
  if (j == 0) {
    if (has_variance) params[D] = (T)np_;
    else am[0] = av[0] = 0.0;
  } else if (has_transform) {
    if (ard) params[j - 1] = (T)np_;
    else
      for (int d = 0; d < D; ++d) params[d] = (T)np_;
  }
}

template <typename T>
__device__ __forceinline__ T kernel_dbase(int kind, T d2) {
  d2 = d2 > T(0) ? d2 : T(0);
  if (kind == K_SQEXP) return T(-0.5) * exp(T(-0.5) * d2);
  T r = sqrt(d2);
  if (kind == K_MATERN52) {
    const T s5 = T(2.23606797749978969641);
    return -(T(5) / T(6)) * (T(1) + s5 * r) * exp(-s5 * r);
  }
  const T s3 = T(1.73205080756887729353);  // Matern32
  return T(-1.5) * exp(-s3 * r);
}

// phi and phi' of one squared distance with ONE exponential (the backward pass evaluated kernel_base and kernel_dbase separately:
// two software exp per value in fp64).  Same expressions as kernel_base (agp_cavi.h) / kernel_dbase.
template <typename T>
__device__ __forceinline__ void kernel_base_dbase(int kind, T d2, T& base, T& dbase) {
  if (kind == K_SQEXP) {
    const T e = exp(T(-0.5) * (d2 > T(0) ? d2 : T(0)));
    base = e;
    dbase = T(-0.5) * e;
    return;
  }
  base = kernel_base<T>(kind, d2);
  dbase = kernel_dbase<T>(kind, d2);
}

// ---------------------------------------------------------------------------------------------------
// Backward of k_kernelmatrix for one 64x64 tile: given G = dL/dk(x_i, z_j),
//   dvar   += sum_ij G_ij phi_ij
//   dscale_d += (2/s_d) sum_ij c_ij t_ijd^2         c = G * variance * phi'(d2) ,  t = s_d (x_d - z_d)
//   dZ[j][d] += -2 s_d sum_i c_ij t_ijd               (gradient w.r.t. the SECOND argument z_j)
// Partials (deterministic two-stage reduction):  pvar[tile], pscale[tile][D], pZ[tile row][p][D].
// grid = (ceil(p/64), ceil(n/RT)) (RT rows of X per workgroup, see below).  D <= HB_MAXD.
// ---------------------------------------------------------------------------------------------------
constexpr int HB_MAXD = 64;

// RT: rows of X per workgroup (the tile is RT x 64), grid = (ceil(p / 64), ceil(n / RT)).  RT = 32 -- twice the workgroups, two per
// CU, for a launch that looked bound by the latency of its phases at one wave per SIMD -- was measured in round 4: 23.6 + 20.0 ->
// 21.6 + 19.5 us for the two passes at m = B = 1024, and the reduction behind them 10.0 -> 17.7 us (twice the partial sums): the
// pass is bound by its LDS traffic (nine ds_read per row of C in pass 2), not by occupancy.  64 it stays.
constexpr int HB_RT = 64;
// (by: the workgroup's row block -- the y index of a launch of its own, or its offset inside a launch that carries two passes)
template <typename T, int RT = HB_RT>
__device__ __forceinline__ void kernel_backward_body(const T* __restrict__ X, int64_t ldx, const int64_t* __restrict__ idx, int64_t n,
                                                     const T* __restrict__ Y, int64_t ldy, int64_t p, int64_t D,
                                                     const T* __restrict__ scales, int kind, T variance,
                                                     const T* __restrict__ G, int64_t ldg, double* __restrict__ pvar,
                                                     double* __restrict__ pscale, T* __restrict__ pZ, int64_t p_pad, int64_t by) {
  if (variance < T(0)) variance = scales[D];  // device-resident kernel parameters (see k_kernelmatrix)
  static_assert(RT == 32 || RT == 64, "k_kernel_backward: 32 or 64 rows per workgroup");
  constexpr int NA = RT / 16;  // row groups of 16 per thread
  __shared__ T xs[RT][KM_DC + 1];
  __shared__ T ys[TILE][KM_DC + 1];
  __shared__ T Cs[RT][TILE + 1];  // c_ij of the tile (pass 2 contracts it against the staged coordinates)
  __shared__ T rows[RT];
  __shared__ double red[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ty = tid >> 4, tx = tid & 15;
  const int64_t i0 = by * (int64_t)RT, j0 = blockIdx.x * (int64_t)TILE;
  const int64_t tile = by * (int64_t)gridDim.x + blockIdx.x;
  T acc[NA][4];
#pragma unroll
  for (int a = 0; a < NA; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = T(0);
  // a thread stages one dimension of NST rows; the row indices are resolved once (the gather's index loads used to sit in front of
  // every coordinate load: at one wave per SIMD the two dependent round trips per element were most of the kernel's time)
  constexpr int NST = TILE * KM_DC / NTHREADS, NSX = RT * KM_DC / NTHREADS;
  static_assert(NTHREADS % KM_DC == 0 && TILE * KM_DC % NTHREADS == 0 && RT * KM_DC % NTHREADS == 0, "k_kernel_backward: staging layout");
  const int sd = tid % KM_DC, sr0 = tid / KM_DC;
  int64_t xrow[NSX];
#pragma unroll
  for (int q = 0; q < NSX; ++q) {
    const int64_t gi = i0 + sr0 + q * (NTHREADS / KM_DC);
    xrow[q] = gi < n ? (idx ? idx[gi] : gi) : (int64_t)-1;
  }
  auto stage = [&](int64_t d0) {
    const int64_t gd = d0 + sd;
    const bool dok = gd < D;
    const T sc = (scales && dok) ? scales[gd] : T(1);
    T xv[NSX], yv[NST];
#pragma unroll
    for (int q = 0; q < NSX; ++q) xv[q] = (dok && xrow[q] >= 0) ? X[xrow[q] * ldx + gd] : T(0);
#pragma unroll
    for (int q = 0; q < NST; ++q) {
      const int64_t gj = j0 + sr0 + q * (NTHREADS / KM_DC);
      yv[q] = (dok && gj < p) ? Y[gj * ldy + gd] : T(0);
    }
#pragma unroll
    for (int q = 0; q < NSX; ++q) xs[sr0 + q * (NTHREADS / KM_DC)][sd] = xv[q] * sc;
#pragma unroll
    for (int q = 0; q < NST; ++q) ys[sr0 + q * (NTHREADS / KM_DC)][sd] = yv[q] * sc;
  };
  // pass 1: squared distances
  for (int64_t d0 = 0; d0 < D; d0 += KM_DC) {
    stage(d0);
    __syncthreads();
#pragma unroll 8
    for (int d = 0; d < KM_DC; ++d) {
      T xv[NA], yv[4];
#pragma unroll
      for (int a = 0; a < NA; ++a) xv[a] = xs[ty + 16 * a][d];
#pragma unroll
      for (int b = 0; b < 4; ++b) yv[b] = ys[tx + 16 * b][d];
#pragma unroll
      for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          T df = xv[a] - yv[b];
          acc[a][b] += df * df;
        }
    }
    __syncthreads();
  }
  // c_ij = G_ij * variance * phi'(d2) ; dvar partial
  double dv = 0.0;
#pragma unroll
  for (int a = 0; a < NA; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      int64_t gi = i0 + ty + 16 * a, gj = j0 + tx + 16 * b;
      T g = (gi < n && gj < p) ? G[gi * ldg + gj] : T(0);
      T kb, kd;
      kernel_base_dbase<T>(kind, acc[a][b], kb, kd);
      dv += (double)(g * kb);
      acc[a][b] = g * variance * kd;
    }
  dv = block_sum<double>(dv, red);
  if (tid == 0) pvar[tile] = dv;
  // pass 2: per-dimension reductions in product form (round 4: the per-dimension shuffle reductions of the first version cost
  // 45 us per pass at m = B = 1024; this one ~ 10).  With t_ijd = xs_id - ys_jd:
  //   sum_i c_ij t_ijd        = (C'xs)_jd - colsum_j ys_jd
  //   sum_ij c_ij t_ijd^2     = sum_i rowsum_i xs_id^2 - 2 sum_j ys_jd (C'xs)_jd + sum_j colsum_j ys_jd^2   (combined in double)
  // thread (j = lane, dg = wave) owns (C'xs)_{j, 8 dg .. 8 dg + 7} of the staged chunk.
#pragma unroll
  for (int a = 0; a < NA; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) Cs[ty + 16 * a][tx + 16 * b] = acc[a][b];
  __syncthreads();
  if (tid < RT) {
    T s = T(0);
#pragma unroll 8
    for (int j = 0; j < TILE; ++j) s += Cs[tid][j];
    rows[tid] = s;
  }
  T cols = T(0);
#pragma unroll 8
  for (int i = 0; i < RT; ++i) cols += Cs[i][lane];
  constexpr int DPT = KM_DC / 4;  // dimensions per thread
  for (int64_t d0 = 0; d0 < D; d0 += KM_DC) {
    if (D > KM_DC) {  // (one chunk: pass 1 left it staged)
      __syncthreads();
      stage(d0);
    }
    __syncthreads();
    T zs[DPT];
#pragma unroll
    for (int q = 0; q < DPT; ++q) zs[q] = T(0);
#pragma unroll 4
    for (int i = 0; i < RT; ++i) {
      const T c = Cs[i][lane];
#pragma unroll
      for (int q = 0; q < DPT; ++q) zs[q] += c * xs[i][wave * DPT + q];
    }
    const bool rl = lane < RT;  // the row-sum term lives on the tile's RT rows
    const T rw = rl ? rows[lane] : T(0);
#pragma unroll
    for (int q = 0; q < DPT; ++q) {
      const int d = wave * DPT + q;
      const int64_t gd = d0 + d, gj = j0 + lane;
      const T sc = (scales && gd < D) ? scales[gd] : T(1);
      const double xv = rl ? (double)xs[lane][d] : 0.0, yv = (double)ys[lane][d];
      double ss = (double)rw * xv * xv - 2.0 * yv * (double)zs[q] + (double)cols * yv * yv;
      for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
      if (gd < D) {
        if (gj < p_pad) pZ[(by * p_pad + gj) * D + gd] = T(-2) * sc * (zs[q] - cols * ys[lane][d]);
        if (lane == 0) pscale[tile * D + gd] = 2.0 / (double)sc * ss;
      }
    }
  }
}
template <typename T, int RT = HB_RT>
__global__ __launch_bounds__(NTHREADS) void k_kernel_backward(const T* __restrict__ X, int64_t ldx,
                                                              const int64_t* __restrict__ idx, int64_t n,
                                                              const T* __restrict__ Y, int64_t ldy, int64_t p, int64_t D,
                                                              const T* __restrict__ scales, int kind, T variance,
                                                              const T* __restrict__ G, int64_t ldg,
                                                              double* __restrict__ pvar, double* __restrict__ pscale,
                                                              T* __restrict__ pZ, int64_t p_pad) {
  kernel_backward_body<T, RT>(X, ldx, idx, n, Y, ldy, p, D, scales, kind, variance, G, ldg, pvar, pscale, pZ, p_pad,
                              (int64_t)blockIdx.y);
}
// The backward passes through K_nm = k(x, Z) and K_ZZ = k(Z, Z) of one gradient evaluation in ONE launch (round 5): they are
// independent (both read a finished G), each is 256 workgroups of one wave per SIMD at m = B = 1024 and bound by the latency of
// its phases -- side by side they overlap (24.4 + 20.1 us as two launches).  Row blocks [0, ny1): first pass; the rest: second.
// Same second argument Y (the inducing points) and kernel; each pass writes its own partial sums, exactly as on its own.
template <typename T>
struct KbPass {
  const T* X;
  int64_t ldx;
  const int64_t* idx;
  int64_t n;
  const T* G;
  int64_t ldg;
  double* pvar;
  double* pscale;
  T* pZ;
};
template <typename T, int RT = HB_RT>
__global__ __launch_bounds__(NTHREADS) void k_kernel_backward2(KbPass<T> a, KbPass<T> b, int64_t ny1, const T* __restrict__ Y,
                                                               int64_t ldy, int64_t p, int64_t D, const T* __restrict__ scales,
                                                               int kind, T variance, int64_t p_pad) {
  const bool second = (int64_t)blockIdx.y >= ny1;  // (uniform: the selects below are scalar)
  kernel_backward_body<T, RT>(second ? b.X : a.X, second ? b.ldx : a.ldx, second ? b.idx : a.idx, second ? b.n : a.n, Y, ldy, p, D,
                              scales, kind, variance, second ? b.G : a.G, second ? b.ldg : a.ldg, second ? b.pvar : a.pvar,
                              second ? b.pscale : a.pscale, second ? b.pZ : a.pZ, p_pad, (int64_t)blockIdx.y - (second ? ny1 : 0));
}

// out[0] += wgt * sum pvar ; out[1 + d] += wgt * sum_tiles pscale[tile][d] : one workgroup per output, fixed-order tree
template <typename T>
__global__ void k_hyper_reduce_scalar(int64_t ntiles, int64_t D, const double* __restrict__ pvar,
                                      const double* __restrict__ pscale, double* __restrict__ out, double wgt) {
  __shared__ double red[16];
  const int d = blockIdx.x;  // 0: variance, 1..D: scales
  double s = 0.0;
  for (int64_t t = threadIdx.x; t < ntiles; t += blockDim.x) s += (d == 0) ? pvar[t] : pscale[t * D + d - 1];
  s = block_sum<double>(s, red);
  if (threadIdx.x == 0) out[d] += wgt * s;
}

// dZ[j][d] (+)= wgt * sum_by pZ[by][j][d]
template <typename T>
__global__ void k_hyper_reduce_Z(int64_t nrowtiles, int64_t p, int64_t p_pad, int64_t D, const T* __restrict__ pZ,
                                 T* __restrict__ dZ, T wgt, int accumulate) {
  int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (e >= p * D) return;
  int64_t j = e / D, d = e % D;
  T s = T(0);
  for (int64_t b = 0; b < nrowtiles; ++b) s += pZ[(b * p_pad + j) * D + d];
  dZ[e] = (accumulate ? dZ[e] : T(0)) + wgt * s;
}

// sum over the first B entries of x -> out[0] += wgt * sum   (G_kdiag term of the variance gradient)
template <typename T>
__global__ void k_hyper_sum(int64_t B, const T* __restrict__ x, double wgt, double* __restrict__ out) {
  __shared__ double red[16];
  double s = 0.0;
  for (int64_t i = threadIdx.x; i < B; i += blockDim.x) s += (double)x[i];
  s = block_sum<double>(s, red);
  if (threadIdx.x == 0) out[0] += wgt * s;
}

// the two reductions behind one backward pass in ONE launch (they were two, plus a memset in front of the first pass): workgroups
// 0 .. D take the scalar outputs (k_hyper_reduce_scalar), the rest the inducing-point gradient (k_hyper_reduce_Z).  init: this is
// the first pass of a gradient evaluation -- the scalars start from zero (+ the kdiag term  wgt_kd * sum_{i < Bkd} xkd[i]  on the
// variance, k_hyper_sum's job) and dZ is overwritten.
template <typename T>
__global__ void k_hyper_reduce(int64_t ntiles, int64_t D, const double* __restrict__ pvar, const double* __restrict__ pscale,
                               double* __restrict__ out, double wgt, int init, int64_t nrowtiles, int64_t p, int64_t p_pad,
                               const T* __restrict__ pZ, T* __restrict__ dZ, T wgtZ, const T* __restrict__ xkd, int64_t Bkd,
                               double wgt_kd) {
  if ((int64_t)blockIdx.x <= D) {
    __shared__ double red[16];
    const int d = blockIdx.x;  // 0: variance, 1..D: scales
    double s = 0.0;
    for (int64_t t = threadIdx.x; t < ntiles; t += blockDim.x) s += (d == 0) ? pvar[t] : pscale[t * D + d - 1];
    s = block_sum<double>(s, red);
    double extra = 0.0;
    if (d == 0 && xkd) {
      double q = 0.0;
      for (int64_t i = threadIdx.x; i < Bkd; i += blockDim.x) q += (double)xkd[i];
      extra = wgt_kd * block_sum<double>(q, red);
    }
    if (threadIdx.x == 0) out[d] = (init ? 0.0 : out[d]) + wgt * s + extra;
    return;
  }
  if (!dZ) return;
  const int64_t e = ((int64_t)blockIdx.x - (D + 1)) * (int64_t)blockDim.x + threadIdx.x;
  if (e >= p * D) return;
  const int64_t j = e / D, d = e % D;
  T s = T(0);
  for (int64_t b = 0; b < nrowtiles; ++b) s += pZ[(b * p_pad + j) * D + d];
  dZ[e] = (init ? T(0) : dZ[e]) + wgtZ * s;
}

// The reductions behind BOTH backward passes of a gradient evaluation in one launch (round 4): set 1 = the pass through K_nm (weight
// 1 on the scalars and on Z, + the kdiag term of the variance), set 2 = the pass through K_ZZ (weight 1 on the scalars, 2 on Z: both
// arguments are Z and G_K is symmetric).  Same per-set summation order as two k_hyper_reduce launches (init = 1, then init = 0).
template <typename T>
__global__ void k_hyper_reduce2(int64_t D, double* __restrict__ out, int64_t p, int64_t p_pad, T* __restrict__ dZ, int64_t ntiles1,
                                const double* __restrict__ pvar1, const double* __restrict__ pscale1, int64_t nrow1,
                                const T* __restrict__ pZ1, int64_t ntiles2, const double* __restrict__ pvar2,
                                const double* __restrict__ pscale2, int64_t nrow2, const T* __restrict__ pZ2,
                                const T* __restrict__ xkd, int64_t Bkd, double wgt_kd) {
  if ((int64_t)blockIdx.x <= D) {
    __shared__ double red[16];
    const int d = blockIdx.x;  // 0: variance, 1..D: scales
    double s1 = 0.0, s2 = 0.0;
    for (int64_t t = threadIdx.x; t < ntiles1; t += blockDim.x) s1 += (d == 0) ? pvar1[t] : pscale1[t * D + d - 1];
    s1 = block_sum<double>(s1, red);
    for (int64_t t = threadIdx.x; t < ntiles2; t += blockDim.x) s2 += (d == 0) ? pvar2[t] : pscale2[t * D + d - 1];
    s2 = block_sum<double>(s2, red);
    double extra = 0.0;
    if (d == 0 && xkd) {
      double q = 0.0;
      for (int64_t i = threadIdx.x; i < Bkd; i += blockDim.x) q += (double)xkd[i];
      extra = wgt_kd * block_sum<double>(q, red);
    }
    if (threadIdx.x == 0) out[d] = ((0.0 + 1.0 * s1 + extra)) + 1.0 * s2;
    return;
  }
  if (!dZ) return;
  const int64_t e = ((int64_t)blockIdx.x - (D + 1)) * (int64_t)blockDim.x + threadIdx.x;
  if (e >= p * D) return;
  const int64_t j = e / D, d = e % D;
  // (the loads of a set are independent: issued in groups of four, added in row-tile order as two k_hyper_reduce launches did)
  T a = T(0), b = T(0);
  int64_t r = 0;
  for (; r + 4 <= nrow1; r += 4) {
    const T x0 = pZ1[(r * p_pad + j) * D + d], x1 = pZ1[((r + 1) * p_pad + j) * D + d];
    const T x2 = pZ1[((r + 2) * p_pad + j) * D + d], x3 = pZ1[((r + 3) * p_pad + j) * D + d];
    a = (((a + x0) + x1) + x2) + x3;
  }
  for (; r < nrow1; ++r) a += pZ1[(r * p_pad + j) * D + d];
  for (r = 0; r + 4 <= nrow2; r += 4) {
    const T x0 = pZ2[(r * p_pad + j) * D + d], x1 = pZ2[((r + 1) * p_pad + j) * D + d];
    const T x2 = pZ2[((r + 2) * p_pad + j) * D + d], x3 = pZ2[((r + 3) * p_pad + j) * D + d];
    b = (((b + x0) + x1) + x2) + x3;
  }
  for (; r < nrow2; ++r) b += pZ2[(r * p_pad + j) * D + d];
  dZ[e] = (T(0) + T(1) * a) + T(2) * b;
}

// tied-Z mode: gradients of several latents are summed (and all-reduced across ranks) in double
template <typename T>
__global__ void k_acc_to_double(int64_t n, const T* __restrict__ g, double* __restrict__ acc) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < n) acc[i] += (double)g[i];
}
template <typename T>
__global__ void k_double_to(int64_t n, const double* __restrict__ a, T* __restrict__ out) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < n) out[i] = (T)a[i];
}

// ADAM ascent on Z (update_Z!, autotuning_utils.jl:70-76): z += eta * mhat / (sqrt(vhat) + eps)
template <typename T>
__global__ void k_adam_ascent(int64_t n, T* __restrict__ z, const T* __restrict__ g, double* __restrict__ am,
                              double* __restrict__ av, int step, double eta, double b1, double b2, double eps, int rule = 0,
                              double rho = 0.0) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  z[i] = (T)((double)z[i] + opt_rule_delta(rule, (double)g[i], am + i, av + i, step, eta, b1, b2, eps, rho));
}

// both optimiser steps of a latent in ONE launch (round 4): workgroups 0 .. nzb - 1 are k_adam_ascent on Z, the last one is
// k_adam_kernel_params (one thread per kernel parameter) -- they touch disjoint state
template <typename T>
__global__ void k_adam_z_and_params(int64_t nz, int64_t nzb, T* __restrict__ z, const T* __restrict__ gz, double* __restrict__ zm,
                                    double* __restrict__ zv, int zstep, double zeta, int zrule, double zrho, int D, int ard,
                                    int has_variance, int has_transform, const double* __restrict__ g, T* __restrict__ params,
                                    double* __restrict__ am, double* __restrict__ av, int kstep, double keta, int krule,
                                    double krho, double b1, double b2, double eps) {
  if ((int64_t)blockIdx.x < nzb) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= nz) return;
    z[i] = (T)((double)z[i] + opt_rule_delta(zrule, (double)gz[i], zm + i, zv + i, zstep, zeta, b1, b2, eps, zrho));
    return;
  }
  const int np = 1 + (ard ? D : 1);
  const int j = threadIdx.x;
  if (j >= np) return;
  const double p = j == 0 ? (double)params[D] : (double)params[ard ? j - 1 : 0];
  double gl;
  if (j == 0) {
    gl = has_variance ? p * g[0] : 0.0;
  } else if (ard) {
    gl = has_transform ? p * g[j] : 0.0;
  } else {
    double sgm = 0.0;
    for (int d = 0; d < D; ++d) sgm += g[1 + d];
    gl = has_transform ? p * sgm : 0.0;
  }
  const double np_ = exp(log(p) + opt_rule_delta(krule, gl, am + j, av + j, kstep, keta, b1, b2, eps, krho));
  if (j == 0) {
    if (has_variance) params[D] = (T)np_;
    else am[0] = av[0] = 0.0;
  } else if (has_transform) {
    if (ard) params[j - 1] = (T)np_;
    else
      for (int d = 0; d < D; ++d) params[d] = (T)np_;
  }
}

// k_hyper_reduce2 + k_adam_z_and_params in ONE launch (round 5): the hyper step of the training loop never looks at the gradient
// between the two, and every kernel costs the in-order queue ~5 us.  Same arithmetic in the same order as the two kernels -- the
// scalar reductions reproduce block_sum's order (thread-strided partial sums, halving shuffles per wave, waves added in order)
// with one WAVE standing in for the 256-thread workgroup, so that hypergrad + hyper_apply (the split a tied-Z driver all-reduces
// in between) stays bit-identical to hyper_step.
//   workgroup 0: the D + 1 kernel-parameter gradients (waves take them in turn), out[0 .. D], then the parameters' optimiser step;
//   workgroups 1 ..: elements of Z -- reduce dZ[e] from both sets of partial sums, store it, take its optimiser step.
// emulation of block_sum<double> over two sequences f1(t), t < n1, and f2(t), t < n2 (t = thread index of a 256-thread workgroup) by
// one wave; the first element of every virtual thread is fetched before any arithmetic (the usual case, n <= 256, is then ONE round
// of eight independent loads -- a wave that takes several gradients in turn pays a memory latency per round, not per load)
template <typename F1, typename F2>
__device__ __forceinline__ void wave_as_block_sum2(int64_t n1, F1 f1, int64_t n2, F2 f2, double& r1, double& r2) {
  const int lane = threadIdx.x & 63;
  double s1[4], s2[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {  // virtual wave q of the 256-thread workgroup: threads 64 q .. 64 q + 63
    s1[q] = lane + 64 * q < n1 ? f1(lane + 64 * q) : 0.0;
    s2[q] = lane + 64 * q < n2 ? f2(lane + 64 * q) : 0.0;
  }
  double t1 = 0.0, t2 = 0.0;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    double a = 0.0 + s1[q], b = 0.0 + s2[q];
    for (int64_t t = lane + 64 * q + 256; t < n1; t += 256) a += f1(t);
    for (int64_t t = lane + 64 * q + 256; t < n2; t += 256) b += f2(t);
    for (int o = 32; o > 0; o >>= 1) {
      a += __shfl_down(a, o);
      b += __shfl_down(b, o);
    }
    t1 += __shfl(a, 0);
    t2 += __shfl(b, 0);
  }
  r1 = t1;
  r2 = t2;
}
constexpr int HYPER_RA_THREADS = 1024;  // workgroup 0 takes the D + 1 scalar gradients wave by wave: 16 waves
template <typename T>
__global__ __launch_bounds__(HYPER_RA_THREADS) void k_hyper_reduce2_adam(
    int64_t D, double* __restrict__ out, int64_t p, int64_t p_pad, T* __restrict__ dZ, int64_t ntiles1, const double* __restrict__ pvar1,
    const double* __restrict__ pscale1, int64_t nrow1, const T* __restrict__ pZ1, int64_t ntiles2, const double* __restrict__ pvar2,
    const double* __restrict__ pscale2, int64_t nrow2, const T* __restrict__ pZ2, const T* __restrict__ xkd, int64_t Bkd, double wgt_kd,
    // optimiser state (k_adam_z_and_params)
    T* __restrict__ z, double* __restrict__ zm, double* __restrict__ zv, int zstep, double zeta, int zrule, double zrho, int ard,
    int has_variance, int has_transform, T* __restrict__ params, double* __restrict__ am, double* __restrict__ av, int kstep, double keta,
    int krule, double krho, double b1, double b2, double eps) {
  if (blockIdx.x == 0) {
    __shared__ double gsh[257];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nwave = blockDim.x >> 6;
    for (int d = wave; d <= (int)D; d += nwave) {
      double s1, s2, extra = 0.0, unused;
      wave_as_block_sum2(
          ntiles1, [&](int64_t t) { return d == 0 ? pvar1[t] : pscale1[t * D + d - 1]; }, ntiles2,
          [&](int64_t t) { return d == 0 ? pvar2[t] : pscale2[t * D + d - 1]; }, s1, s2);
      if (d == 0 && xkd) {
        wave_as_block_sum2(Bkd, [&](int64_t i) { return (double)xkd[i]; }, 0, [&](int64_t) { return 0.0; }, extra, unused);
        extra *= wgt_kd;
      }
      const double gd = ((0.0 + 1.0 * s1 + extra)) + 1.0 * s2;
      if (lane == 0) {
        out[d] = gd;
        gsh[d] = gd;
      }
    }
    __syncthreads();
    const int np = 1 + (ard ? (int)D : 1);
    const int j = threadIdx.x;
    if (j >= np) return;
    const double pj = j == 0 ? (double)params[D] : (double)params[ard ? j - 1 : 0];
    double gl;
    if (j == 0) {
      gl = has_variance ? pj * gsh[0] : 0.0;
    } else if (ard) {
      gl = has_transform ? pj * gsh[j] : 0.0;
    } else {
      double sgm = 0.0;
      for (int d = 0; d < (int)D; ++d) sgm += gsh[1 + d];
      gl = has_transform ? pj * sgm : 0.0;
    }
    const double np_ = exp(log(pj) + opt_rule_delta(krule, gl, am + j, av + j, kstep, keta, b1, b2, eps, krho));
    if (j == 0) {
      if (has_variance) params[D] = (T)np_;
      else am[0] = av[0] = 0.0;
    } else if (has_transform) {
      if (ard) params[j - 1] = (T)np_;
      else
        for (int d = 0; d < (int)D; ++d) params[d] = (T)np_;
    }
    return;
  }
  const int64_t e = ((int64_t)blockIdx.x - 1) * (int64_t)blockDim.x + threadIdx.x;
  if (e >= p * D) return;
  const int64_t j = e / D, d = e % D;
  T a = T(0), b = T(0);
  int64_t r = 0;
  for (; r + 4 <= nrow1; r += 4) {
    const T x0 = pZ1[(r * p_pad + j) * D + d], x1 = pZ1[((r + 1) * p_pad + j) * D + d];
    const T x2 = pZ1[((r + 2) * p_pad + j) * D + d], x3 = pZ1[((r + 3) * p_pad + j) * D + d];
    a = (((a + x0) + x1) + x2) + x3;
  }
  for (; r < nrow1; ++r) a += pZ1[(r * p_pad + j) * D + d];
  for (r = 0; r + 4 <= nrow2; r += 4) {
    const T x0 = pZ2[(r * p_pad + j) * D + d], x1 = pZ2[((r + 1) * p_pad + j) * D + d];
    const T x2 = pZ2[((r + 2) * p_pad + j) * D + d], x3 = pZ2[((r + 3) * p_pad + j) * D + d];
    b = (((b + x0) + x1) + x2) + x3;
  }
  for (; r < nrow2; ++r) b += pZ2[(r * p_pad + j) * D + d];
  const T gz = (T(0) + T(1) * a) + T(2) * b;
  dZ[e] = gz;
  z[e] = (T)((double)z[e] + opt_rule_delta(zrule, (double)gz, zm + e, zv + e, zstep, zeta, b1, b2, eps, zrho));
}

// C(M x N) = A(K x M)^T B(K x N)  (both operands row-contiguous; general, non-symmetric).  grid = (N/64, M/64)
template <typename T, int KG>
__global__ __launch_bounds__(NTHREADS * KG) void k_gemm_tn(const T* __restrict__ A, int64_t lda,
                                                           const T* __restrict__ B, int64_t ldb, int64_t K,
                                                           T* __restrict__ C, int64_t ldc) {
  __shared__ __attribute__((aligned(16))) T smem[KG * smem_elems<T>()];
  const int64_t r0 = blockIdx.y * (int64_t)TILE, c0 = blockIdx.x * (int64_t)TILE;
  Acc<T> acc;
  acc.zero();
  gemm_tile<T, RC, RC, KG>(A + r0, lda, B + c0, ldb, 0, K, nullptr, acc, smem);
  if (KG > 1 && threadIdx.x >= NTHREADS) return;
  acc_foreach<T>(acc, [&](int r, int c, T val) { C[(r0 + r) * ldc + c0 + c] = val; });
}

}  // namespace agp
