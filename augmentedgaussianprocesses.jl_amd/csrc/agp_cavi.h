// agp_cavi.h -- kernel-matrix evaluation, per-likelihood local updates, natural-parameter vector updates, ELBO
// reductions and prediction post-processing for the SVGP / AnalyticVI path on gfx950.
// Reference: src/gpblocks/latentgp.jl:171-215, src/likelihood/{gaussian,logistic,studentt,logisticsoftmax}.jl,
// src/inference/analyticVI.jl:143-274, src/functions/KLdivergences.jl, src/training/predictions.jl.
#pragma once
#include "agp_device.h"

namespace agp {

enum { K_SQEXP = 0, K_MATERN52 = 1, K_MATERN32 = 2, K_EXPONENTIAL = 3 };
enum {
  LIK_GAUSSIAN = 0,
  LIK_LOGISTIC = 1,
  LIK_STUDENTT = 2,
  LIK_LSM = 3,
  LIK_MO = 4,
  LIK_LAPLACE = 5,  // p0 = beta
  LIK_BSVM = 6,
  LIK_POISSON = 7,  // lambda lives in device memory (re-estimated by every local update)
  LIK_NEGBIN = 8,   // p0 = r
  LIK_HETERO = 9    // two latents (f, g) ; lambda in device memory
};
enum { FLAG_NEG_KTILDE = 1, FLAG_BAD_LABEL = 2 };

// ---------------------------------------------------------------------------------------------------
// exp for the kernel functions (round 6).  Every kernel of the path evaluates exp at a NON-POSITIVE argument
// (-d2/2, -sqrt(5) r, -sqrt(3) r, -r), N m times per streaming prediction: 1.07e9 values at C2.  The device library's exp costs a
// lane ~36 VALU instructions there (its polynomial as v_mov_b64 + v_fmac pairs, overflow / underflow / NaN selects); this one 17:
//   n = rint(x log2 e) ; r = x - n ln2 (two-constant Cody-Waite, exact for |n| < 2^20) ; e^r by its degree-11 Taylor polynomial on
//   |r| <= ln2 / 2 (truncation 6.3e-15 relative, Horner with the coefficients as SGPR operands of v_fma_f64) ; v_ldexp_f64.
// Arguments below -750 are clamped there (the result is exactly 0 through v_ldexp's denormal handling -- the padded "far" inducing
// points of the online model rely on exact zeros); a NaN argument gives a finite value like the d2 > 0 ? d2 : 0 in front of every
// call always did.  Max relative error against exp over [-745, 0]: 9e-15 (the truncation at |r| = ln2 / 2; tests/test_gpu_round6.py).
// fp32: v_exp_f32 with log2 e folded into the argument (1.5 ulp of the instruction + the argument's rounding, |x| 2^-24 relative).
// ---------------------------------------------------------------------------------------------------
struct ExpScalePlain {  // e^x, x <= 0
  static constexpr double ct = 1.44269504088896338700e+00;    // n = rint(ct x)
  static constexpr double chi = -6.93147180369123816490e-01;  // r = x + n chi + n clo
  static constexpr double clo = -1.90821492927058770002e-10;
  static constexpr double lim = -750.0;                       // clamp (towards zero) of x
  static constexpr double sc = 1.0;                           // polynomial variable is sc * r
};
struct ExpScaleHalfNeg {  // e^(-u/2), u >= 0: the squared-exponential kernel straight from the squared distance
  static constexpr double ct = -0.5 * 1.44269504088896338700e+00;
  static constexpr double chi = 2.0 * 6.93147180369123816490e-01;
  static constexpr double clo = 2.0 * 1.90821492927058770002e-10;
  static constexpr double lim = 1500.0;
  static constexpr double sc = -0.5;
};
template <typename S>
__device__ __forceinline__ double exp_core_d(double x) {
  // (the caller has clamped x into [lim, 0] resp. [0, lim])
  const double n = __builtin_rint(x * S::ct);
  double r = __builtin_fma(n, S::chi, x);
  r = __builtin_fma(n, S::clo, r);
  // p(r) = sum_k (sc r)^k / k!, k = 0 .. 11
  constexpr double s1 = S::sc, s2 = s1 * s1, s3 = s2 * s1, s4 = s2 * s2, s5 = s4 * s1, s6 = s3 * s3, s7 = s6 * s1, s8 = s4 * s4,
                   s9 = s8 * s1, s10 = s5 * s5, s11 = s10 * s1;
  // ONE asm statement for the Horner chain: between separate asm statements the compiler puts an s_nop each, and left to itself it
  // keeps the coefficients in VGPRs and emits v_mov_b64 + v_fmac_f64 pairs (9 more VALU instructions per value)
  double p;
  const double p0 = s11 / 39916800.0;  // (a VGPR operand that stays live across the values of a tile: VOP3 takes one SGPR source)
  asm("v_fma_f64 %0, %13, %1, %2\n\tv_fma_f64 %0, %0, %1, %3\n\tv_fma_f64 %0, %0, %1, %4\n\tv_fma_f64 %0, %0, %1, %5\n\t"
      "v_fma_f64 %0, %0, %1, %6\n\tv_fma_f64 %0, %0, %1, %7\n\tv_fma_f64 %0, %0, %1, %8\n\tv_fma_f64 %0, %0, %1, %9\n\t"
      "v_fma_f64 %0, %0, %1, %10\n\tv_fma_f64 %0, %0, %1, %11\n\tv_fma_f64 %0, %0, %1, %12"
      : "=&v"(p)
      : "v"(r), "s"(s10 / 3628800.0), "s"(s9 / 362880.0), "s"(s8 / 40320.0), "s"(s7 / 5040.0), "s"(s6 / 720.0), "s"(s5 / 120.0),
        "s"(s4 / 24.0), "s"(s3 / 6.0), "s"(s2 / 2.0), "s"(s1), "s"(1.0), "v"(p0));
  return __builtin_amdgcn_ldexp(p, (int)n);
}
// e^x for x <= 0 (x > 0 is not an error but loses the clamp's protection against overflow of the exponent: not used that way)
__device__ __forceinline__ double exp_nonpos(double x) { return exp_core_d<ExpScalePlain>(__builtin_fmax(x, ExpScalePlain::lim)); }
__device__ __forceinline__ float exp_nonpos(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896338700f); }
// e^(-u/2) for u >= 0 (negative or NaN u: treated as 0, like kernel_base's own clamp)
__device__ __forceinline__ double exp_mhalf(double u) {
  return exp_core_d<ExpScaleHalfNeg>(__builtin_fmin(__builtin_fmax(u, 0.0), ExpScaleHalfNeg::lim));
}
__device__ __forceinline__ float exp_mhalf(float u) { return __builtin_amdgcn_exp2f(__builtin_fmaxf(u, 0.0f) * (-0.5f * 1.44269504088896338700f)); }

template <typename T>
__device__ __forceinline__ T kernel_base(int kind, T d2) {
  if (kind == K_SQEXP) return exp_mhalf(d2);
  d2 = d2 > T(0) ? d2 : T(0);
  T d = sqrt(d2);
  if (kind == K_MATERN52) {
    const T s5 = T(2.23606797749978969641);
    return (T(1) + s5 * d + T(5) * d2 / T(3)) * exp_nonpos(-s5 * d);
  }
  if (kind == K_MATERN32) {
    const T s3 = T(1.73205080756887729353);
    return (T(1) + s3 * d) * exp_nonpos(-s3 * d);
  }
  return exp_nonpos(-d);
}

// ---------------------------------------------------------------------------------------------------
// kernelmatrix(k, X[idx], Y)  (src/gpblocks/latentgp.jl:206,210 ; KernelFunctions.jl semantics):
//   out[i][j] = variance * base( || scales .* (x_i - y_j) ||^2 )     direct squared differences, no GEMM trick.
// 64x64 output tile per workgroup, 4x4 outputs per thread, operands staged through LDS in chunks of 32 dims with
// the minibatch gather (x row = X[idx[i]]) fused into the staging loads (256-byte coalesced rows at D = 32 f64).
//   rows >= n or cols >= p (padding up to n_out x p_out) are written as 0 ;
//   sym != 0 : self matrix -> adds diag_add (jitter) on the diagonal and 1 on the padded diagonal.
//   alpha != nullptr : fused row-dot  part[blockIdx.x][i] = sum_j out[i][j] alpha[j]  (predict mean without K_*m)
//   out == nullptr : nothing stored (mean-only prediction)
// ---------------------------------------------------------------------------------------------------
constexpr int KM_DC = 32;

template <typename T>
__global__ __launch_bounds__(NTHREADS) void k_kernelmatrix(const T* __restrict__ X, int64_t ldx,
                                                           const int64_t* __restrict__ idx, int64_t n,
                                                           const T* __restrict__ Y, int64_t ldy, int64_t p, int64_t D,
                                                           const T* __restrict__ scales, int kind, T variance,
                                                           T* __restrict__ out, int64_t ldo, int64_t n_out,
                                                           int64_t p_out, int sym, T diag_add,
                                                           const T* __restrict__ alpha, T* __restrict__ part,
                                                           int64_t ldp) {
  // variance < 0: the kernel parameters are device-resident state (they are stepped by a device-side ADAM during training) and
  // the variance is element D of the scales array
  if (variance < T(0)) variance = scales[D];
  __shared__ T xs[TILE][KM_DC + 1];
  __shared__ T ys[TILE][KM_DC + 1];
  const int tid = threadIdx.x;
  const int ty = tid >> 4, tx = tid & 15;
  const int64_t i0 = blockIdx.y * (int64_t)TILE, j0 = blockIdx.x * (int64_t)TILE;
  T acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = T(0);
  for (int64_t d0 = 0; d0 < D; d0 += KM_DC) {
    for (int e = tid; e < TILE * KM_DC; e += NTHREADS) {
      int r = e / KM_DC, d = e % KM_DC;
      int64_t gd = d0 + d;
      T sc = (scales && gd < D) ? scales[gd] : T(1);
      int64_t gi = i0 + r, gj = j0 + r;
      T xv = T(0), yv = T(0);
      if (gd < D) {
        if (gi < n) {
          int64_t src = idx ? idx[gi] : gi;
          xv = X[src * ldx + gd] * sc;
        }
        if (gj < p) yv = Y[gj * ldy + gd] * sc;
      }
      xs[r][d] = xv;
      ys[r][d] = yv;
    }
    __syncthreads();
#pragma unroll 8
    for (int d = 0; d < KM_DC; ++d) {
      T xv[4], yv[4];
#pragma unroll
      for (int a = 0; a < 4; ++a) xv[a] = xs[ty + 16 * a][d];
#pragma unroll
      for (int b = 0; b < 4; ++b) yv[b] = ys[tx + 16 * b][d];
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          T df = xv[a] - yv[b];
          acc[a][b] += df * df;
        }
    }
    __syncthreads();
  }
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    int64_t gi = i0 + ty + 16 * a;
    T rsum = T(0);
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      int64_t gj = j0 + tx + 16 * b;
      T val = T(0);
      if (gi < n && gj < p) {
        val = variance * kernel_base<T>(kind, acc[a][b]);
        if (sym && gi == gj) val += diag_add;
      } else if (sym && gi == gj) {
        val = T(1);
      }
      if (out && gi < n_out && gj < p_out) out[gi * ldo + gj] = val;
      if (alpha && gj < p) rsum += val * alpha[gj];
    }
    if (alpha) {
      rsum = row16_sum(rsum);
      if (tx == 0 && gi < n_out) part[blockIdx.x * ldp + gi] = rsum;
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// The same kernel matrix on the matrix cores (the default up to D = KMM_MAXD): squared distances in GEMM form
//     d2(i, j) = ||s.x_i||^2 + ||s.y_j||^2 - 2 (s.x_i).(s.y_j)
// with the cross term on v_mfma 16x16x4 (4 waves as 2 x 2, each a 32 x 32 sub-tile), operands staged ONCE per tile through LDS
// with 16-byte coalesced loads (a gathered minibatch row is one contiguous segment) and the scale folded in on the way.
// The GEMM form loses digits when two points (nearly) coincide -- K_ZZ's diagonal, inducing points picked from the data --
// which the first-order kernels (Exponential, Matern) would amplify through sqrt: every element with
// d2 < KMM_CLOSE (||x||^2 + ||y||^2) is recomputed by direct differences from the tiles still in LDS (rare, so the divergent
// branch costs little); everything else carries an absolute error of a few ulp(||x||^2 + ||y||^2) in d2.
// One workgroup = one 64-row tile x `ctiles` consecutive column tiles (grid.x = column groups).  With alpha, the fused row-dot
// sum_j out[i][j] alpha[j] is accumulated in registers over all column tiles of the group and leaves ONE partial slice per
// group (part[blockIdx.x][i]): streaming prediction runs with a single group -- K*m and per-tile partials never reach memory.
// ---------------------------------------------------------------------------------------------------
constexpr int KMM_MAXD = 128;

// Y side of k_kernelmatrix_mma, prepared once per (kernel, Y): Ysc[j][d] = s_d Y[j][d] (rows padded to a multiple of 64 and
// columns to Dp with zeros) and yn[j] = ||s . y_j||^2.  For the inducing points this is done at every K refresh, so the step
// and the streaming predictor copy ready-made tiles.  One wave per row.
template <typename T>
__global__ void k_scale_rows(const T* __restrict__ Y, int64_t ldy, int64_t p, int64_t p_pad, int64_t D, int Dp,
                             const T* __restrict__ scales, T* __restrict__ Ysc, T* __restrict__ yn) {
  const int64_t row = blockIdx.x * (int64_t)(blockDim.x >> 6) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= p_pad) return;
  T s = T(0);
  for (int d = lane; d < Dp; d += 64) {
    T v = T(0);
    if (row < p && d < D) v = Y[row * ldy + d] * (scales ? scales[d] : T(1));
    Ysc[row * Dp + d] = v;
    s += v * v;
  }
  for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o);
  if (lane == 0) yn[row] = s;
}

// KIND: the kernel function as a template parameter (K_SQEXP .. K_EXPONENTIAL).  With a run-time kind all four bodies were inlined
// into every one of the 16 values a lane finishes per tile: 223 VGPRs + 32 AGPRs, two waves per SIMD, and the accumulators were
// copied between AGPRs and VGPRs around every k-step.  With one body and __launch_bounds__(256, 4): 128 VGPRs, MFMA on VGPRs, four
// workgroups per CU (the LDS allows exactly that up to D = 32 in fp64) -- streaming prediction 4.8 -> 3.5 ms at C2 (a hand-written
// exp for non-positive arguments was also tried: no different from the library's).
// SPEC 1: row-dot only (streaming prediction: no matrix output, not symmetric); SPEC 2: store only (Knm of the step: not
// symmetric, no row-dot); SPEC 0: everything at run time.  The specialised forms drop the unused code and its registers.
// (SPEC 3 keeps the symmetric / diagonal logic and spilled 14 registers under the 128-VGPR cap; a K_ZZ launch is one workgroup per CU
//  anyway, so it is compiled for two.  Round 5: the same for the general form SPEC 0 -- set-up launches, the ABI's agp_kernelmatrix,
//  full predictive covariances --, whose fp64 instantiations carried 160 - 188 bytes of scratch per lane under that cap.
//  Round 6: the fp64 forms of the first-order kernels keep eight column slots per lane in the 4x4x4 accumulator layout and spilled 12
//  bytes under the 128-VGPR cap: three workgroups per CU for them)
template <typename T, int KIND, int SPEC = 0>
__global__ __launch_bounds__(NTHREADS, ((SPEC == 3 || SPEC == 0) ? 2 : (sizeof(T) == 8 && KIND != K_SQEXP) ? 3 : 4)) void k_kernelmatrix_mma(const T* __restrict__ X, int64_t ldx,
                                                               const int64_t* __restrict__ idx, int64_t n,
                                                               const T* __restrict__ Ysc, const T* __restrict__ yng,
                                                               int64_t p, int64_t D, int Dp, const T* __restrict__ scales,
                                                               T variance, T* __restrict__ out_, int64_t ldo,
                                                               int64_t n_out, int64_t p_out, int sym_, T diag_add,
                                                               const T* __restrict__ alpha_, T* __restrict__ part,
                                                               int64_t ldp, int64_t ctiles) {
  if (variance < T(0)) variance = scales[D];  // device-resident kernel parameters (see k_kernelmatrix)
  constexpr bool L4 = use_mfma4<T>();  // (AGP_F64_MFMA4 builds) fp64: 4x4x4 MFMA form and its accumulator layout (see the epilogue)
  T* __restrict__ out = SPEC == 1 ? nullptr : out_;
  const int sym = SPEC == 3 ? 1 : SPEC != 0 ? 0 : sym_;  // SPEC 3 (round 4): K_ZZ of a refresh -- store only, symmetric, no row-dot
  const T* __restrict__ alpha = (SPEC == 2 || SPEC == 3) ? nullptr : alpha_;
  extern __shared__ __attribute__((aligned(16))) unsigned char kmm_smem[];
  const int LDX = Dp + 2;  // 16 rows x {k, k+1} land on distinct banks (same stride rule as LDK in agp_device.h)
  T* Xs = reinterpret_cast<T*>(kmm_smem);  // [64][LDX]
  T* Ys = Xs + TILE * LDX;                 // [64][LDX]
  T* xn = Ys + TILE * LDX;                 // [64]
  T* yn = xn + TILE;                       // [64]
  T* sc = yn + TILE;                       // [Dp]
  T* red = sc + Dp;                        // [2][64] row-dot hand-over between the two column waves
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
  typedef typename Mfma<T>::vec_t vec_t;
  constexpr int VEC = Mfma<T>::VEC;
  const int nv = Dp / VEC;  // vectors per staged row
  const int64_t i0 = blockIdx.y * (int64_t)TILE;
  const bool vec_ok_x = (ldx % VEC) == 0 && (D % VEC) == 0 && ((uintptr_t)X % (sizeof(T) * VEC)) == 0;
  for (int d = tid; d < Dp; d += NTHREADS) sc[d] = d < D ? (scales ? scales[d] : T(1)) : T(0);
  __syncthreads();
  // X tile (gathered, scaled) -> LDS, then its squared norms
  for (int e = tid; e < TILE * nv; e += NTHREADS) {
    const int r = e / nv, dv = (e % nv) * VEC;
    const int64_t gr = i0 + r;
    T v[VEC];
#pragma unroll
    for (int q = 0; q < VEC; ++q) v[q] = T(0);
    if (gr < n) {
      const int64_t src = idx ? idx[gr] : gr;
      const T* row = X + src * ldx;
      if (vec_ok_x && dv + VEC <= D) {
        const vec_t x = *reinterpret_cast<const vec_t*>(row + dv);
#pragma unroll
        for (int q = 0; q < VEC; ++q) v[q] = x[q];
      } else {
#pragma unroll
        for (int q = 0; q < VEC; ++q)
          if (dv + q < D) v[q] = row[dv + q];
      }
    }
#pragma unroll
    for (int q = 0; q < VEC; ++q) Xs[r * LDX + dv + q] = v[q] * sc[dv + q];
  }
  __syncthreads();
  {  // squared norms: four lanes per row, combined in a fixed order
    const int r = tid >> 2, q4 = tid & 3;
    T s = T(0);
    for (int d = q4; d < Dp; d += 4) s += Xs[r * LDX + d] * Xs[r * LDX + d];
    s += __shfl_xor(s, 1);
    s += __shfl_xor(s, 2);
    if (q4 == 0) xn[r] = s;
  }
  const T close_thr = sizeof(T) == 8 ? T(1e-3) : T(3e-2);
  T rs[2][4];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int r = 0; r < 4; ++r) rs[mi][r] = T(0);
  const int cp_r0 = tid / nv, cp_v0 = tid % nv, cp_dr = NTHREADS / nv, cp_dv = NTHREADS % nv;
  const int64_t nct = (p_out + TILE - 1) / TILE;
  const int64_t ct0 = blockIdx.x * ctiles;
  const int64_t ct1 = (ct0 + ctiles < nct) ? ct0 + ctiles : nct;
  // column tiles: ready-made (pre-scaled, zero-padded) 64 x Dp blocks of Ysc, copied as they are.  No software pipelining here:
  // measured (r02), fetching tile t+1 into registers under tile t's products cost more than it hid -- the third LDS tile and the
  // prefetch registers take a workgroup per CU away, and four co-resident workgroups already overlap each other's loads
  // (C2 streaming prediction: 5.2 ms like this, 8.5 ms double-buffered, 8.0 ms with the VALU kernel)
  for (int64_t ct = ct0; ct < ct1; ++ct) {
    const int64_t j0 = ct * TILE;
    if (ct != ct0) __syncthreads();  // the previous tile's epilogue is done with Ys / yn
    {
      const T* src = Ysc + ct * TILE * Dp;  // contiguous 64 x Dp block
      // (row, vector) of element e = tid + 256 q advanced incrementally: two integer divisions per element and tile showed up as
      // ~300 VALU instructions per wave and tile in the PMC instruction counts
      int r = cp_r0, v = cp_v0;
      for (int e = tid; e < TILE * nv; e += NTHREADS) {
        const vec_t x = *reinterpret_cast<const vec_t*>(src + (int64_t)e * VEC);
        T* dst = Ys + r * LDX + v * VEC;
#pragma unroll
        for (int w = 0; w < VEC; ++w) dst[w] = x[w];
        r += cp_dr;
        v += cp_dv;
        if (v >= nv) {
          v -= nv;
          ++r;
        }
      }
      if (tid < TILE) yn[tid] = yng[ct * TILE + tid];
    }
    __syncthreads();
    const T* ynb = yn;
    typename Mfma<T>::acc_t acc[2][2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[mi][ni][r] = T(0);
    const T* pa = Xs + (wm * 32 + (lane & 15)) * LDX + (lane >> 4);
    const T* pb = Ys + (wn * 32 + (lane & 15)) * LDX + (lane >> 4);
#pragma unroll 4
    for (int kk = 0; kk < Dp / 4; ++kk) {
      const T a0 = pa[kk * 4], a1 = pa[16 * LDX + kk * 4];
      const T b0 = pb[kk * 4], b1 = pb[16 * LDX + kk * 4];
      if constexpr (L4) {  // fp64: the 4-block 4x4x4 MFMA (agp_device.h, mma_slab): same operand registers, B rotated block-wise
        const double b01 = dpp_ror<12>(b0), b02 = dpp_ror<8>(b0), b03 = dpp_ror<4>(b0);
        const double b11 = dpp_ror<12>(b1), b12 = dpp_ror<8>(b1), b13 = dpp_ror<4>(b1);
        acc[0][0] = mma4x4(a0, b0, b01, b02, b03, acc[0][0]);
        acc[0][1] = mma4x4(a0, b1, b11, b12, b13, acc[0][1]);
        acc[1][0] = mma4x4(a1, b0, b01, b02, b03, acc[1][0]);
        acc[1][1] = mma4x4(a1, b1, b11, b12, b13, acc[1][1]);
      } else {
        acc[0][0] = Mfma<T>::mma(a0, b0, acc[0][0]);
        acc[0][1] = Mfma<T>::mma(a0, b1, acc[0][1]);
        acc[1][0] = Mfma<T>::mma(a1, b0, acc[1][0]);
        acc[1][1] = Mfma<T>::mma(a1, b1, acc[1][1]);
      }
    }
    // epilogue.  Everything that depends on the column only or on the row only is taken out of the element loop: PMC showed ~96 VALU
    // instructions per kernel value with the bounds checks, the 64-bit index arithmetic and the alpha loads inside it -- the VALU,
    // not the MFMA or the exp, bounded the streaming predictor.
    // Accumulator layouts: element r of acc[mi][ni] is (row rowof(mi, r), column colof(ni, r)) of the 64 x 64 tile.  fp32 (16x16x4):
    // four rows, one column per lane; fp64 (4x4x4, round 6): ONE row, four columns per lane (NQ = 4 column slots per ni).
    constexpr int NQ = L4 ? 4 : 1;
    auto rowof = [&](int mi, int r) { return wm * 32 + mi * 16 + (L4 ? g4_row(lane) : Mfma<T>::row(lane, r)); };
    auto colof = [&](int ni, int q) { return wn * 32 + ni * 16 + (L4 ? g4_col(lane, q) : (lane & 15)); };
    int cl_[2][NQ];
    bool cok[2][NQ], cout_[2][NQ];
    T ynv[2][NQ], al[2][NQ];
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        cl_[ni][q] = colof(ni, q);
        const int64_t gj = j0 + cl_[ni][q];
        cok[ni][q] = gj < p;
        cout_[ni][q] = out != nullptr && gj < p_out;
        ynv[ni][q] = ynb[cl_[ni][q]];
        al[ni][q] = (alpha != nullptr && cok[ni][q]) ? alpha[gj] : T(0);
      }
    if constexpr (SPEC == 1 && KIND == K_SQEXP && sizeof(T) == 8) {
      // Streaming prediction with the squared-exponential kernel in fp64 (round 6): branch-free, 21 VALU instructions per kernel value
      // (s2, d2, two clamps, the 17 of exp_mhalf, the row-dot FMA) where the general form below spends ~48 and a branch.  No masks:
      // rows beyond n are zero rows of Xs (finite values, never stored), columns beyond p have alpha = 0.  No direct-difference
      // repair of (nearly) coincident points either: the GEMM form leaves d2 an absolute error of a few ulp(|x|^2 + |y|^2), which
      // this kernel function turns into HALF that as a relative error of its value -- it is the first-order kernels below (sqrt at
      // d2 -> 0) that need the repair.  The variance is folded into alpha.
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int q = 0; q < NQ; ++q) al[ni][q] *= variance;
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const T xnv = xn[rowof(mi, r)];  // (4x4x4 layout: the same row for every r)
          const int q = L4 ? r : 0;
#pragma unroll
          for (int ni = 0; ni < 2; ++ni) {
            const T d2 = __builtin_fma(T(-2), acc[mi][ni][r], xnv + ynv[ni][q]);
            rs[mi][L4 ? 0 : r] = __builtin_fma(exp_mhalf(d2), al[ni][q], rs[mi][L4 ? 0 : r]);
          }
        }
      continue;
    }
    const int64_t dgi = sym ? (j0 - i0) : (int64_t)1 << 40;  // gi == gj  <=>  rl - cl == j0 - i0 (only tiles on the diagonal can hit)
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int rl = rowof(mi, r);
        const int q = L4 ? r : 0;
        const int64_t gi = i0 + rl;
        const bool rok = gi < n, rout = gi < n_out;
        const T xnv = xn[rl];
        T* orow = out ? out + gi * ldo + j0 : nullptr;
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
          const int cl = cl_[ni][q];
          const bool diag = sym && (int64_t)(rl - cl) == dgi;
          T val = T(0);
          if (rok && cok[ni][q]) {
            const T s2 = xnv + ynv[ni][q];
            T d2 = s2 - T(2) * acc[mi][ni][r];
            if (diag) {
              // the point against itself: the direct differences below would give exactly 0 -- but through a 32-deep rolled loop that
              // every wave on a diagonal tile runs for half of its values: 21 us for K_ZZ at m = 1024 against 14 us for a K_nm of
              // the same size (round 4)
              d2 = T(0);
            } else if (d2 < close_thr * s2) {  // (nearly) coincident points: direct differences, no cancellation
              T t = T(0);
#pragma unroll 1
              for (int d = 0; d < Dp; ++d) {  // rare path: kept rolled (unrolled it cost registers on the common one)
                const T df = Xs[rl * LDX + d] - Ys[cl * LDX + d];
                t += df * df;
              }
              d2 = t;
            }
            val = variance * kernel_base<T>(KIND, d2);
            if (diag) val += diag_add;
          } else if (diag) {
            val = T(1);
          }
          if (cout_[ni][q] && rout) orow[cl] = val;
          rs[mi][L4 ? 0 : r] += val * al[ni][q];
        }
      }
  }
  if (alpha) {
    if constexpr (L4) {  // one row per lane and mi: the four lanes that share it differ in lane & 3
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
        T sv = rs[mi][0];
        sv += __shfl_xor(sv, 1);
        sv += __shfl_xor(sv, 2);
        if ((lane & 3) == 0) red[wn * TILE + wm * 32 + mi * 16 + g4_row(lane)] = sv;
      }
    } else {
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const T sv = row16_sum(rs[mi][r]);
          if ((lane & 15) == 0) red[wn * TILE + wm * 32 + mi * 16 + Mfma<T>::row(lane, r)] = sv;
        }
    }
    __syncthreads();
    if (tid < TILE && i0 + tid < n_out) part[blockIdx.x * ldp + i0 + tid] = red[tid] + red[TILE + tid];
  }
}

template <typename T>
inline size_t kmm_smem_bytes(int Dp) {
  return sizeof(T) * (size_t)(2 * TILE * (Dp + 2) + 2 * TILE + Dp + 2 * TILE);
}

// ---------------------------------------------------------------------------------------------------
// special functions
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ double digamma_d(double x) {
  // psi(x), x > 0 : upward recurrence to x >= 10, then the asymptotic series (|err| < 1e-15 relative)
  if (!(x > 0.0)) return __builtin_nan("");  // (same domain guard as lgamma_pos_d: the loop below would not end for x = -inf)
  double r = 0.0;
  while (x < 10.0) {
    r -= 1.0 / x;
    x += 1.0;
  }
  double f = 1.0 / (x * x);
  double t = f * (-1.0 / 12.0 +
                  f * (1.0 / 120.0 +
                       f * (-1.0 / 252.0 +
                            f * (1.0 / 240.0 + f * (-1.0 / 132.0 + f * (691.0 / 32760.0 + f * (-1.0 / 12.0)))))));
  return r + log(x) - 0.5 / x + t;
}

// log Gamma(x), x > 0: upward recurrence to x >= 10 (the product of the skipped factors, one log), then Stirling's series
// (|abs err| < 2e-15 there; the last term kept is 1/(156 x^13) < 7e-16).  The device library's lgamma needs more than the 128
// registers a 1024-thread workgroup has: the three ELBO kernels that call it per data point carried ~300 bytes of scratch per lane.
__device__ __forceinline__ double lgamma_pos_d(double x) {
  // domain guard: the recurrence below walks up to x >= 10 one step at a time -- x = -inf would never get there and a
  // hugely negative x (a bad count label handed through the C ABI) would keep the ELBO kernel's one workgroup, and its stream,
  // busy for |x| iterations.  Outside x > 0 (and for NaN) the answer is NaN, as the library's lgamma reports poles / bad input.
  if (!(x > 0.0) || x == __builtin_inf()) return x == __builtin_inf() ? x : __builtin_nan("");
  double p = 1.0;
  while (x < 10.0) {
    p *= x;
    x += 1.0;
  }
  const double f = 1.0 / (x * x);
  const double t = (1.0 / 12.0 +
                    f * (-1.0 / 360.0 +
                         f * (1.0 / 1260.0 +
                              f * (-1.0 / 1680.0 + f * (1.0 / 1188.0 + f * (-691.0 / 360360.0 + f * (1.0 / 156.0))))))) / x;
  return (x - 0.5) * log(x) - x + 0.91893853320467274178 + t - log(p);
}

// E[omega] for PG(1, c): tanh(c/2)/(2c), series 1/4 - c^2/48 near 0   (src/likelihood/logistic.jl:47-49)
template <typename T>
__device__ __forceinline__ T theta_pg(T c) {
  T ac = fabs(c);
  const T small = sizeof(T) == 8 ? T(1e-6) : T(1e-2);
  if (ac < small) return T(0.25) - c * c / T(48) + c * c * c * c / T(480);
  return tanh(c / T(2)) / (T(2) * c);
}

// log(cosh(x)) = log(exp(-2x)+1) + x - log 2   (src/functions/utils.jl:89-91)
__device__ __forceinline__ double logcosh_d(double x) { return log(exp(-2.0 * x) + 1.0) + x - 0.69314718055994530942; }

// exp(mu)/cosh(c) with the reference's overflow fallback (src/functions/utils.jl:84-86)
__device__ __forceinline__ double safe_expcosh_d(double mu, double c) {
  double r = exp(mu) / cosh(c);
  if (isfinite(r)) return r;
  double z = 2.0 * (mu > c ? mu : c);
  return 2.0 / (1.0 + exp(-z));
}

// ---------------------------------------------------------------------------------------------------
// Local step: finish K~, mean_f, var_f from the GEMM partial slices and run the likelihood's local_updates! +
// grad_E_mu / grad_E_Sigma (Appendix B of SURVEY.md):
//   K~   = kdiag + jitter - sum_s pk[s][i]          latentgp.jl:212
//   varf = sum_s pw0[s][i] + K~ ; muf = sum_s pw1[s][i]   latentgp.jl:179,189
//   r = rho * grad_E_mu , w = rho * grad_E_Sigma  (what the batch statistics consume, analyticVI.jl:168,179)
// For LogisticSoftMax only c_k is produced here; gamma/alpha/theta follow in the k_lsm_* kernels.
// ---------------------------------------------------------------------------------------------------
template <typename T>
struct LikParams {
  int kind;
  T p0;  // gaussian sigma2 / studentt nu / laplace beta / negbinomial r
  T p1;  // studentt sigma
  int noise_dev = 0;  // GaussianLikelihood(...; opt_noise=ADAM(eta)): sigma2 is STATE, stepped by every local update and read from the
                      // handle's likelihood-parameter word (lam[0], like the lambda of Poisson / Heteroscedastic) instead of p0
};

// Point-wise local update + expectation gradients of the single-latent likelihoods whose q(omega) needs no other state:
//   th = theta (E[omega] as the reference defines it), cc = c (or b for Laplace), g1 = grad_E_mu ; grad_E_Sigma = th/2
template <typename T>
__device__ __forceinline__ void lik_point(int kind, T p0, T p1, T m, T v, T yi, T& th, T& cc, T& g1) {
  cc = T(0);
  if (kind == LIK_GAUSSIAN) {  // gaussian.jl:70-80
    th = T(1) / p0;
    g1 = yi / p0;
  } else if (kind == LIK_LOGISTIC) {  // logistic.jl:39-51,64-69
    cc = sqrt(m * m + v);
    th = theta_pg<T>(cc);
    g1 = yi / T(2);
  } else if (kind == LIK_STUDENTT) {  // studentt.jl:68-82,96-99
    T alpha = (p0 + T(1)) / T(2);
    cc = ((m - yi) * (m - yi) + v + p1 * p1 * p0) / T(2);
    th = alpha / cc;
    g1 = th * yi;
  } else if (kind == LIK_LAPLACE) {  // laplace.jl:60-73,85-90 : b = sqrt(E[(f-y)^2]), theta = sqrt(a)/b, a = beta^-2
    cc = sqrt((m - yi) * (m - yi) + v);
    th = T(1) / (p0 * cc);
    g1 = th * yi;
  } else if (kind == LIK_BSVM) {  // bayesiansvm.jl:43-67
    T d = T(1) - yi * m;
    cc = d * d + v;
    th = T(1) / sqrt(cc);
    g1 = yi * (th + T(1));
  } else {  // LIK_NEGBIN negativebinomial.jl:69-99 : theta = (r + y) tanh(c/2)/c (no 1/2, as the reference writes it)
    cc = sqrt(m * m + v);
    th = (p0 + yi) * T(2) * theta_pg<T>(cc);
    g1 = (yi - p0) / T(2);
  }
}

// One wave per minibatch row i: row statistics of W (s0 = sum_j W_ij^2, s1 = sum_j W_ij v_j), the K~ slice sum, then
// lane 0 finishes K~ / mean_f / var_f and runs the likelihood update -- rowstats and local update in ONE launch.
// All latents of a handle in one launch: blockIdx.y = latent; the per-latent inputs come from the struct, the per-latent outputs
// sit `ostride` elements apart.
constexpr int ROWSTATS_MAXB = 16;
template <typename T>
struct RowstatsBatch {
  const T* pk[ROWSTATS_MAXB];  // K~ partial slices
  const T* W[ROWSTATS_MAXB];   // kappa L_A^-T
  const T* v[ROWSTATS_MAXB];   // L_A^-1 eta1
  T kdiag[ROWSTATS_MAXB];      // kernel variance (diagonal of the kernel matrix); < 0: read it from kd_ptr (device-resident)
  const T* kd_ptr[ROWSTATS_MAXB];
  int use_kt[ROWSTATS_MAXB];   // K~ kept from the previous full-batch step
};
// what one thread does for row i once the three row sums are known (s0 = sum_j W_ij^2, s1 = sum_j W_ij v_j, sk = sum of the K~
// slices): K~ / mean_f / var_f and the likelihood's local update + expectation gradients.  Shared by the row-statistics kernels and
// by the epilogue of the CAVI step's task-graph launch (agp_chol.h, round 3).  The output pointers are those of the row's latent.
template <typename T>
__device__ __forceinline__ void rowstats_finish(int64_t i, T s0, T s1, T sk, T kdiag, int use_kt, T jitter, T rho,
                                                const LikParams<T>& lp, const T* __restrict__ y,
                                                const int64_t* __restrict__ idx, T* __restrict__ Kt, T* __restrict__ muf,
                                                T* __restrict__ varf, T* __restrict__ c, T* __restrict__ theta, T* __restrict__ r,
                                                T* __restrict__ w, int* __restrict__ flags, const T* __restrict__ lam,
                                                T* __restrict__ gamma, const T* yi_pre = nullptr) {
  // (yi_pre: the row's target, already fetched by the caller)
  // use_kt: kappa (hence K~) was kept from the previous full-batch step (training.jl:199-205)
  T kt = use_kt ? Kt[i] : kdiag + jitter - sk;
  if (!(kt > T(0))) atomicOr(flags, FLAG_NEG_KTILDE);
  T var = s0 + kt, mu = s1;
  Kt[i] = kt;
  muf[i] = mu;
  varf[i] = var;
  if (lp.kind == LIK_LSM) {
    c[i] = sqrt(mu * mu + var);  // logisticsoftmax.jl:62-64
    return;
  }
  if (lp.kind == LIK_MO || lp.kind == LIK_HETERO) return;  // mixing / two-latent coupling follow in k_mo_local, k_hetero_*
  if (lp.kind == LIK_GAUSSIAN && lp.noise_dev) return;     // the noise step comes first (k_noise_*), then theta / gradients
  T yi = yi_pre ? *yi_pre : y[idx ? idx[i] : i];
  T th, cc, g1;
  if (lp.kind == LIK_POISSON) {  // poisson.jl:64-80,94-103 (lambda itself is re-estimated afterwards by k_poisson_*)
    cc = sqrt(mu * mu + var);
    T gam = (T)((double)lam[0] * safe_expcosh_d(-0.5 * (double)mu, 0.5 * (double)cc) / 2.0);
    th = (yi + gam) * T(2) * theta_pg<T>(cc);
    g1 = (yi - gam) / T(2);
    gamma[i] = gam;
  } else {
    lik_point<T>(lp.kind, lp.p0, lp.p1, mu, var, yi, th, cc, g1);
  }
  c[i] = cc;
  theta[i] = th;
  r[i] = rho * g1;
  w[i] = rho * th / T(2);
}

// one wave's work on row i of latent q (shared by k_rowstats_local and the kernel that also carries the task graph's fallback)
template <typename T>
__device__ __forceinline__ void rowstats_row(int64_t i, int lane, int q, int64_t B, int nslices, const RowstatsBatch<T>& rb,
                                             int64_t ldp, int64_t ldw, int64_t cols, T jitter, T rho, const LikParams<T>& lp,
                                             const T* __restrict__ y, const int64_t* __restrict__ idx, T* __restrict__ Kt,
                                             T* __restrict__ muf, T* __restrict__ varf, T* __restrict__ c,
                                             T* __restrict__ theta, T* __restrict__ r, T* __restrict__ w, int64_t ostride,
                                             int* __restrict__ flags, const T* __restrict__ lam, T* __restrict__ gamma) {
  const T* __restrict__ pk = rb.pk[q];
  const T* __restrict__ W = rb.W[q];
  const T* __restrict__ v = rb.v[q];
  const T kdiag = rb.kdiag[q] < T(0) ? rb.kd_ptr[q][0] : rb.kdiag[q];
  const int use_kt = rb.use_kt[q];
  Kt += q * ostride;
  muf += q * ostride;
  varf += q * ostride;
  c += q * ostride;
  theta += q * ostride;
  r += q * ostride;
  w += q * ostride;
  gamma += q * ostride;
  if (i >= B) return;
  T s0 = T(0), s1 = T(0), sk = T(0);
  {
    // 16-byte loads, four independent accumulator pairs (cols is a multiple of 64 and the rows are 512-byte aligned: padded layout)
    typedef T T2 __attribute__((ext_vector_type(2)));
    const T2* Wr = reinterpret_cast<const T2*>(W + i * ldw);
    const T2* vr = reinterpret_cast<const T2*>(v);
    T a0 = T(0), a1 = T(0), b0 = T(0), b1 = T(0);
    const int64_t n2 = cols >> 1;
    int64_t j = lane;
    for (; j + 64 < n2; j += 128) {
      const T2 w0 = Wr[j], w1 = Wr[j + 64], v0 = vr[j], v1 = vr[j + 64];
      a0 += w0.x * w0.x + w0.y * w0.y;
      b0 += w0.x * v0.x + w0.y * v0.y;
      a1 += w1.x * w1.x + w1.y * w1.y;
      b1 += w1.x * v1.x + w1.y * v1.y;
    }
    for (; j < n2; j += 64) {
      const T2 w0 = Wr[j], v0 = vr[j];
      a0 += w0.x * w0.x + w0.y * w0.y;
      b0 += w0.x * v0.x + w0.y * v0.y;
    }
    s0 = a0 + a1;
    s1 = b0 + b1;
  }
  if (!use_kt)
    for (int s = lane; s < nslices; s += 64) sk += pk[s * ldp + i];
  for (int o = 32; o > 0; o >>= 1) {
    s0 += __shfl_down(s0, o);
    s1 += __shfl_down(s1, o);
    sk += __shfl_down(sk, o);
  }
  if (lane != 0) return;
  rowstats_finish<T>(i, s0, s1, sk, kdiag, use_kt, jitter, rho, lp, y, idx, Kt, muf, varf, c, theta, r, w, flags, lam, gamma);
}

template <typename T>
__global__ void k_rowstats_local(int64_t B, int nslices, RowstatsBatch<T> rb, int64_t ldp, int64_t ldw, int64_t cols,
                                 T jitter, T rho, LikParams<T> lp, const T* __restrict__ y,
                                 const int64_t* __restrict__ idx, T* __restrict__ Kt, T* __restrict__ muf,
                                 T* __restrict__ varf, T* __restrict__ c, T* __restrict__ theta, T* __restrict__ r,
                                 T* __restrict__ w, int64_t ostride, int* __restrict__ flags, const T* __restrict__ lam,
                                 T* __restrict__ gamma) {
  rowstats_row<T>(blockIdx.x * (int64_t)(blockDim.x >> 6) + (threadIdx.x >> 6), threadIdx.x & 63, (int)blockIdx.y, B, nslices, rb,
                  ldp, ldw, cols, jitter, rho, lp, y, idx, Kt, muf, varf, c, theta, r, w, ostride, flags, lam, gamma);
}

// Poisson: lambda <- sum(y) / sum_i E_{N(mu_i, var_i)}[logistic]  (poisson.jl:78 ; expectation = Gauss-Hermite, utils.jl:16-19)
// stage 1: per-block partial sums part[2*b] = sum y, part[2*b+1] = sum E[sigma(f)]
template <typename T>
__global__ void k_poisson_partial(int64_t B, const T* __restrict__ y, const int64_t* __restrict__ idx,
                                  const T* __restrict__ muf, const T* __restrict__ varf, int nn,
                                  const double* __restrict__ nodes, const double* __restrict__ weights,
                                  double* __restrict__ part) {
  __shared__ double red[16];
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  double sy = 0.0, se = 0.0;
  if (i < B) {
    sy = (double)y[idx ? idx[i] : i];
    double m = (double)muf[i], sd = sqrt(fmax((double)varf[i], 0.0));
    for (int q = 0; q < nn; ++q) {
      double x = nodes[q] * sd + m;
      se += weights[q] * (x >= 0.0 ? 1.0 / (1.0 + exp(-x)) : exp(x) / (1.0 + exp(x)));
    }
  }
  sy = block_sum<double>(sy, red);
  se = block_sum<double>(se, red);
  if (threadIdx.x == 0) {
    part[2 * blockIdx.x] = sy;
    part[2 * blockIdx.x + 1] = se;
  }
}

// stage 2 (one workgroup): mode 0 Poisson  lam = S0/S1 ; mode 1 heteroscedastic lam = max(B / (2 S0), lam)
template <typename T>
__global__ void k_lambda_finish(int nparts, int stride, const double* __restrict__ part, int mode, double Bn,
                                T* __restrict__ lam) {
  __shared__ double red[16];
  double s0 = 0.0, s1 = 0.0;
  for (int b = threadIdx.x; b < nparts; b += blockDim.x) {
    s0 += part[stride * b];
    if (stride > 1) s1 += part[stride * b + 1];
  }
  s0 = block_sum<double>(s0, red);
  s1 = block_sum<double>(s1, red);
  if (threadIdx.x == 0) {
    if (mode == 0) {
      lam[0] = (T)(s0 / s1);
    } else {
      double cand = Bn / (2.0 * s0), cur = (double)lam[0];
      lam[0] = (T)(cand > cur ? cand : cur);
    }
  }
}

// GaussianLikelihood(sigma2; opt_noise = ADAM(eta))  (gaussian.jl:18-23, 56-72): before theta = 1 / sigma2 is refreshed, every
// local update takes one ADAM ascent step on log sigma2 with
//     grad = ((sum_i (y_i - mu_i)^2 + sum_i var_f,i) / sigma2 - B) / 2          (= d E_q[log p(y|f)] / d log sigma2)
//     sigma2 <- exp(log sigma2 + ADAM(grad))
// stage 1: per-block partial sums of (y - mu)^2 + var_f ; stage 2 (one workgroup): the sum in block order, the ADAM step on the
// state adam = [m, v, t] (doubles) and the new sigma2 ; then theta, r = rho y / sigma2, w = rho theta / 2 with the NEW sigma2.
template <typename T>
__global__ void k_noise_partial(int64_t B, const T* __restrict__ y, const int64_t* __restrict__ idx, const T* __restrict__ muf,
                                const T* __restrict__ varf, double* __restrict__ part) {
  __shared__ double red[16];
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  double s = 0.0;
  if (i < B) {
    const double d = (double)y[idx ? idx[i] : i] - (double)muf[i];
    s = d * d + (double)varf[i];
  }
  s = block_sum<double>(s, red);
  if (threadIdx.x == 0) part[blockIdx.x] = s;
}
// S = sum of part[0 .. nparts) ; Bn_ptr (nullable): the batch size as a device double (after an all-reduce over the shards)
template <typename T>
__global__ void k_noise_finish(int nparts, const double* __restrict__ part, double Bn, const double* __restrict__ Bn_ptr, double eta,
                               double b1, double b2, double eps, double* __restrict__ adam, T* __restrict__ sigma2) {
  __shared__ double red[16];
  double s = 0.0;
  for (int b = threadIdx.x; b < nparts; b += blockDim.x) s += part[b];
  s = block_sum<double>(s, red);
  if (threadIdx.x != 0) return;
  const double s2 = (double)sigma2[0], n = Bn_ptr ? Bn_ptr[0] : Bn;
  const double g = (s / s2 - n) / 2.0;
  const double t = adam[2] + 1.0;
  const double m = b1 * adam[0] + (1.0 - b1) * g, v = b2 * adam[1] + (1.0 - b2) * g * g;
  adam[0] = m;
  adam[1] = v;
  adam[2] = t;
  const double mh = m / (1.0 - pow(b1, t)), vh = v / (1.0 - pow(b2, t));
  sigma2[0] = (T)exp(log(s2) + eta * mh / (sqrt(vh) + eps));
}
template <typename T>
__global__ void k_gauss_grads(int64_t B, T rho, const T* __restrict__ y, const int64_t* __restrict__ idx,
                              const T* __restrict__ sigma2, T* __restrict__ theta, T* __restrict__ c, T* __restrict__ r,
                              T* __restrict__ w) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= B) return;
  const T th = T(1) / sigma2[0];
  theta[i] = th;
  c[i] = T(0);
  r[i] = rho * y[idx ? idx[i] : i] * th;
  w[i] = rho * th / T(2);
}

// batch-sharded handles (round 3): the two sums and the batch size leave stage 1 as three doubles red3 = [S0, S1, B_local], are
// all-reduced over the ranks, and stage 2 finishes from the reduced values -- poisson.jl:78 / heteroscedastic.jl:94 over the WHOLE
// minibatch, not the shard
__global__ void k_lambda_reduce(int nparts, int stride, const double* __restrict__ part, double Bn, double* __restrict__ red3) {
  __shared__ double red[16];
  double s0 = 0.0, s1 = 0.0;
  for (int b = threadIdx.x; b < nparts; b += blockDim.x) {
    s0 += part[stride * b];
    if (stride > 1) s1 += part[stride * b + 1];
  }
  s0 = block_sum<double>(s0, red);
  s1 = block_sum<double>(s1, red);
  if (threadIdx.x == 0) {
    red3[0] = s0;
    red3[1] = s1;
    red3[2] = Bn;
  }
}
template <typename T>
__global__ void k_lambda_finish_red(const double* __restrict__ red3, int mode, T* __restrict__ lam) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  if (mode == 0) {
    lam[0] = (T)(red3[0] / red3[1]);
  } else {
    const double cand = red3[2] / (2.0 * red3[0]), cur = (double)lam[0];
    lam[0] = (T)(cand > cur ? cand : cur);
  }
}

// Heteroscedastic Gaussian (heteroscedastic.jl:71-97): latent 0 = f, latent 1 = g ; arrays are [2][ldb]:
//   c[0] = phi = E[(f-y)^2]/2 ; c[1] = c = sqrt(E[g^2]) ; gamma[1] = sigg ~ E[sigma(-g)] ; gamma[0] = gamma = lam phi sigg ;
//   theta[1] = (1/2 + gamma) tanh(c/2)/(2c) ; part[b] = sum_block phi (1 - sigg)
template <typename T>
__global__ void k_hetero_local(int64_t B, int64_t ldb, const T* __restrict__ y, const int64_t* __restrict__ idx,
                               const T* __restrict__ muf, const T* __restrict__ varf, const T* __restrict__ lam,
                               T* __restrict__ c, T* __restrict__ gamma, T* __restrict__ theta,
                               double* __restrict__ part) {
  __shared__ double red[16];
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  double contrib = 0.0;
  if (i < B) {
    T yi = y[idx ? idx[i] : i];
    T m0 = muf[i], v0 = varf[i], m1 = muf[ldb + i], v1 = varf[ldb + i];
    T phi = ((m0 - yi) * (m0 - yi) + v0) / T(2);
    T cc = sqrt(m1 * m1 + v1);
    T sg = (T)(safe_expcosh_d(-0.5 * (double)m1, 0.5 * (double)cc) / 2.0);
    T gam = lam[0] * phi * sg;
    c[i] = phi;
    c[ldb + i] = cc;
    gamma[ldb + i] = sg;
    gamma[i] = gam;
    theta[ldb + i] = (T(0.5) + gam) * theta_pg<T>(cc);
    contrib = (double)phi * (1.0 - (double)sg);
  }
  contrib = block_sum<double>(contrib, red);
  if (threadIdx.x == 0) part[blockIdx.x] = contrib;
}

// grad_E_mu / grad_E_Sigma with the UPDATED lambda (heteroscedastic.jl:113-129): theta[0] keeps lam*sigg so that
// grad_E_Sigma = theta/2 holds for both latents
template <typename T>
__global__ void k_hetero_grads(int64_t B, int64_t ldb, T rho, const T* __restrict__ y, const int64_t* __restrict__ idx,
                               const T* __restrict__ lam, const T* __restrict__ gamma, T* __restrict__ theta,
                               T* __restrict__ r, T* __restrict__ w) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= B) return;
  T yi = y[idx ? idx[i] : i];
  T ls = lam[0] * gamma[ldb + i];
  theta[i] = ls;
  r[i] = rho * yi * ls / T(2);
  w[i] = rho * ls / T(2);
  r[ldb + i] = rho * (T(0.5) - gamma[i]) / T(2);
  w[ldb + i] = rho * theta[ldb + i] / T(2);
}

// extension block of the augmented Cholesky: row 0 = eta1', rows 1..63 = 0
template <typename T>
__global__ void k_set_ext_rows(T* __restrict__ ext, int64_t ld, int64_t n, const T* __restrict__ eta1) {
  int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (e >= 64 * n) return;
  int64_t rr = e / n, cc2 = e % n;
  ext[rr * ld + cc2] = (rr == 0) ? eta1[cc2] : T(0);
}

// LogisticSoftMax fixed point (logisticsoftmax.jl:65-72), arrays are [nl][ldb] per local latent:
//   gamma_k = exp(psi(alpha)) * safe_expcosh(-mu_k/2, c_k/2) / (2 beta) ; gsum = sum_k gamma_k (local latents)
template <typename T>
__global__ void k_lsm_gamma(int64_t B, int nl, int64_t ldb, const T* __restrict__ muf, const T* __restrict__ c,
                            const T* __restrict__ alpha, const T* __restrict__ beta, T* __restrict__ gamma,
                            T* __restrict__ gsum) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= B) return;
  double epsi = exp(digamma_d((double)alpha[i]));
  double b2 = 2.0 * (double)beta[i];
  double s = 0.0;
  for (int k = 0; k < nl; ++k) {
    double g = epsi * safe_expcosh_d(-0.5 * (double)muf[k * ldb + i], 0.5 * (double)c[k * ldb + i]) / b2;
    gamma[k * ldb + i] = (T)g;
    s += g;
  }
  gsum[i] = (T)s;
}

template <typename T>
__global__ void k_lsm_alpha(int64_t B, const T* __restrict__ gsum, T* __restrict__ alpha) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < B) alpha[i] = T(1) + gsum[i];
}

// theta_k = (Y_k + gamma_k) tanh(c_k/2)/(2 c_k) ; r_k = rho (Y_k - gamma_k)/2 ; w_k = rho theta_k/2
// (logisticsoftmax.jl:73-77, 98-103) ; ycls = 0-based class index, Y_k = (ycls == latent_offset + k)
template <typename T>
__global__ void k_lsm_finish(int64_t B, int nl, int64_t ldb, int latent_offset, T rho,
                             const int32_t* __restrict__ ycls, const int64_t* __restrict__ idx,
                             const T* __restrict__ c, const T* __restrict__ gamma, T* __restrict__ theta,
                             T* __restrict__ r, T* __restrict__ w, int n_class, int* __restrict__ flags) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= B) return;
  int cls = ycls[idx ? idx[i] : i];
  // a class index outside the likelihood's K classes has no one-hot row (multiclass.jl:81-83 throws on it)
  if (cls < 0 || cls >= n_class) atomicOr(flags, FLAG_BAD_LABEL);
  for (int k = 0; k < nl; ++k) {
    T yk = (cls == latent_offset + k) ? T(1) : T(0);
    T g = gamma[k * ldb + i];
    T th = (yk + g) * theta_pg<T>(c[k * ldb + i]);
    theta[k * ldb + i] = th;
    r[k * ldb + i] = rho * (yk - g) / T(2);
    w[k * ldb + i] = rho * th / T(2);
  }
}

// The whole local update of a handle that holds ALL latents of the model -- two rounds of (gamma, alpha) and the final theta, r, w
// (k_lsm_gamma, k_lsm_alpha, k_lsm_gamma, k_lsm_alpha, k_lsm_finish) -- in one launch: the fixed point is per data point, and five
// launches of four workgroups each cost the 8-class step ~55 us of an in-order queue.
// Round 5: one LANE per (point, latent) -- LG = the power of two >= nl consecutive lanes own one point -- instead of one lane per
// point walking its latents out of per-thread arrays (196 - 668 bytes of scratch per lane, four workgroups for the whole C4 update,
// 41 us): every lane evaluates its own gamma_k, the sum over the latents is taken with shuffles IN LATENT ORDER (k = 0, 1, ...), so
// it is the sum k_lsm_gamma forms -- the same operations in the same order per point and latent as the separate kernels (which the
// latent-parallel driver keeps: its sum over latents crosses ranks): bitwise the same results.  psi(alpha) is evaluated by every
// lane of a point (same argument, same value).
constexpr int LSM_FUSED_MAXL = 16;
template <typename T, int LG>
__global__ void __launch_bounds__(256)
k_lsm_fused(int64_t B, int nl, int64_t ldb, int latent_offset, T rho, const int32_t* __restrict__ ycls,
            const int64_t* __restrict__ idx, const T* __restrict__ muf, const T* __restrict__ c, T* __restrict__ alpha,
            const T* __restrict__ beta, T* __restrict__ gamma, T* __restrict__ gsum, T* __restrict__ theta, T* __restrict__ r,
            T* __restrict__ w, int n_class, int* __restrict__ flags) {
  static_assert(LG >= 1 && LG <= LSM_FUSED_MAXL && (LG & (LG - 1)) == 0, "lanes per point: a power of two <= 16");
  const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  const int64_t i_raw = t / LG;
  const int k = (int)(t % LG);
  const bool pt = i_raw < B;            // every lane stays in the shuffles; lanes beyond B / nl compute on clamped operands
  const bool on = pt && k < nl;
  const int64_t i = pt ? i_raw : B - 1;
  const int kk = k < nl ? k : nl - 1;
  const int lane = threadIdx.x & 63, base = lane & ~(LG - 1);
  const T cc = c[kk * ldb + i];
  const double hm = -0.5 * (double)muf[kk * ldb + i], hc = 0.5 * (double)cc;
  T a = alpha[i], gs = T(0), gam = T(0);
  const double b2 = 2.0 * (double)beta[i];
  for (int it = 0; it < 2; ++it) {  // logisticsoftmax.jl:65
    const double epsi = exp(digamma_d((double)a));
    const double g = epsi * safe_expcosh_d(hm, hc) / b2;
    gam = (T)g;
    double s = 0.0;
    for (int q = 0; q < nl; ++q) s += __shfl(g, base + q);  // latent order: the sum of k_lsm_gamma
    gs = (T)s;
    a = T(1) + gs;
  }
  if (!on) return;
  const int cls = ycls[idx ? idx[i] : i];
  if (k == 0) {
    alpha[i] = a;
    gsum[i] = gs;
    if (cls < 0 || cls >= n_class) atomicOr(flags, FLAG_BAD_LABEL);
  }
  const T yk = (cls == latent_offset + k) ? T(1) : T(0);
  const T th = (yk + gam) * theta_pg<T>(cc);
  gamma[k * ldb + i] = gam;
  theta[k * ldb + i] = th;
  r[k * ldb + i] = rho * (yk - gam) / T(2);
  w[k * ldb + i] = rho * th / T(2);
}

// ---------------------------------------------------------------------------------------------------
// Multi-output mixing (src/models/single_and_multi_output_utils.jl:24-118, MOSVGP): task t sees f_t = sum_q A[t][q] f_q.
// ---------------------------------------------------------------------------------------------------
constexpr int MO_MAXT = 16;
template <typename T>
struct MoCfg {
  int nT;
  int kind[MO_MAXT];
  T p0[MO_MAXT];
  T p1[MO_MAXT];
};

// grad_E_mu from the stored theta (the local variables of a previous step)
template <typename T>
__device__ __forceinline__ T lik_g1(int kind, T p0, T yi, T thv) {
  if (kind == LIK_GAUSSIAN) return yi / p0;
  if (kind == LIK_LOGISTIC) return yi / T(2);
  if (kind == LIK_BSVM) return yi * (thv + T(1));
  if (kind == LIK_NEGBIN) return (yi - p0) / T(2);
  return thv * yi;  // StudentT, Laplace
}

// mixed mean_f / var_f (lines 24-45), per-task local updates, mixed gradients (lines 48-84):
//   r_q = rho sum_t A_tq (g1_t - 2 g2_t (m_t - A_tq mu_q)) ; w_q = rho sum_t A_tq^2 g2_t
template <typename T>
__global__ void __launch_bounds__(256)  // (launched with 256 threads; the default 1024-thread budget of 128 registers spilled 716 B per lane in fp32)
k_mo_local(int64_t B, int Q, int64_t ldb, MoCfg<T> cfg, const T* __restrict__ A, T rho,
                           const T* __restrict__ y, int64_t ystride, const int64_t* __restrict__ idx,
                           const T* __restrict__ muf, const T* __restrict__ varf, T* __restrict__ mixm,
                           T* __restrict__ mixv, T* __restrict__ th, T* __restrict__ cc, T* __restrict__ r,
                           T* __restrict__ w, int do_local, int q_lo, int q_n) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= B) return;
  T gm[MO_MAXT], gs[MO_MAXT], mt[MO_MAXT];
  const int64_t src = idx ? idx[i] : i;
#pragma unroll
  for (int t = 0; t < MO_MAXT; ++t) {
    gm[t] = gs[t] = mt[t] = T(0);
    if (t < cfg.nT) {
      T m = T(0), v = T(0);
      for (int q = 0; q < Q; ++q) {
        T a = A[t * Q + q];
        m += a * muf[q * ldb + i];
        v += a * a * varf[q * ldb + i];
      }
      mixm[t * ldb + i] = m;
      mixv[t * ldb + i] = v;
      mt[t] = m;
      if (do_local) {
        T thv, cv, g1;
        lik_point<T>(cfg.kind[t], cfg.p0[t], cfg.p1[t], m, v, y[src * ystride + t], thv, cv, g1);
        th[t * ldb + i] = thv;
        cc[t * ldb + i] = cv;
        gm[t] = g1;
        gs[t] = thv / T(2);
      }
    }
  }
  if (!do_local) return;
  // gradients of the latents this handle owns (all of them, or the slice [q_lo, q_lo + q_n) of a latent-sharded model)
  for (int ql = 0; ql < q_n; ++ql) {
    const int q = q_lo + ql;
    T mq = muf[q * ldb + i], g1 = T(0), g2 = T(0);
#pragma unroll
    for (int t = 0; t < MO_MAXT; ++t) {
      if (t < cfg.nT) {
        T a = A[t * Q + q];
        g1 += a * (gm[t] - T(2) * gs[t] * (mt[t] - a * mq));
        g2 += a * a * gs[t];
      }
    }
    r[ql * ldb + i] = rho * g1;
    w[ql * ldb + i] = rho * g2;
  }
}

// hyper-gradient inputs of latent l in the multi-output model: the data term sees f_l only through the mixed f_t, so
//   dE/dmu_l = sum_t A_tl (g1_t - theta_t m_t) ,  dE/dsigma2_l = -sum_t A_tl^2 theta_t / 2
// (m_t = mixed mean under the CURRENT posterior; theta_t, g1_t from the step's local variables; elbo_ref keeps the
// reference's dot(theta, mu) / BayesianSVM expressions per task like k_hyper_gvec)
template <typename T>
__global__ void k_mo_hyper_gvec(int64_t B, int Q, int64_t ldb, MoCfg<T> cfg, const T* __restrict__ A,
                                const T* __restrict__ y, int64_t ystride, const int64_t* __restrict__ idx,
                                const T* __restrict__ muf, const T* __restrict__ th, int l, int elbo_ref,
                                T* __restrict__ gmu, T* __restrict__ gs) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= B) return;
  const int64_t src = idx ? idx[i] : i;
  T a1 = T(0), a2 = T(0);
  for (int t = 0; t < cfg.nT; ++t) {
    T m = T(0);
    for (int q = 0; q < Q; ++q) m += A[t * Q + q] * muf[q * ldb + i];
    const T thv = th[t * ldb + i], yi = y[src * ystride + t], atl = A[t * Q + l];
    const int kind = cfg.kind[t];
    T g;
    if (elbo_ref && (kind == LIK_LOGISTIC || kind == LIK_NEGBIN)) g = lik_g1<T>(kind, cfg.p0[t], yi, thv) - thv / T(2);
    else if (elbo_ref && kind == LIK_BSVM) g = yi - T(2) * thv * (T(1) - yi * m) * yi;
    else g = lik_g1<T>(kind, cfg.p0[t], yi, thv) - thv * m;
    a1 += atl * g;
    a2 += atl * atl * thv;
  }
  gmu[i] = a1;
  gs[i] = -a2 / T(2);
}

// update_A! gradient (lines 87-109) with the local variables of the PREVIOUS step: grid (Q, nT), one workgroup each
template <typename T>
__global__ void k_mo_gradA(int64_t B, int Q, int64_t ldb, MoCfg<T> cfg, const T* __restrict__ A,
                           const T* __restrict__ y, int64_t ystride, const int64_t* __restrict__ idx,
                           const T* __restrict__ muf, const T* __restrict__ varf, const T* __restrict__ th,
                           double* __restrict__ gradA) {
  __shared__ double red[16];
  const int q = blockIdx.x, t = blockIdx.y;
  const T atq = A[t * Q + q];
  double x1 = 0.0, x2 = 0.0;
  for (int64_t i = threadIdx.x; i < B; i += blockDim.x) {
    T thv = th[t * ldb + i];
    T yi = y[(idx ? idx[i] : i) * ystride + t];
    T g1 = lik_g1<T>(cfg.kind[t], cfg.p0[t], yi, thv);
    T g2 = thv / T(2);
    T m = T(0);
    for (int qq = 0; qq < Q; ++qq) m += A[t * Q + qq] * muf[qq * ldb + i];
    T mq = muf[q * ldb + i];
    x1 += (double)(g1 * mq - T(2) * g2 * mq * (m - atq * mq));
    x2 += (double)(g2 * (mq * mq + varf[q * ldb + i]));
  }
  x1 = block_sum<double>(x1, red);
  x2 = block_sum<double>(x2, red);
  if (threadIdx.x == 0) gradA[t * Q + q] = x1 - 2.0 * (double)atq * x2;
}

// ADAM ascent step on each row of A followed by the projection on the unit sphere (lines 110-112); one thread per task
template <typename T>
__global__ void k_mo_applyA(int nT, int Q, T* __restrict__ A, const double* __restrict__ gradA, double* __restrict__ am,
                            double* __restrict__ av, int step, double eta, double b1, double b2, double eps) {
  int t = threadIdx.x;
  if (t >= nT) return;
  const double c1 = 1.0 - pow(b1, (double)step), c2 = 1.0 - pow(b2, (double)step);
  double nrm = 0.0;
  for (int q = 0; q < Q; ++q) {
    double g = gradA[t * Q + q];
    double m = b1 * am[t * Q + q] + (1.0 - b1) * g;
    double v = b2 * av[t * Q + q] + (1.0 - b2) * g * g;
    am[t * Q + q] = m;
    av[t * Q + q] = v;
    double a = (double)A[t * Q + q] + eta * (m / c1) / (sqrt(v / c2) + eps);
    A[t * Q + q] = (T)a;
    nrm += a * a;
  }
  nrm = sqrt(nrm);
  for (int q = 0; q < Q; ++q) A[t * Q + q] = (T)((double)A[t * Q + q] / nrm);
}

// out[t][i] = sum_q A[t][q]^p in[q][i]   (p = 1 means, p = 2 variances)   predictions.jl:60-64,75-79
template <typename T>
__global__ void k_mo_mix(int64_t n, int Q, int nT, const T* __restrict__ A, int lda, const T* __restrict__ in,
                         int64_t ldi, T* __restrict__ out, int64_t ldo, int square) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  for (int t = 0; t < nT; ++t) {
    T s = T(0);
    for (int q = 0; q < Q; ++q) {
      T a = A[t * lda + q];
      s += (square ? a * a : a) * in[q * ldi + i];
    }
    out[t * ldo + i] = s;
  }
}

// mean_f / var_f only (ELBO with the updated posterior): mu = sum pw1 ; var = sum pw0 + K~
template <typename T>
__global__ void k_meanvar_finish(int64_t B, int nslices, const T* __restrict__ pw0, const T* __restrict__ pw1,
                                 int64_t ldp, const T* __restrict__ Kt, T* __restrict__ muf, T* __restrict__ varf) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= B) return;
  T s0 = T(0), s1 = T(0);
  for (int s = 0; s < nslices; ++s) {
    s0 += pw0[s * ldp + i];
    s1 += pw1[s * ldp + i];
  }
  muf[i] = s1;
  varf[i] = s0 + Kt[i];
}

template <typename T>
__global__ void k_zero_strict_upper(T* __restrict__ A, int64_t ld, int64_t n) {
  int64_t i = blockIdx.y * (int64_t)blockDim.y + threadIdx.y;
  int64_t j = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < n && j < n && j > i) A[i * ld + j] = T(0);
}

template <typename T>
__global__ void k_add_diag(T* __restrict__ A, int64_t ld, int64_t n, T v) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < n) A[i * ld + i] += v;
}

// dst (rd x cd) = src (rs x cs) zero-padded
template <typename T>
__global__ void k_copy2d_zero(const T* __restrict__ src, int64_t lds, int64_t rs, int64_t cs, T* __restrict__ dst,
                              int64_t ldd, int64_t rd, int64_t cd) {
  int64_t r = blockIdx.y * (int64_t)blockDim.y + threadIdx.y;
  int64_t c = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (r >= rd || c >= cd) return;
  dst[r * ldd + c] = (r < rs && c < cs) ? src[r * lds + c] : T(0);
}

// y[j] = sum_{i < rows} M[i][j] x[i]   (M' x ; one thread per column, fixed summation order)
template <typename T>
__global__ void k_gemv_cols(const T* __restrict__ M, int64_t ld, int64_t rows, int64_t cols, const T* __restrict__ x,
                            T* __restrict__ y) {
  int64_t j = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (j >= cols) return;
  T s = T(0);
  for (int64_t i = 0; i < rows; ++i) s += M[i * ld + j] * x[i];
  y[j] = s;
}

// out = A + (B + B')/2 on the n x n block (ld shared)  -- Kinv + sym(kappa_a' invD_a kappa_a) of the online natural gradient
template <typename T>
__global__ void k_add_sym(const T* __restrict__ A, const T* __restrict__ Bm, int64_t ld, int64_t n, T* __restrict__ out) {
  int64_t r = blockIdx.y * (int64_t)blockDim.y + threadIdx.y;
  int64_t c = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (r >= n || c >= n) return;
  out[r * ld + c] = A[r * ld + c] + T(0.5) * (Bm[r * ld + c] + Bm[c * ld + r]);
}

// A -= B on an n x n block
template <typename T>
__global__ void k_sub2d(T* __restrict__ A, const T* __restrict__ Bm, int64_t ld, int64_t n) {
  int64_t r = blockIdx.y * (int64_t)blockDim.y + threadIdx.y;
  int64_t c = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (r < n && c < n) A[r * ld + c] -= Bm[r * ld + c];
}

// out = A + x x' on the n x n block (Sigma + mu mu')
template <typename T>
__global__ void k_add_outer(const T* __restrict__ A, const T* __restrict__ x, int64_t ld, int64_t n, T* __restrict__ out) {
  int64_t r = blockIdx.y * (int64_t)blockDim.y + threadIdx.y;
  int64_t c = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (r >= ld || c >= ld) return;
  out[r * ld + c] = (r < n && c < n) ? A[r * ld + c] + x[r] * x[c] : T(0);
}

// G_kappa_a = H2 - eta_a mu' - H3/2   (rows < ma, cols < m ; zero elsewhere)  -- online hyper-gradient
template <typename T>
__global__ void k_online_gkappa(int64_t ma, int64_t m, int64_t rows, int64_t cols, int64_t ld, const T* __restrict__ ea,
                                const T* __restrict__ mu, const T* __restrict__ H3, T* __restrict__ H2) {
  int64_t r = blockIdx.y * (int64_t)blockDim.y + threadIdx.y;
  int64_t c = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (r >= rows || c >= cols) return;
  H2[r * ld + c] = (r < ma && c < m) ? H2[r * ld + c] - ea[r] * mu[c] - T(0.5) * H3[r * ld + c] : T(0);
}

// A += alpha * Bm (rows x cols, shared ld)
template <typename T>
__global__ void k_axpy2d(int64_t rows, int64_t cols, int64_t ld, T alpha, const T* __restrict__ Bm, T* __restrict__ A) {
  int64_t r = blockIdx.y * (int64_t)blockDim.y + threadIdx.y;
  int64_t c = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (r < rows && c < cols) A[r * ld + c] += alpha * Bm[r * ld + c];
}

// multi-output full predictive covariance (predictions.jl:82-90): cov_t (+)= A[t][q]^2 C_q for every task t ; C is n x n with
// leading dimension ldc, out is T[n_task][n][n] ; first = 1 initialises
template <typename T>
__global__ void k_mo_cov_acc(int64_t n, int nT, const T* __restrict__ Aq, int64_t lda, const T* __restrict__ Cq, int64_t ldc,
                             T* __restrict__ out, int first) {
  int64_t r = blockIdx.y * (int64_t)blockDim.y + threadIdx.y;
  int64_t c = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (r >= n || c >= n) return;
  const T v = Cq[r * ldc + c];
  for (int t = 0; t < nT; ++t) {
    const T a = Aq[t * lda];
    T* o = out + ((int64_t)t * n + r) * n + c;
    *o = (first ? T(0) : *o) + a * a * v;
  }
}

// A -= (M + M')/2 on the n x n block
template <typename T>
__global__ void k_sub_sym(T* __restrict__ A, const T* __restrict__ M, int64_t ld, int64_t n) {
  int64_t r = blockIdx.y * (int64_t)blockDim.y + threadIdx.y;
  int64_t c = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (r < n && c < n) A[r * ld + c] -= T(0.5) * (M[r * ld + c] + M[c * ld + r]);
}

// out[0] = sum_{i<n} x[i] y[i]
template <typename T>
__global__ void k_dot(const T* __restrict__ x, const T* __restrict__ y, int64_t n, double* __restrict__ out) {
  __shared__ double red[16];
  double s = 0.0;
  for (int64_t i = threadIdx.x; i < n; i += blockDim.x) s += (double)x[i] * (double)y[i];
  s = block_sum<double>(s, red);
  if (threadIdx.x == 0) out[0] = s;
}

template <typename T>
__global__ void k_set_scalar(T* p, T v) {
  if (threadIdx.x == 0 && blockIdx.x == 0) *p = v;
}

template <typename T>
__global__ void k_fill(T* p, int64_t n, T v) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

// dst(n x n, ld) = diag value on the diagonal, 0 elsewhere
template <typename T>
__global__ void k_set_identity(T* p, int64_t n, int64_t ld, T diag) {
  int64_t i = blockIdx.y * (int64_t)blockDim.y + threadIdx.y;
  int64_t j = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < n && j < n) p[i * ld + j] = (i == j) ? diag : T(0);
}

// strided 2-D copy with optional padding: dst[r][c] = (r < rs && c < cs) ? src[r][c] : padval(r,c)
template <typename T>
__global__ void k_copy2d(const T* __restrict__ src, int64_t lds, int64_t rs, int64_t cs, T* __restrict__ dst,
                         int64_t ldd, int64_t rd, int64_t cd, T pad_diag, T scale) {
  int64_t r = blockIdx.y * (int64_t)blockDim.y + threadIdx.y;
  int64_t c = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (r >= rd || c >= cd) return;
  T v = (r < rs && c < cs) ? src[r * lds + c] * scale : ((r == c) ? pad_diag : T(0));
  dst[r * ldd + c] = v;
}

// ---------------------------------------------------------------------------------------------------
// ELBO data terms (analyticVI.jl:255-274): one workgroup, double accumulation, deterministic.
//   out[0] = expec_loglikelihood (gaussian.jl:82-93, logistic.jl:73-84, studentt.jl:103-119, logisticsoftmax.jl:106-115)
//   out[1] = AugmentedKL          (logistic.jl:86-92, studentt.jl:121-127, logisticsoftmax.jl:117-140)
// LogisticSoftMax: sums over the nl local latents; the GammaEntropy term and nothing else is global and is added
// only when add_global != 0 (latent_offset == 0 rank).  elbo_ref != 0 reproduces logistic.jl:82 (dot(theta, mu)).
// ---------------------------------------------------------------------------------------------------
// Round 5: one instantiation per likelihood (KIND) -- the single kernel with every likelihood's branch in its loop body compiled to
// 636 bytes of scratch per lane (lgamma + digamma + nine branches under a 1024-thread register budget); per-launch constants
// (lgamma / digamma of the Student-t shape parameters, logs of fixed parameters) are hoisted out of the loop over the points.
template <typename T, int KIND>
__global__ void __launch_bounds__(1024)
k_elbo_terms(int64_t B, int nl, int64_t ldb, LikParams<T> lp, int elbo_ref, int latent_offset, int add_global,
             const T* __restrict__ y, const int32_t* __restrict__ ycls, const int64_t* __restrict__ idx,
             const T* __restrict__ muf, const T* __restrict__ varf, const T* __restrict__ c, const T* __restrict__ theta,
             const T* __restrict__ gamma, const T* __restrict__ alpha, const T* __restrict__ beta, double* __restrict__ out,
             int64_t ystr, const T* __restrict__ lam, int once) {
  // once = 0 on the minibatch shards of a batch-parallel run other than the first: the reference's once-per-evaluation terms
  // (not sums over points) must enter the all-reduced ELBO a single time
  __shared__ double red[16];
  double e = 0.0, kl = 0.0;
  constexpr double LOG2 = 0.69314718055994530942, LOG2PI = 1.83787706640934548356, LOGPI = 1.14472988584940017414;
  const double lamv = (KIND == LIK_POISSON || KIND == LIK_HETERO) ? (double)lam[0] : 1.0;
  // per-launch constants
  double k0 = 0.0, k1 = 0.0, k2 = 0.0, k3 = 0.0;
  if constexpr (KIND == LIK_STUDENTT) {
    const double nu = (double)lp.p0, sig = (double)lp.p1;
    const double al = 0.5 * (nu + 1.0), alp = 0.5 * nu, bp = alp * sig * sig;
    k0 = -0.5 * log(2.0 * 3.14159265358979323846 * sig * sig);
    k1 = digamma_d(al);
    k2 = (al - alp) * k1 - lgamma(al) + lgamma(alp);
    k3 = log(bp);
  } else if constexpr (KIND == LIK_GAUSSIAN) {
    k0 = lp.noise_dev ? (double)lam[0] : (double)lp.p0;
    k1 = log(k0);
  } else if constexpr (KIND == LIK_NEGBIN) {
    k0 = lgamma((double)lp.p0);
  } else if constexpr (KIND == LIK_POISSON || KIND == LIK_HETERO) {
    k0 = log(lamv);
  }
  for (int64_t i = threadIdx.x; i < B; i += blockDim.x) {
    if constexpr (KIND == LIK_LSM) {
      int cls = ycls[idx ? idx[i] : i];
      double a = (double)alpha[i], b = (double)beta[i];
      double psi = digamma_d(a);
      double lam0 = a / b, psil = psi - log(b);
      for (int k = 0; k < nl; ++k) {
        double yk = (cls == latent_offset + k) ? 1.0 : 0.0;
        double g = (double)gamma[k * ldb + i], th = (double)theta[k * ldb + i];
        double mu = (double)muf[k * ldb + i], s = (double)varf[k * ldb + i], cc = (double)c[k * ldb + i];
        e += -LOG2 - (g + yk) * LOG2 + 0.5 * (mu * (yk - g) - th * mu * mu - th * s);
        kl += (yk + g) * logcosh_d(0.5 * cc) - 0.5 * cc * cc * th;          // PolyaGammaKL
        kl += lam0 - g + (g > 0.0 ? g * log(g) : 0.0) - g * psil;           // PoissonKL
      }
      if (add_global) kl += -a - lgamma_pos_d(a) - (1.0 - a) * psi;               // GammaEntropy (sum parts)
    } else if constexpr (KIND == LIK_HETERO) {  // heteroscedastic.jl:142-179 ; layout as in k_hetero_local
      double yi = (double)y[(idx ? idx[i] : i) * ystr];
      double m0 = (double)muf[i], s0 = (double)varf[i], m1 = (double)muf[ldb + i], s1 = (double)varf[ldb + i];
      double g = (double)gamma[i], th = (double)theta[ldb + i], cc = (double)c[ldb + i];
      double lam0 = lamv * ((yi - m0) * (yi - m0) + s0) / 2.0;
      e += 0.5 * k0 - log(2.0 * sqrt(2.0 * 3.14159265358979323846));
      e += 0.5 * (m1 * (0.5 - g) - m1 * m1 * th - s1 * th);
      e -= lam0 - g + (g > 0.0 ? g * log(g) : 0.0) - g * log(lam0);  // PoissonKL(gamma, lam0, log lam0)
      kl += (0.5 + g) * logcosh_d(0.5 * cc) - 0.5 * cc * cc * th;
    } else {
      double yi = (double)y[(idx ? idx[i] : i) * ystr];
      double mu = (double)muf[i], s = (double)varf[i];
      if constexpr (KIND == LIK_GAUSSIAN) {
        e += -0.5 * (LOG2PI + k1 + ((yi - mu) * (yi - mu) + s) / k0);
      } else if constexpr (KIND == LIK_LOGISTIC) {
        double th = (double)theta[i], cc = (double)c[i];
        double quad = elbo_ref ? th * mu : th * mu * mu;
        e += -0.5 * LOG2 + 0.5 * (mu * yi - th * s - quad);
        kl += logcosh_d(0.5 * cc) - 0.5 * cc * cc * th;
      } else if constexpr (KIND == LIK_LAPLACE) {
        // laplace.jl:93-123 ; GIGEntropy (KLdivergences.jl:105-113) at p = 1/2 in closed form:
        //   log(2 K_1/2(s)) = log2 + (log(pi/2) - log s)/2 - s ;  s (K_3/2 + K_-1/2) / (2 K_1/2) = s + 1/2
        // elbo_ref keeps the reference's scalar-iteration quirks: log(a) counted once, log(2 K_p) of the first point only
        double be = (double)lp.p0, a = 1.0 / (be * be);
        double th = (double)theta[i], b = (double)c[i], sab = sqrt(a) * b;
        e += -0.5 * LOG2PI + 0.5 * log(th) - 0.5 * th * (s + mu * mu - 2.0 * mu * yi + yi * yi);
        double l2k = LOG2 + 0.5 * (LOGPI - LOG2 - log(sab)) - sab;
        double ent = -0.5 * log(b * b) + sab + 0.5;
        if (elbo_ref) ent += (i == 0 && once) ? 0.5 * log(a) + l2k : 0.0;
        else ent += 0.5 * log(a) + l2k;
        double expo = -log(2.0 * be * be) - 0.5 * (a * b + b * b * sqrt(a)) / (a * b * b * be * be);
        kl += ent - expo;
      } else if constexpr (KIND == LIK_BSVM) {  // bayesiansvm.jl:71-92 ; elbo_ref: + theta (1 - y mu)^2 as written in line 81
        double th = (double)theta[i], cc = (double)c[i], d = 1.0 - yi * mu, sc = sqrt(cc);
        e += -0.5 * LOG2 + mu * yi - 0.5 * th * s + (elbo_ref ? th * d * d : -0.5 * th * d * d);
        kl += 0.5 * log(cc) + (LOG2 + 0.5 * (LOGPI - LOG2 - log(sc)) - sc) - 0.5 * sc;
      } else if constexpr (KIND == LIK_POISSON) {  // poisson.jl:106-132
        double th = (double)theta[i], cc = (double)c[i], g = (double)gamma[i];
        e += 0.5 * (mu * (yi - g) - th * mu * mu - th * s) + yi * k0 - lgamma_pos_d(yi + 1.0) - LOG2 * (yi + g);
        kl += lamv - (1.0 + k0) * g + (g > 0.0 ? g * log(g) : 0.0);             // PoissonKL(gamma, lambda)
        kl += (yi + g) * logcosh_d(0.5 * cc) - 0.5 * cc * cc * th;              // PolyaGammaKL(y + gamma, c, theta)
      } else if constexpr (KIND == LIK_NEGBIN) {  // negativebinomial.jl:103-131 ; elbo_ref: dot(theta, mu) as written in line 125
        double th = (double)theta[i], cc = (double)c[i], rr = (double)lp.p0;
        e += lgamma_pos_d(yi + rr) - lgamma_pos_d(yi + 1.0) - k0 - LOG2 * (yi + rr);
        e += 0.5 * mu * (yi - rr) - 0.5 * (elbo_ref ? th * mu : th * mu * mu) - 0.5 * th * s;
        kl += (yi + rr) * logcosh_d(0.5 * cc) - 0.5 * cc * cc * th;
      } else {  // Student-t, studentt.jl:103-127
        static_assert(KIND == LIK_STUDENTT, "k_elbo_terms: unknown likelihood");
        const double nu = (double)lp.p0, sig = (double)lp.p1;
        const double al = 0.5 * (nu + 1.0), alp = 0.5 * nu, bp = alp * sig * sig;
        double th = (double)theta[i], cc = (double)c[i];
        const double lc = log(cc);
        e += k0 - (lc - k1) - 0.5 * (th * s + th * mu * mu - 2.0 * th * mu * yi + th * yi * yi);
        kl += k2 + alp * (lc - k3) + al * (bp - cc) / cc;
      }
    }
  }
  e = block_sum<double>(e, red);
  kl = block_sum<double>(kl, red);
  if (threadIdx.x == 0) {
    if (KIND == LIK_LSM && add_global && once) kl += log((double)beta[0]);  // sum(log, first(beta)) (Q16)
    out[0] = e;
    out[1] = kl;
  }
}

// sum_{a,b < m} A[a][b]*Bm[a][b]  -> out[0]   (tr(K^-1 Sigma) as a Frobenius dot, KLdivergences.jl:17), deterministic in two
// stages: one workgroup per row writes part[a], k_sum_parts adds them up (a single workgroup over the m^2 elements took
// 446 us at m = 1024, a third of an ELBO evaluation)
template <typename T>
__global__ void k_frob_dot_rows(const T* __restrict__ A, const T* __restrict__ Bm, int64_t ld, int64_t m,
                                double* __restrict__ part) {
  __shared__ double red[16];
  const int64_t a = blockIdx.x;
  double s = 0.0;
  for (int64_t b = threadIdx.x; b < m; b += blockDim.x) s += (double)A[a * ld + b] * (double)Bm[a * ld + b];
  s = block_sum<double>(s, red);
  if (threadIdx.x == 0) part[a] = s;
}
__global__ void k_sum_parts(const double* __restrict__ part, int64_t n, double* __restrict__ out) {
  __shared__ double red[16];
  double s = 0.0;
  for (int64_t i = threadIdx.x; i < n; i += blockDim.x) s += part[i];
  s = block_sum<double>(s, red);
  if (threadIdx.x == 0) out[0] = s;
}

// out[0] = sum_{i<n} x[i]^2
template <typename T>
__global__ void k_sumsq(const T* __restrict__ x, int64_t n, double* __restrict__ out) {
  __shared__ double red[16];
  double s = 0.0;
  for (int64_t i = threadIdx.x; i < n; i += blockDim.x) s += (double)x[i] * (double)x[i];
  s = block_sum<double>(s, red);
  if (threadIdx.x == 0) out[0] = s;
}

template <typename T>
__global__ void k_axpby(int64_t n, T a, const T* __restrict__ x, T b, const T* __restrict__ y, T* __restrict__ z) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < n) z[i] = a * (x ? x[i] : T(0)) + b * (y ? y[i] : T(0));
}

// ---------------------------------------------------------------------------------------------------
// prediction post-processing (src/training/predictions.jl:25-50,178-247)
// ---------------------------------------------------------------------------------------------------
// mu[i] = sum_s pm[s][i] ; var[i] = kdiag + jitter - sum_s pv[s][i]
template <typename T>
__global__ void k_predict_finish(int64_t n, int nsm, const T* __restrict__ pm, int nsv, const T* __restrict__ pv,
                                 int64_t ldp, T kdiag, T jitter, T* __restrict__ mu, T* __restrict__ var) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  T s = T(0);
  for (int q = 0; q < nsm; ++q) s += pm[q * ldp + i];
  mu[i] = s;
  if (var) {
    T v = T(0);
    for (int q = 0; q < nsv; ++q) v += pv[q * ldp + i];
    var[i] = kdiag + jitter - v;
  }
}

// predict_y: logistic -> mu > 0 ; LogisticSoftMax -> argmax_k (first maximum, like Julia argmax)
template <typename T>
__global__ void k_predict_label(int64_t n, int nl, int64_t ldm, int latent_offset, const T* __restrict__ mu,
                                int32_t* __restrict__ out, int binary) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (binary) {
    out[i] = mu[i] > T(0) ? 1 : 0;
    return;
  }
  int best = 0;
  T bv = mu[i];
  for (int k = 1; k < nl; ++k) {
    T v = mu[k * ldm + i];
    if (v > bv) {
      bv = v;
      best = k;
    }
  }
  out[i] = best + latent_offset;
}

// x <- (x > 0) ? 1 : 0   (predict_y of a Bernoulli task inside a multi-output float buffer)
template <typename T>
__global__ void k_step01(T* __restrict__ x, int64_t n) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < n) x[i] = x[i] > T(0) ? T(1) : T(0);
}

// compute_proba for BernoulliLikelihood{LogisticLink} (classification.jl:14-26): Gauss-Hermite expectation of
// sigma(f) and sigma(f)^2 with f ~ N(mu, max(var,0)) ; nodes/weights in device memory (already x*sqrt2, w/sqrt(pi)).
template <typename T>
__global__ void k_proba_logistic(int64_t n, const T* __restrict__ mu, const T* __restrict__ var, int nn,
                                 const double* __restrict__ nodes, const double* __restrict__ weights,
                                 T* __restrict__ p, T* __restrict__ pv) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  double m = (double)mu[i], sd = sqrt(fmax((double)var[i], 0.0));
  double s1 = 0.0, s2 = 0.0;
  for (int q = 0; q < nn; ++q) {
    double x = nodes[q] * sd + m;
    double sg = x >= 0.0 ? 1.0 / (1.0 + exp(-x)) : exp(x) / (1.0 + exp(x));
    s1 += weights[q] * sg;
    s2 += weights[q] * sg * sg;
  }
  p[i] = (T)s1;
  pv[i] = (T)fmax(s2 - s1 * s1, 0.0);
}

// compute_proba by Gauss-Hermite for the other links: link 1 = SVMLink (bayesiansvm.jl:27-40, variance clamped at 0 as in
// classification.jl:23), 2 = lambda*logistic (poisson.jl:46-57), 3 = NegBinomial mean p r/(1-p) (negativebinomial.jl:46-62)
template <typename T>
__global__ void k_proba_gh(int64_t n, const T* __restrict__ mu, const T* __restrict__ var, int nn,
                           const double* __restrict__ nodes, const double* __restrict__ weights, int link, double par,
                           const T* __restrict__ lam, T* __restrict__ p, T* __restrict__ pv) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (link == 2 && lam) par = (double)lam[0];
  double m = (double)mu[i], sd = sqrt(fmax((double)var[i], 0.0));
  double s1 = 0.0, s2 = 0.0;
  for (int q = 0; q < nn; ++q) {
    double x = nodes[q] * sd + m, v;
    if (link == 1) {
      double pos = exp(-2.0 * fmax(1.0 - x, 0.0)), neg = exp(-2.0 * fmax(1.0 + x, 0.0));
      v = pos / (pos + neg);
    } else {
      double sg = x >= 0.0 ? 1.0 / (1.0 + exp(-x)) : exp(x) / (1.0 + exp(x));
      v = link == 2 ? par * sg : sg * par / (1.0 - sg);
    }
    s1 += weights[q] * v;
    s2 += weights[q] * v * v;
  }
  p[i] = (T)s1;
  double vv = s2 - s1 * s1;
  pv[i] = (T)(link == 1 ? fmax(vv, 0.0) : vv);
}

// predict_y of the event likelihoods (predictions.jl:211): Poisson lambda*logistic(mu) ; NegBinomial r (1-p)/p, p = logistic(-mu)
template <typename T>
__global__ void k_predict_event(int64_t n, const T* __restrict__ mu, int negbin, double par, const T* __restrict__ lam,
                                T* __restrict__ out) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  double x = (double)mu[i];
  if (!negbin) {
    if (lam) par = (double)lam[0];
    out[i] = (T)(par * (x >= 0.0 ? 1.0 / (1.0 + exp(-x)) : exp(x) / (1.0 + exp(x))));
  } else {
    double pn = -x >= 0.0 ? 1.0 / (1.0 + exp(x)) : exp(-x) / (1.0 + exp(-x));
    out[i] = (T)(par * (1.0 - pn) / pn);
  }
}

// heteroscedastic compute_proba (heteroscedastic.jl:63-69): (mu_f, var_f + 1/(lambda logistic(mu_g))) ; mu/var are [2][ld]
template <typename T>
__global__ void k_proba_hetero(int64_t n, int64_t ld, const T* __restrict__ mu, const T* __restrict__ var,
                               const T* __restrict__ lam, T* __restrict__ o0, T* __restrict__ o1) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  double x = (double)mu[ld + i];
  double sg = x >= 0.0 ? 1.0 / (1.0 + exp(-x)) : exp(x) / (1.0 + exp(x));
  o0[i] = mu[i];
  o1[i] = (T)((double)var[i] + 1.0 / ((double)lam[0] * sg));
}

// (mu, var + add) for Gaussian (gaussian.jl:41-45) ; (mu, max(var,0) + add) for StudentT (studentt.jl:57-61)
template <typename T>
__global__ void k_proba_regression(int64_t n, const T* __restrict__ mu, const T* __restrict__ var, T add, int clamp,
                                   T* __restrict__ o0, T* __restrict__ o1) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  o0[i] = mu[i];
  T v = var[i];
  if (clamp) v = v > T(0) ? v : T(0);
  o1[i] = v + add;
}

// LogisticSoftMax link on the means only (multiclass.jl:96-117, logisticsoftmax.jl:29-31): out[i][k] row-major n x nl
template <typename T>
__global__ void k_proba_lsm(int64_t n, int nl, int64_t ldm, const T* __restrict__ mu, T* __restrict__ out) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  double s = 0.0;
  for (int k = 0; k < nl; ++k) {
    double x = (double)mu[k * ldm + i];
    double sg = x >= 0.0 ? 1.0 / (1.0 + exp(-x)) : exp(x) / (1.0 + exp(x));
    s += sg;
  }
  for (int k = 0; k < nl; ++k) {
    double x = (double)mu[k * ldm + i];
    double sg = x >= 0.0 ? 1.0 / (1.0 + exp(-x)) : exp(x) / (1.0 + exp(x));
    out[i * nl + k] = (T)(sg / s);
  }
}

}  // namespace agp
