// agp_linalg.h -- dense kernels of the CAVI path for gfx950, all on padded (multiple-of-64) row-major
// matrices so no tile needs a bounds check:
//   k_gemm_nt      C = A B^T with fused row-dot epilogues          (kappa = Knm K^-1, W = kappa L_A^-T, predict var)
//   k_syrk_tn      S = A^T diag(w) A, lower tiles mirrored, fused natural-gradient step on eta2
//   k_potrf_*      blocked right-looking Cholesky: LDS/register diagonal block + MFMA panel and trailing update
//   k_trtri_step   triangular inverse by recursive doubling of MFMA products
// Reference call sites: src/gpblocks/latentgp.jl:205-215, src/inference/analyticVI.jl:160-180,
// src/inference/inference.jl:25-28 (all LAPACK/BLAS there).
#pragma once
#include "agp_device.h"

namespace agp {

// ---------------------------------------------------------------------------------------------------
// C(M x N) = A(M x K) * B(N x K)^T, both operands k-contiguous ("NT").  grid = (N/64, M/64).
//   tri_b != 0 : B is lower triangular (B[j][k] = 0 for k > j) -> k range stops at the tile's last column.
// Epilogues:
//   EPI_STORE   C = acc
//   EPI_KAPPA   C = acc ; part0[slice][row] = sum_col acc * E[row][col]         (K~ ingredient, latentgp.jl:212)
//   EPI_W       no C    ; part0 = sum acc^2 ; part1 = sum acc * v[col]          (var_f / mean_f, latentgp.jl:179,189)
//   EPI_ROWDOT  no C    ; part0 = sum acc * E[row][col]                         (predict variance, predictions.jl:42)
//   EPI_EMINUS  C = E - acc                                                     (A = K^-1 - K^-1 Sigma K^-1, predictions.jl:38)
// Partial slices: slice = blockIdx.x*2 + wn, each of length ldp; the consumer sums slices in fixed order
// (deterministic, no atomics).
// ---------------------------------------------------------------------------------------------------
enum { EPI_STORE = 0, EPI_KAPPA = 1, EPI_W = 2, EPI_ROWDOT = 3, EPI_EMINUS = 4 };

template <typename T, int EPI>
__global__ __launch_bounds__(NTHREADS) void k_gemm_nt(const T* __restrict__ A, int64_t lda, const T* __restrict__ B,
                                                      int64_t ldb, int64_t K, int tri_b, T* __restrict__ C,
                                                      int64_t ldc, const T* __restrict__ E, int64_t lde,
                                                      const T* __restrict__ v, T* __restrict__ part0,
                                                      T* __restrict__ part1, int64_t ldp) {
  __shared__ __attribute__((aligned(16))) T smem[SMEM_ELEMS];
  const int64_t bn = blockIdx.x, bm = blockIdx.y;
  const int64_t r0 = bm * TILE, c0 = bn * TILE;
  Acc<T> acc;
  acc.zero();
  int64_t kEnd = tri_b ? ((c0 + TILE) < K ? (c0 + TILE) : K) : K;
  gemm_tile<T, KC, KC>(A + r0 * lda, lda, B + c0 * ldb, ldb, 0, kEnd, nullptr, acc, smem);
  if (EPI == EPI_STORE) {
    acc_foreach<T>(acc, [&](int r, int c, T val) { C[(r0 + r) * ldc + c0 + c] = val; });
  }
  if (EPI == EPI_KAPPA) {  // kappa goes to C and to the augmented-Cholesky workspace (part1, same ldc)
    acc_foreach<T>(acc, [&](int r, int c, T val) {
      C[(r0 + r) * ldc + c0 + c] = val;
      if (part1) part1[(r0 + r) * ldc + c0 + c] = val;
    });
  }
  if (EPI == EPI_EMINUS) {
    acc_foreach<T>(acc, [&](int r, int c, T val) { C[(r0 + r) * ldc + c0 + c] = E[(r0 + r) * lde + c0 + c] - val; });
  }
  if (EPI == EPI_KAPPA || EPI == EPI_ROWDOT || EPI == EPI_W) {
    const int wn = (threadIdx.x >> 6) & 1;
    T* p0 = part0 + (bn * 2 + wn) * ldp;
    T* p1 = (EPI == EPI_W) ? part1 + (bn * 2 + wn) * ldp : nullptr;
    acc_row_reduce<T>(
        acc,
        [&](int r, int c, T val, T& s0, T& s1) {
          if (EPI == EPI_W) {
            s0 += val * val;
            s1 += val * v[c0 + c];
          } else {
            s0 += val * E[(r0 + r) * lde + c0 + c];
          }
        },
        p0, p1, r0);
  }
}

// ---------------------------------------------------------------------------------------------------
// S(n x n) = A(Kdim x n)^T diag(w) A(Kdim x n)  ("TN", operands row-contiguous), lower tiles only, mirrored.
// grid = nt*(nt+1)/2 linear over lower-triangular tiles.
//   lower_a != 0 : A is lower triangular (A[k][a] = 0 for k < a) -> k range starts at the tile's first row
//                  (K^-1 = X'X with X = L^-1 ; Sigma = X_A' X_A).
// Modes:
//   SY_STORE : S -> out (both triangles)
//   SY_ETA2  : fused natural-gradient step (analyticVI.jl:172-180, 229-246):
//                g = -(S + Kinv/2) - eta2 ; eta2 += lr*g ; out(=Amat) = -2*eta2     (both triangles)
//              lr = RobbinsMonro step (or 1 for AnalyticVI), passed by value.
// ---------------------------------------------------------------------------------------------------
enum { SY_STORE = 0, SY_ETA2 = 1 };

__device__ __forceinline__ void tri_index(int64_t idx, int64_t& ti, int64_t& tj) {
  // idx = ti*(ti+1)/2 + tj, tj <= ti
  int64_t t = (int64_t)((sqrt(8.0 * (double)idx + 1.0) - 1.0) * 0.5);
  while (t * (t + 1) / 2 > idx) --t;
  while ((t + 1) * (t + 2) / 2 <= idx) ++t;
  ti = t;
  tj = idx - t * (t + 1) / 2;
}

template <typename T, int MODE>
__global__ __launch_bounds__(NTHREADS) void k_syrk_tn(const T* __restrict__ A, int64_t lda, int64_t Kdim,
                                                      const T* __restrict__ w, int lower_a, T* __restrict__ out,
                                                      int64_t ldo, T* __restrict__ eta2, const T* __restrict__ Kinv,
                                                      int64_t ldm, T lr) {
  __shared__ __attribute__((aligned(16))) T smem[SMEM_ELEMS];
  int64_t ta, tb;
  tri_index(blockIdx.x, ta, tb);
  const int64_t a0 = ta * TILE, b0 = tb * TILE;
  Acc<T> acc;
  acc.zero();
  int64_t kBegin = lower_a ? a0 : 0;
  gemm_tile<T, RC, RC>(A + a0, lda, A + b0, lda, kBegin, Kdim, w, acc, smem);
  if (MODE == SY_STORE) {
    acc_foreach<T>(acc, [&](int r, int c, T val) {
      int64_t gr = a0 + r, gc = b0 + c;
      if (ta != tb) {
        out[gr * ldo + gc] = val;
        out[gc * ldo + gr] = val;
      } else if (gc <= gr) {  // diagonal tile: take the lower half as the truth, mirror it
        out[gr * ldo + gc] = val;
        out[gc * ldo + gr] = val;
      }
    });
  } else {
    acc_foreach<T>(acc, [&](int r, int c, T val) {
      int64_t gr = a0 + r, gc = b0 + c;
      if (ta != tb || gc <= gr) {
        T e2 = eta2[gr * ldm + gc];
        T g = -(val + T(0.5) * Kinv[gr * ldm + gc]) - e2;
        e2 += lr * g;
        eta2[gr * ldm + gc] = e2;
        eta2[gc * ldm + gr] = e2;
        out[gr * ldo + gc] = T(-2) * e2;
        out[gc * ldo + gr] = T(-2) * e2;
      }
    });
  }
}

// eta2 step from an already reduced statistic S (batch-parallel multi-GPU path: S was all-reduced)
template <typename T>
__global__ void k_eta2_from_stats(const T* __restrict__ S, int64_t n, T* __restrict__ eta2,
                                  const T* __restrict__ Kinv, T* __restrict__ Amat, T lr) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n * n) return;
  T e2 = eta2[i];
  T g = -(S[i] + T(0.5) * Kinv[i]) - e2;
  e2 += lr * g;
  eta2[i] = e2;
  Amat[i] = T(-2) * e2;
}

// ---------------------------------------------------------------------------------------------------
// Cholesky A = L L' (in place, lower, nb = 64) fused with the triangular inverse X = L^-1, one launch per
// block column.  Launch S(k), k = 0..nt, has three kinds of workgroups:
//   P (blockIdx < nt-k, k < nt)   panel of block column k: applies the pending rank-64 update from column k-1 to
//       its own tile and (redundantly, every P workgroup) to the diagonal tile, factors the diagonal tile, then
//       b == 0 stores L_kk (into Dg[k]) and X_kk = L_kk^-1, b > 0 forms L_ik = A_ik L_kk^-T with MFMA.
//   U (k >= 1)                    trailing update from column k-1 for tiles (i, j), j > k: A_ij -= L_i,k-1 L_j,k-1'
//   X (k >= 2)                    row q = k-1 of the inverse: X_qj = -X_qq sum_{i=j}^{q-1} L_qi X_ij   (j < q)
// so the whole potrf + trtri of an m x m matrix costs nt+1 dependent launches (17 at m = 1024).
//
// Diagonal-tile factorisation (the critical path): 256 threads hold the tile as 4x4 register sub-blocks of A and
// of an identity M; column j is eliminated from BOTH with the same multipliers (Gauss-Jordan on [A | I]), ONE
// barrier per column, no data-dependent branches: owners publish u_j (column j below the diagonal, zeros above)
// and row j of M to LDS, everybody applies  a -= (u_j[R]/p_j) u_j[C],  g -= (u_j[R]/p_j) M_j[C].
// After 64 steps L = a diag(p)^-1/2 (lower part) and L^-1 = diag(p)^-1/2 g.
// L_kk goes to the side buffer Dg (nobody may overwrite A_kk while other P workgroups still read it); the P
// workgroup 0 of the NEXT launch copies it into A.
// info: first non-positive pivot (1-based global column), 0 = success.
// ---------------------------------------------------------------------------------------------------
constexpr int LDP = TILE + 2;  // 66: KC-style stride for 64-deep LDS tiles (conflict-free fragment reads)

__device__ __forceinline__ double fast_rcp(double p) {
  double r = __builtin_amdgcn_rcp(p);
  double e = fma(-p, r, 1.0);
  r = fma(r, e, r);
  e = fma(-p, r, 1.0);
  return fma(r, e, r);
}
__device__ __forceinline__ float fast_rcp(float p) {
  float r = __builtin_amdgcn_rcpf(p);
  float e = fmaf(-p, r, 1.0f);
  return fmaf(r, e, r);
}

// global 64x64 tile (row-major, leading dimension ld) -> LDS [r*LDP + c], 16-byte loads
template <typename T>
__device__ __forceinline__ void load_tile_lds(const T* __restrict__ G, int64_t ld, T* S) {
  typedef typename Mfma<T>::vec_t vec_t;
  constexpr int VEC = Mfma<T>::VEC, NV = TILE / VEC;
#pragma unroll
  for (int v = 0; v < TILE * NV / NTHREADS; ++v) {
    int vi = threadIdx.x + v * NTHREADS;
    int r = vi / NV, cv = vi % NV;
    vec_t x = *reinterpret_cast<const vec_t*>(G + (int64_t)r * ld + cv * VEC);
#pragma unroll
    for (int e = 0; e < VEC; ++e) S[r * LDP + cv * VEC + e] = x[e];
  }
}

// acc += As(64 x 64, [r][k] stride LDP) * Bs(64 x 64 given as [c][k] stride LDP)^T
template <typename T>
__device__ __forceinline__ void mma_lds64(const T* As, const T* Bs, Acc<T>& acc) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wm = wave >> 1, wn = wave & 1;
#pragma unroll 4
  for (int kk = 0; kk < TILE / 4; ++kk) {
    T a0 = As[(wm * 32 + (lane & 15)) * LDP + kk * 4 + (lane >> 4)];
    T a1 = As[(wm * 32 + 16 + (lane & 15)) * LDP + kk * 4 + (lane >> 4)];
    T b0 = Bs[(wn * 32 + (lane & 15)) * LDP + kk * 4 + (lane >> 4)];
    T b1 = Bs[(wn * 32 + 16 + (lane & 15)) * LDP + kk * 4 + (lane >> 4)];
    acc.a[0][0] = Mfma<T>::mma(a0, b0, acc.a[0][0]);
    acc.a[0][1] = Mfma<T>::mma(a0, b1, acc.a[0][1]);
    acc.a[1][0] = Mfma<T>::mma(a1, b0, acc.a[1][0]);
    acc.a[1][1] = Mfma<T>::mma(a1, b1, acc.a[1][1]);
  }
}

// One 16-column group (J = j/16) of the elimination.  Thread (ti, tj) owns the CYCLIC 4x4 sub-blocks
// rows R = ti + 16 r, cols C = tj + 16 c, so "row/col block still alive" is a compile-time property of (r, c, J):
// A side only touches lower blocks J <= c <= r, M side blocks r >= J, c <= J -- about a third of the dense work,
// with no lane-divergent branches.
template <typename T, int J, int VAR = 0>
__device__ __forceinline__ void eliminate_group(T (&a)[4][4], T (&g)[4][4], T* U, T* MR, T* piv, int ti, int tj) {
  for (int jj = 0; jj < 16; ++jj) {
    const int j = J * 16 + jj;
    if (tj == jj) {  // owners of column j publish u_j (zero on and above the diagonal)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        T v = (r < J) ? T(0) : ((r > J) ? a[r][J] : (ti > jj ? a[r][J] : T(0)));
        U[j * TILE + ti + 16 * r] = v;
      }
      if (ti == jj) piv[j] = a[J][J];
    }
    if (ti == jj) {  // owners of row j of M publish it (columns beyond block J are zero and never read)
#pragma unroll
      for (int c = 0; c <= J; ++c) MR[j * TILE + tj + 16 * c] = g[J][c];
    }
    if (VAR != 1) __syncthreads();
    const T rinv = (VAR == 2) ? T(0.5) : fast_rcp(piv[j]);
    T f[4], uC[4], mC[4];
#pragma unroll
    for (int r = J; r < 4; ++r) f[r] = U[j * TILE + ti + 16 * r] * rinv;
#pragma unroll
    for (int c = J; c < 4; ++c) uC[c] = U[j * TILE + tj + 16 * c];
#pragma unroll
    for (int c = 0; c <= J; ++c) mC[c] = MR[j * TILE + tj + 16 * c];
    if (VAR != 3)
#pragma unroll
    for (int r = J; r < 4; ++r) {
#pragma unroll
      for (int c = J; c <= r; ++c) a[r][c] = fma(-f[r], uC[c], a[r][c]);
#pragma unroll
      for (int c = 0; c <= J; ++c) g[r][c] = fma(-f[r], mC[c], g[r][c]);
    }
  }
}

// FOUR columns per LDS round trip.  The elimination chain is latency-bound (LDS write -> barrier -> read is ~180
// cycles, every dependent f64 op ~25), so a 16-column group is processed as 4 rounds: owners publish the 4 raw panel
// columns (rows below the dead zone) and the 4 M rows; EVERY thread redundantly factors the 4x4 pivot block
// (LDL', 4 reciprocals), transforms the panel entries it needs (its rows, its columns, its M columns) and applies a
// rank-4 update.  One barrier per 4 columns instead of one per column; reciprocal = v_rcp_f64 + 1 Newton step
// (1.8e-15 measured).
__device__ __forceinline__ double rcp1(double p) {
  double r = __builtin_amdgcn_rcp(p);
  return fma(r, fma(-p, r, 1.0), r);
}
__device__ __forceinline__ float rcp1(float p) {
  float r = __builtin_amdgcn_rcpf(p);
  return fmaf(r, fmaf(-p, r, 1.0f), r);
}

template <typename T, int J>
__device__ __forceinline__ void eliminate_group4(T (&a)[4][4], T (&g)[4][4], T* PL, T* MW, T* piv, int ti, int tj) {
#pragma unroll 1
  for (int rr = 0; rr < 4; ++rr) {
    const int jj0 = rr * 4, j0 = J * 16 + jj0;
    T* P = PL + (rr & 1) * 4 * TILE;   // double-buffered: round r+1 may publish while stragglers read round r
    T* Mw = MW + (rr & 1) * 4 * TILE;
    const int qc = tj - jj0, qr = ti - jj0;
    if (qc >= 0 && qc < 4) {
#pragma unroll
      for (int r = J; r < 4; ++r) {
        T v = a[r][J];
        if (r == J) v = (ti >= jj0) ? v : T(0);  // rows above the panel are dead
        P[qc * TILE + ti + 16 * r] = v;
      }
    }
    if (qr >= 0 && qr < 4) {
#pragma unroll
      for (int c = 0; c <= J; ++c) Mw[qr * TILE + tj + 16 * c] = g[J][c];
    }
    __syncthreads();
    // ---- 4x4 pivot block, LDL' (every thread, redundantly) ----
    const T d00 = P[0 * TILE + j0], d10 = P[0 * TILE + j0 + 1], d20 = P[0 * TILE + j0 + 2], d30 = P[0 * TILE + j0 + 3];
    T d11 = P[1 * TILE + j0 + 1], d21 = P[1 * TILE + j0 + 2], d31 = P[1 * TILE + j0 + 3];
    T d22 = P[2 * TILE + j0 + 2], d32 = P[2 * TILE + j0 + 3], d33 = P[3 * TILE + j0 + 3];
    const T r0 = rcp1(d00);
    const T l10 = d10 * r0, l20 = d20 * r0, l30 = d30 * r0;
    d11 = fma(-l10, d10, d11);
    d21 = fma(-l20, d10, d21);
    d31 = fma(-l30, d10, d31);
    d22 = fma(-l20, d20, d22);
    d32 = fma(-l30, d20, d32);
    d33 = fma(-l30, d30, d33);
    const T r1 = rcp1(d11);
    const T l21 = d21 * r1, l31 = d31 * r1;
    d22 = fma(-l21, d21, d22);
    d32 = fma(-l31, d21, d32);
    d33 = fma(-l31, d31, d33);
    const T r2 = rcp1(d22);
    const T l32 = d32 * r2;
    d33 = fma(-l32, d32, d33);
    const T r3 = rcp1(d33);
    if (ti == 0 && tj == 0) {
      piv[j0] = d00;
      piv[j0 + 1] = d11;
      piv[j0 + 2] = d22;
      piv[j0 + 3] = d33;
    }
    // ---- panel transforms for my columns / my M columns, then row by row: multipliers + rank-4 update ----
    T u[4][4], mw[4][4];  // [q][c]
#pragma unroll
    for (int c = J; c < 4; ++c) {
      const int x = tj + 16 * c;
      T y0 = P[x], y1 = P[TILE + x], y2 = P[2 * TILE + x], y3 = P[3 * TILE + x];
      y1 = fma(-l10, y0, y1);
      y2 = fma(-l21, y1, fma(-l20, y0, y2));
      y3 = fma(-l32, y2, fma(-l31, y1, fma(-l30, y0, y3)));
      u[0][c] = y0;
      u[1][c] = y1;
      u[2][c] = y2;
      u[3][c] = y3;
      if (c == J) {  // columns on/left of pivot q are frozen
#pragma unroll
        for (int q = 0; q < 4; ++q) u[q][c] = (tj > jj0 + q) ? u[q][c] : T(0);
      }
    }
#pragma unroll
    for (int c = 0; c <= J; ++c) {
      const int x = tj + 16 * c;
      T y0 = Mw[x], y1 = Mw[TILE + x], y2 = Mw[2 * TILE + x], y3 = Mw[3 * TILE + x];
      y1 = fma(-l10, y0, y1);
      y2 = fma(-l21, y1, fma(-l20, y0, y2));
      y3 = fma(-l32, y2, fma(-l31, y1, fma(-l30, y0, y3)));
      mw[0][c] = y0;
      mw[1][c] = y1;
      mw[2][c] = y2;
      mw[3][c] = y3;
    }
#pragma unroll
    for (int r = J; r < 4; ++r) {
      const int x = ti + 16 * r;
      T y0 = P[x], y1 = P[TILE + x], y2 = P[2 * TILE + x], y3 = P[3 * TILE + x];
      y1 = fma(-l10, y0, y1);
      y2 = fma(-l21, y1, fma(-l20, y0, y2));
      y3 = fma(-l32, y2, fma(-l31, y1, fma(-l30, y0, y3)));
      T f[4] = {y0 * r0, y1 * r1, y2 * r2, y3 * r3};
      if (r == J) {  // rows on/above pivot q take no part in its rank-1 update
#pragma unroll
        for (int q = 0; q < 4; ++q) f[q] = (ti > jj0 + q) ? f[q] : T(0);
      }
#pragma unroll
      for (int c = J; c <= r; ++c) {
        T s0 = a[r][c];
#pragma unroll
        for (int q = 0; q < 4; ++q) s0 = fma(-f[q], u[q][c], s0);
        a[r][c] = s0;
      }
#pragma unroll
      for (int c = 0; c <= J; ++c) {
        T s0 = g[r][c];
#pragma unroll
        for (int q = 0; q < 4; ++q) s0 = fma(-f[q], mw[q][c], s0);
        g[r][c] = s0;
      }
    }
  }
}

// In: bufA holds the SPD tile as [R*LDP + C] (lower triangle valid).  Out: bufA = L (upper zero), bufB = L^-1.
template <typename T, int VAR = 4>
__device__ __forceinline__ void factor_diag_tile(T* bufA, T* bufB, T* piv, int32_t* info, int64_t col0,
                                                 int64_t nvalid) {
  const int tid = threadIdx.x;
  const int ti = tid >> 4, tj = tid & 15;
  T a[4][4], g[4][4];
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      int R = ti + 16 * r, Cc = tj + 16 * c;
      int lo = R >= Cc ? R : Cc, hi = R >= Cc ? Cc : R;
      a[r][c] = bufA[lo * LDP + hi];
      g[r][c] = (R == Cc) ? T(1) : T(0);
    }
  __syncthreads();  // bufA / bufB are reused as the u_j / M-row stores from here on
  if (VAR == 4) {
    eliminate_group4<T, 0>(a, g, bufA, bufB, piv, ti, tj);
    eliminate_group4<T, 1>(a, g, bufA, bufB, piv, ti, tj);
    eliminate_group4<T, 2>(a, g, bufA, bufB, piv, ti, tj);
    eliminate_group4<T, 3>(a, g, bufA, bufB, piv, ti, tj);
  } else {
    eliminate_group<T, 0, VAR>(a, g, bufA, bufB, piv, ti, tj);
    eliminate_group<T, 1, VAR>(a, g, bufA, bufB, piv, ti, tj);
    eliminate_group<T, 2, VAR>(a, g, bufA, bufB, piv, ti, tj);
    eliminate_group<T, 3, VAR>(a, g, bufA, bufB, piv, ti, tj);
  }
  __syncthreads();  // all reads of U / MR done; piv complete
  if (tid < TILE) {
    const T p = piv[tid];
    const bool bad = !(p > T(0)) && (col0 + tid) < nvalid;
    const unsigned long long mask = __ballot(bad);
    if (mask != 0ull && tid == 0) {
      int32_t want = (int32_t)(col0 + (__ffsll((long long)mask) - 1) + 1);
      int32_t old = atomicCAS(info, 0, want);
      while (old != 0 && old > want) {
        int32_t prev = atomicCAS(info, old, want);
        if (prev == old) break;
        old = prev;
      }
    }
  }
  T rsC[4], rsR[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    T pc = piv[tj + 16 * q], pr = piv[ti + 16 * q];
    rsC[q] = T(1) / sqrt(pc > T(0) ? pc : T(1));
    rsR[q] = T(1) / sqrt(pr > T(0) ? pr : T(1));
  }
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      int R = ti + 16 * r, Cc = tj + 16 * c;
      bufA[R * LDP + Cc] = (R >= Cc) ? a[r][c] * rsC[c] : T(0);
      bufB[R * LDP + Cc] = (R >= Cc) ? g[r][c] * rsR[r] : T(0);
    }
  __syncthreads();
}

// Extension rows ("augmented Cholesky"): ne extra 64-row blocks E (ld lde) are treated as block rows nt..nt+ne-1
// BELOW A: they receive the panel solve and the trailing updates but never become diagonal blocks, so on exit
// E = E_in * L^-T.  With E_in = [kappa ; eta1'] this yields W = kappa L_A^-T and v' = (L_A^-1 eta1)' -- all that
// mean_f / var_f need (latentgp.jl:179,189) -- without forming L_A^-1 and without a separate B x m x m GEMM.
// do_x == 0 drops the X (inverse) role.  Diagonal factors always go to Dg (A's diagonal tiles keep their input).
template <typename T>
__global__ __launch_bounds__(NTHREADS) void k_potrf_trtri_step(T* __restrict__ A, int64_t ld, T* __restrict__ X,
                                                               int64_t ldx, T* __restrict__ Dg, T* __restrict__ E,
                                                               int64_t lde, int64_t ne, int do_x, int64_t k,
                                                               int64_t nt, int32_t* __restrict__ info,
                                                               int64_t nvalid) {
  // one LDS block: [bufA | bufB | bufC]; the GEMM staging (SMEM_ELEMS) aliases bufA+bufB and is only live before them
  __shared__ __attribute__((aligned(16))) T sm[3 * TILE * LDP];
  __shared__ T piv[TILE];
  static_assert(2 * TILE * LDP >= SMEM_ELEMS, "gemm staging must fit in bufA+bufB");
  T* bufA = sm;
  T* bufB = sm + TILE * LDP;
  T* bufC = sm + 2 * TILE * LDP;
  T* gsm = sm;
  const int tid = threadIdx.x;
  const int64_t nP = (k < nt) ? (nt - k + ne) : 0;
  const int64_t nr = nt - k - 1;
  const int64_t nU = (k >= 1 && k < nt && nr > 0) ? nr * (nr + 1) / 2 + ne * nr : 0;
  int64_t bid = blockIdx.x;
  if (bid < nP) {
    // ---------------- P: panel of block column k ----------------
    const int64_t b = bid, d0 = k * TILE, p0 = (k - 1) * TILE;
    const bool ext = b >= (nt - k);
    T* rowp = ext ? E + (b - (nt - k)) * TILE * lde : A + (k + b) * TILE * ld;  // first row of this block row
    const int64_t ldr = ext ? lde : ld;
    Acc<T> accD, accT;
    accD.zero();
    accT.zero();
    if (k >= 1) {  // pending rank-64 update from column k-1: both 64-deep products straight from LDS, no k-loop
      load_tile_lds<T>(A + d0 * ld + p0, ld, bufA);
      if (b > 0) load_tile_lds<T>(rowp + p0, ldr, bufC);
      __syncthreads();
      mma_lds64<T>(bufA, bufA, accD);
      if (b > 0) mma_lds64<T>(bufC, bufA, accT);
      __syncthreads();
    }
    acc_foreach<T>(accD, [&](int r, int c, T val) { bufA[r * LDP + c] = A[(d0 + r) * ld + d0 + c] - val; });
    if (b > 0)  // own tile with the pending update applied: parked in LDS so no accumulator lives across the factorisation
      acc_foreach<T>(accT, [&](int r, int c, T val) { bufC[r * LDP + c] = rowp[r * ldr + d0 + c] - val; });
    __syncthreads();
    factor_diag_tile<T>(bufA, bufB, piv, info, d0, nvalid);
    if (b == 0) {
      for (int e = tid; e < TILE * TILE; e += NTHREADS) {
        int R = e >> 6, Cc = e & 63;
        Dg[k * TILE * TILE + e] = bufA[R * LDP + Cc];
        if (do_x) X[(d0 + R) * ldx + d0 + Cc] = bufB[R * LDP + Cc];
      }
      return;
    }
    // L_ik = (A_ik - pending) * Linv^T
    Acc<T> acc;
    acc.zero();
    mma_lds64<T>(bufC, bufB, acc);
    acc_foreach<T>(acc, [&](int r, int c, T val) { rowp[r * ldr + d0 + c] = val; });
    return;
  }
  bid -= nP;
  if (bid < nU) {
    // ---------------- U: trailing update from column k-1, tiles (i, j) with j > k ----------------
    const int64_t ntri = nr * (nr + 1) / 2, p0 = (k - 1) * TILE;
    T* rowp;
    int64_t ldr, j0;
    if (bid < ntri) {
      int64_t ii, jj;
      tri_index(bid, ii, jj);
      rowp = A + (k + 1 + ii) * TILE * ld;
      ldr = ld;
      j0 = (k + 1 + jj) * TILE;
    } else {
      const int64_t t = bid - ntri, e = t / nr, jj = t % nr;
      rowp = E + e * TILE * lde;
      ldr = lde;
      j0 = (k + 1 + jj) * TILE;
    }
    Acc<T> acc;
    acc.zero();
    load_tile_lds<T>(rowp + p0, ldr, bufA);
    load_tile_lds<T>(A + j0 * ld + p0, ld, bufB);
    __syncthreads();
    mma_lds64<T>(bufA, bufB, acc);
    acc_foreach<T>(acc, [&](int r, int c, T val) { rowp[r * ldr + j0 + c] -= val; });
    return;
  }
  bid -= nU;
  if (do_x) {
    // ---------------- X: row q = k-1 of L^-1, tile j = bid < q ----------------
    const int64_t q = k - 1, j = bid, q0 = q * TILE, j0 = j * TILE;
    if (j >= q) return;
    Acc<T> acc;
    acc.zero();
    gemm_tile<T, KC, RC>(A + q0 * ld, ld, X + j0, ldx, j0, q0, nullptr, acc, gsm);
    // stage S transposed (St[c][k]) and X_qq ([r][k]) for the 64-deep product
    acc_foreach<T>(acc, [&](int r, int c, T val) { bufA[c * LDP + r] = val; });
    for (int e = tid; e < TILE * TILE; e += NTHREADS) {
      int R = e >> 6, Cc = e & 63;
      bufB[R * LDP + Cc] = X[(q0 + R) * ldx + q0 + Cc];
    }
    __syncthreads();
    Acc<T> out;
    out.zero();
    mma_lds64<T>(bufB, bufA, out);
    acc_foreach<T>(out, [&](int r, int c, T val) { X[(q0 + r) * ldx + j0 + c] = -val; });
  }
}

// copy the diagonal factors from Dg into the diagonal tiles of an n x n matrix (state export / building blocks)
template <typename T>
__global__ void k_publish_diag(T* __restrict__ A, int64_t ld, const T* __restrict__ Dg) {
  const int64_t k = blockIdx.x;
  for (int e = threadIdx.x; e < TILE * TILE; e += blockDim.x)
    A[(k * TILE + (e >> 6)) * ld + k * TILE + (e & 63)] = Dg[k * TILE * TILE + e];
}

// W row statistics: out0[i] = sum_j W[i][j]^2 ; out1[i] = sum_j W[i][j] v[j]   (one wave per row)
template <typename T>
__global__ void k_w_rowstats(const T* __restrict__ W, int64_t ld, int64_t rows, int64_t cols,
                             const T* __restrict__ v, T* __restrict__ out0, T* __restrict__ out1) {
  int64_t row = blockIdx.x * (int64_t)(blockDim.x >> 6) + (threadIdx.x >> 6);
  int lane = threadIdx.x & 63;
  if (row >= rows) return;
  T s0 = T(0), s1 = T(0);
  for (int64_t j = lane; j < cols; j += 64) {
    T w = W[row * ld + j];
    s0 += w * w;
    s1 += w * v[j];
  }
  for (int o = 32; o > 0; o >>= 1) {
    s0 += __shfl_down(s0, o);
    s1 += __shfl_down(s1, o);
  }
  if (lane == 0) {
    out0[row] = s0;
    out1[row] = s1;
  }
}

// micro-benchmark of the diagonal-tile factorisation alone (tools/bench_diag.py): `reps` factorizations per launch
template <typename T, int VAR>
__global__ __launch_bounds__(NTHREADS) void k_diag_bench(const T* __restrict__ A, T* __restrict__ out, int reps,
                                                         int32_t* info) {
  __shared__ __attribute__((aligned(16))) T sm[2 * TILE * LDP];
  __shared__ T piv[TILE];
  T* bufA = sm;
  T* bufB = sm + TILE * LDP;
  for (int it = 0; it < reps; ++it) {
    for (int e = threadIdx.x; e < TILE * TILE; e += NTHREADS) bufA[(e >> 6) * LDP + (e & 63)] = A[e];
    __syncthreads();
    factor_diag_tile<T, VAR>(bufA, bufB, piv, info, 0, 64);
  }
  for (int e = threadIdx.x; e < TILE * TILE; e += NTHREADS) {
    out[blockIdx.x * 2 * TILE * TILE + e] = bufA[(e >> 6) * LDP + (e & 63)];
    out[blockIdx.x * 2 * TILE * TILE + TILE * TILE + e] = bufB[(e >> 6) * LDP + (e & 63)];
  }
}

// strict-upper 64x64 tiles of an n x n matrix (n = nt*64) -> 0 (so factors read back clean)
template <typename T>
__global__ void k_zero_upper_tiles(T* __restrict__ A, int64_t ld, int64_t n) {
  int64_t i = blockIdx.y * (int64_t)blockDim.y + threadIdx.y;
  int64_t j = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < n && j < n && (j / TILE) > (i / TILE)) A[i * ld + j] = T(0);
}

// y[j] = sum_{k <= j} X[j][k] x[k]   (lower-triangular matvec, one wave per row)
template <typename T>
__global__ void k_trmv_lower(const T* __restrict__ X, int64_t ld, int64_t n, const T* __restrict__ x,
                             T* __restrict__ y) {
  int64_t row = blockIdx.x * (int64_t)(blockDim.x >> 6) + (threadIdx.x >> 6);
  int lane = threadIdx.x & 63;
  if (row >= n) return;
  T s = T(0);
  for (int64_t k = lane; k <= row; k += 64) s += X[row * ld + k] * x[k];
  for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o);
  if (lane == 0) y[row] = s;
}

// y[j] = sum_{k >= j} X[k][j] x[k]   (transpose of the above: mu = X' v) ; one thread per column, coalesced
template <typename T>
__global__ void k_trmv_lower_t(const T* __restrict__ X, int64_t ld, int64_t n, const T* __restrict__ x,
                               T* __restrict__ y) {
  int64_t j = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (j >= n) return;
  T s = T(0);
  for (int64_t k = j; k < n; ++k) s += X[k * ld + j] * x[k];
  y[j] = s;
}

// y = M x for symmetric dense M (n x n): one wave per row
template <typename T>
__global__ void k_symv(const T* __restrict__ M, int64_t ld, int64_t n, const T* __restrict__ x, T* __restrict__ y) {
  int64_t row = blockIdx.x * (int64_t)(blockDim.x >> 6) + (threadIdx.x >> 6);
  int lane = threadIdx.x & 63;
  if (row >= n) return;
  T s = T(0);
  for (int64_t k = lane; k < n; k += 64) s += M[row * ld + k] * x[k];
  for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o);
  if (lane == 0) y[row] = s;
}

// sum of log(diag) over the first nvalid entries (logdet from a Cholesky factor) -> out[0] (double)
template <typename T>
__global__ void k_logdiag_sum(const T* __restrict__ Dg, int64_t nvalid, double* __restrict__ out) {
  __shared__ double red[16];
  double s = 0.0;
  for (int64_t i = threadIdx.x; i < nvalid; i += blockDim.x)
    s += log((double)Dg[(i / TILE) * TILE * TILE + (i % TILE) * (TILE + 1)]);
  s = block_sum<double>(s, red);
  if (threadIdx.x == 0) out[0] = s;
}

// MFMA issue-rate microbenchmark (roofline ceiling): each wave runs `iters` x 8 independent-accumulator MFMAs
template <typename T>
__global__ __launch_bounds__(NTHREADS) void k_mfma_peak(T* out, int iters) {
  typename Mfma<T>::acc_t c[8];
  T a = T(threadIdx.x) * T(1e-3), b = T(blockIdx.x) * T(1e-3) + T(1);
#pragma unroll
  for (int i = 0; i < 8; ++i)
    for (int r = 0; r < 4; ++r) c[i][r] = T(i + r);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) c[i] = Mfma<T>::mma(a, b, c[i]);
  }
  T s = T(0);
#pragma unroll
  for (int i = 0; i < 8; ++i)
    for (int r = 0; r < 4; ++r) s += c[i][r];
  out[blockIdx.x * (int64_t)blockDim.x + threadIdx.x] = s;
}

}  // namespace agp
