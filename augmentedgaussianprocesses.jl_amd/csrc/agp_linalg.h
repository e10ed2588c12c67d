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
  if (EPI == EPI_STORE || EPI == EPI_KAPPA) {
    acc_foreach<T>(acc, [&](int r, int c, T val) { C[(r0 + r) * ldc + c0 + c] = val; });
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
//              lr is read from *lr_dev (RobbinsMonro step or 1 for AnalyticVI).
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
                                                      int64_t ldm, const T* __restrict__ lr_dev) {
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
    const T lr = *lr_dev;
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
                                  const T* __restrict__ Kinv, T* __restrict__ Amat, const T* __restrict__ lr_dev) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n * n) return;
  const T lr = *lr_dev;
  T e2 = eta2[i];
  T g = -(S[i] + T(0.5) * Kinv[i]) - e2;
  e2 += lr * g;
  eta2[i] = e2;
  Amat[i] = T(-2) * e2;
}

// ---------------------------------------------------------------------------------------------------
// Blocked right-looking Cholesky, nb = 64, in place on the lower triangle of A (n = nt*64).
//
// k_potrf_panel (step k, grid = nt-k): EVERY workgroup factors the 64x64 diagonal block redundantly (it is the
//   critical path; redundancy costs nothing and removes a launch): 256 threads hold the block as 4x4 register
//   sub-blocks and eliminate column by column with ONE barrier per column, applying the same row operations to an
//   identity so that L_kk and L_kk^-1 come out together (Gauss-Jordan on [A | I]).  Workgroup 0 stores L_kk and
//   L_kk^-1 (the latter straight into the diagonal block of X = L^-1); workgroup b>0 forms the panel block
//   L_ik = A_ik L_kk^-T with MFMA from LDS.
// k_potrf_update (step k): A_ij -= L_ik L_jk^T for k < j <= i (MFMA, K = 64).
// info: first non-positive pivot (1-based global column) is recorded with atomicMin-style CAS; 0 = success.
// ---------------------------------------------------------------------------------------------------
constexpr int LDP = TILE + 2;  // 66: KC-style stride for 64-deep LDS tiles (conflict-free fragment reads)

template <typename T>
__device__ __forceinline__ T precise_rcp(T x) {
  return T(1) / x;
}

template <typename T>
__device__ __forceinline__ void factor_diag_block(T (&a)[4][4], T* colA, T* rowM, T* Ls, T* Linvs, int32_t* info,
                                                  int64_t col0, int64_t nvalid) {
  const int tid = threadIdx.x;
  const int ti = tid >> 4, tj = tid & 15;
  T g[4][4];
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) g[r][c] = (ti == tj && r == c) ? T(1) : T(0);
  const bool active = (tj <= ti);
  // outer loop dynamic (jb), inner 4 columns unrolled so that register sub-block indices (jr) stay static
  for (int jb = 0; jb < TILE / 4; ++jb)
#pragma unroll
  for (int jr = 0; jr < 4; ++jr) {
    const int j = jb * 4 + jr;
    T* cA = colA + (j & 1) * TILE;
    T* rM = rowM + (j & 1) * TILE;
    if (tj == jb) {
#pragma unroll
      for (int r = 0; r < 4; ++r) cA[4 * ti + r] = a[r][jr];
    }
    if (ti == jb) {
#pragma unroll
      for (int c = 0; c < 4; ++c) rM[4 * tj + c] = g[jr][c];
    }
    __syncthreads();
    T p = cA[j];
    if (tid == 0 && !(p > T(0)) && (col0 + j) < nvalid) {
      int32_t want = (int32_t)(col0 + j + 1);
      int32_t old = atomicCAS(info, 0, want);
      while (old != 0 && old > want) {
        int32_t prev = atomicCAS(info, old, want);
        if (prev == old) break;
        old = prev;
      }
    }
    if (!(p > T(0))) p = T(1);  // keep going with finite garbage; host reports info
    const T rinv = precise_rcp(p);
    const T rs = precise_rcp(sqrt(p));
    if (tj == jb) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int R = 4 * ti + r;
        Ls[R * LDP + j] = (R >= j) ? a[r][jr] * rs : T(0);
      }
    }
    if (ti == jb) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        int Cc = 4 * tj + c;
        Linvs[j * LDP + Cc] = (Cc <= j) ? g[jr][c] * rs : T(0);
      }
    }
    if (active && (4 * ti + 3) > j) {
      T cc[4], mm[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        cc[c] = cA[4 * tj + c];
        mm[c] = rM[4 * tj + c];
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int R = 4 * ti + r;
        if (R > j) {
          T f = cA[R] * rinv;
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            int Cc = 4 * tj + c;
            if (Cc > j) a[r][c] -= f * cc[c];
            else g[r][c] -= f * mm[c];
          }
        }
      }
    }
  }
  __syncthreads();
}

template <typename T>
__global__ __launch_bounds__(NTHREADS) void k_potrf_panel(T* __restrict__ A, int64_t ld, int64_t k, T* __restrict__ X,
                                                          int64_t ldx, int32_t* __restrict__ info, int64_t nvalid) {
  __shared__ __attribute__((aligned(16))) T Ls[TILE * LDP];
  __shared__ __attribute__((aligned(16))) T Linvs[TILE * LDP];
  __shared__ T colA[2 * TILE];
  __shared__ T rowM[2 * TILE];
  const int tid = threadIdx.x;
  const int ti = tid >> 4, tj = tid & 15;
  const int64_t d0 = k * TILE;
  T a[4][4];
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      int R = 4 * ti + r, Cc = 4 * tj + c;
      int lo = R >= Cc ? R : Cc, hi = R >= Cc ? Cc : R;  // read the lower triangle only
      a[r][c] = A[(d0 + lo) * ld + d0 + hi];
    }
  factor_diag_block<T>(a, colA, rowM, Ls, Linvs, info, d0, nvalid);
  const int64_t b = blockIdx.x;
  if (b == 0) {
    for (int e = tid; e < TILE * TILE; e += NTHREADS) {
      int R = e >> 6, Cc = e & 63;
      A[(d0 + R) * ld + d0 + Cc] = Ls[R * LDP + Cc];
      X[(d0 + R) * ldx + d0 + Cc] = Linvs[R * LDP + Cc];
    }
    return;
  }
  // panel block i = k + b : L_ik = A_ik * Linv^T   (C[r][c] = sum_j A_ik[r][j] Linv[c][j])
  const int64_t i0 = (k + b) * TILE;
  for (int e = tid; e < TILE * TILE; e += NTHREADS) {
    int R = e >> 6, Cc = e & 63;
    Ls[R * LDP + Cc] = A[(i0 + R) * ld + d0 + Cc];
  }
  __syncthreads();
  const int lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
  Acc<T> acc;
  acc.zero();
#pragma unroll 4
  for (int kk = 0; kk < TILE / 4; ++kk) {
    T a0 = Ls[(wm * 32 + (lane & 15)) * LDP + kk * 4 + (lane >> 4)];
    T a1 = Ls[(wm * 32 + 16 + (lane & 15)) * LDP + kk * 4 + (lane >> 4)];
    T b0 = Linvs[(wn * 32 + (lane & 15)) * LDP + kk * 4 + (lane >> 4)];
    T b1 = Linvs[(wn * 32 + 16 + (lane & 15)) * LDP + kk * 4 + (lane >> 4)];
    acc.a[0][0] = Mfma<T>::mma(a0, b0, acc.a[0][0]);
    acc.a[0][1] = Mfma<T>::mma(a0, b1, acc.a[0][1]);
    acc.a[1][0] = Mfma<T>::mma(a1, b0, acc.a[1][0]);
    acc.a[1][1] = Mfma<T>::mma(a1, b1, acc.a[1][1]);
  }
  acc_foreach<T>(acc, [&](int r, int c, T val) { A[(i0 + r) * ld + d0 + c] = val; });
}

template <typename T>
__global__ __launch_bounds__(NTHREADS) void k_potrf_update(T* __restrict__ A, int64_t ld, int64_t k) {
  __shared__ __attribute__((aligned(16))) T smem[SMEM_ELEMS];
  int64_t ii, jj;
  tri_index(blockIdx.x, ii, jj);
  const int64_t i0 = (k + 1 + ii) * TILE, j0 = (k + 1 + jj) * TILE, k0 = k * TILE;
  Acc<T> acc;
  acc.zero();
  gemm_tile<T, KC, KC>(A + i0 * ld + k0, ld, A + j0 * ld + k0, ld, 0, TILE, nullptr, acc, smem);
  acc_foreach<T>(acc, [&](int r, int c, T val) { A[(i0 + r) * ld + j0 + c] -= val; });
}

// zero the strict upper 64x64 tiles of an n x n matrix (n = nt*64) so factors read back clean
template <typename T>
__global__ void k_zero_upper_tiles(T* __restrict__ A, int64_t ld, int64_t n) {
  int64_t i = blockIdx.y * (int64_t)blockDim.y + threadIdx.y;
  int64_t j = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < n && j < n && (j / TILE) > (i / TILE)) A[i * ld + j] = T(0);
}

// ---------------------------------------------------------------------------------------------------
// Triangular inverse X = L^-1 by recursive doubling.  Diagonal 64-blocks of X were written by k_potrf_panel.
// Level with half-size h pairs block p = (X11 at rows r0..r0+h, X22 at rows r1 = r0+h .. min(r1+h, n)):
//   phase 0 : Tw[r1+i][r0+j] =  sum_k L[r1+i][r0+k] X[r0+k][r0+j]      (A = L21 KC ; B = X11 RC, lower: k >= j)
//   phase 1 : X [r1+i][r0+j] = -sum_k X[r1+i][r1+k] Tw[r1+k][r0+j]     (A = X22 KC, lower: k <= i ; B = Tw RC)
// grid = (h/64 column tiles, h/64 row tiles, pairs); tiles beyond n exit.
// ---------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(NTHREADS) void k_trtri_step(const T* __restrict__ L, int64_t ldl, T* __restrict__ X,
                                                         int64_t ldx, T* __restrict__ Tw, int64_t ldt, int64_t n,
                                                         int64_t h, int phase) {
  __shared__ __attribute__((aligned(16))) T smem[SMEM_ELEMS];
  const int64_t p = blockIdx.z;
  const int64_t r0 = 2 * p * h, r1 = r0 + h;
  const int64_t ti = blockIdx.y, tj = blockIdx.x;
  if (r1 + ti * TILE >= n) return;
  Acc<T> acc;
  acc.zero();
  const int64_t i0 = r1 + ti * TILE, j0 = r0 + tj * TILE;
  if (phase == 0) {
    // k local in [tj*64, h)
    gemm_tile<T, KC, RC>(L + i0 * ldl + r0, ldl, X + r0 * ldx + j0, ldx, tj * TILE, h, nullptr, acc, smem);
    acc_foreach<T>(acc, [&](int r, int c, T val) { Tw[(i0 + r) * ldt + j0 + c] = val; });
  } else {
    // k local in [0, (ti+1)*64)
    gemm_tile<T, KC, RC>(X + i0 * ldx + r1, ldx, Tw + r1 * ldt + j0, ldt, 0, (ti + 1) * TILE, nullptr, acc, smem);
    acc_foreach<T>(acc, [&](int r, int c, T val) { X[(i0 + r) * ldx + j0 + c] = -val; });
  }
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
__global__ void k_logdiag_sum(const T* __restrict__ L, int64_t ld, int64_t nvalid, double* __restrict__ out) {
  __shared__ double red[16];
  double s = 0.0;
  for (int64_t i = threadIdx.x; i < nvalid; i += blockDim.x) s += log((double)L[i * ld + i]);
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
