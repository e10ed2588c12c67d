// agp_linalg.h -- dense kernels of the CAVI path for gfx950, all on padded (multiple-of-64) row-major
// matrices so no tile needs a bounds check:
//   k_gemm_nt      C = A B^T with fused row-dot epilogues          (kappa = Knm K^-1, W = kappa L_A^-T, predict var)
//   k_syrk_tn      S = A^T diag(w) A, lower tiles mirrored, fused natural-gradient step on eta2
//   (the fused augmented Cholesky lives in agp_chol.h)
// Reference call sites: src/gpblocks/latentgp.jl:205-215, src/inference/analyticVI.jl:160-180,
// src/inference/inference.jl:25-28 (all LAPACK/BLAS there).
#pragma once
#include "agp_chol.h"
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
//   EPI_HK      no C    ; the hyper-gradient's element-wise pass behind T1 = kappa (Sigma K^-1) (k_hyper_hk, agp_hyper.h) as the
//               epilogue of that product (round 5: one launch and 16 MB of T1 traffic less per hyper-on iteration):
//                 h = rho g_mu_i a_j + rho g_s_i (2 T1_ij - kappa_ij) ;  G_Knm_ij = h - rho g_s_i kappa_ij   -> part1 (ld = ldc)
//                 upart[2 (tile row) + row wave][j] = sum over that wave's 32 rows of g_mu_i kappa_ij          -> part0 (ld = ldp)
//               E = kappa, v = a = K^-1 mu; rows >= hk.B give zeros
enum { EPI_STORE = 0, EPI_KAPPA = 1, EPI_W = 2, EPI_ROWDOT = 3, EPI_EMINUS = 4, EPI_HK = 5 };
template <typename T>
struct HkArgs {
  const T* gmu = nullptr;
  const T* gs = nullptr;
  T rho = T(0);
  int64_t B = 0;
};

template <typename T, int EPI, int KG = 1>
__global__ __launch_bounds__(NTHREADS * KG) void k_gemm_nt(const T* __restrict__ A, int64_t lda, const T* __restrict__ B,
                                                      int64_t ldb, int64_t K, int tri_b, T* __restrict__ C,
                                                      int64_t ldc, const T* __restrict__ E, int64_t lde,
                                                      const T* __restrict__ v, T* __restrict__ part0,
                                                      T* __restrict__ part1, int64_t ldp, HkArgs<T> hk = HkArgs<T>{}) {
  __shared__ __attribute__((aligned(16))) T smem[KG * smem_elems<T>()];
  int64_t bn, bm;  // XCD-aware: every XCD works on a compact block of C tiles (agp_chol.h, xcd_contiguous)
#ifndef AGP_GEMM_XCD
#define AGP_GEMM_XCD 1
#endif
  if (AGP_GEMM_XCD) {
    banded_tile(xcd_contiguous((int64_t)blockIdx.x + (int64_t)blockIdx.y * gridDim.x, (int64_t)gridDim.x * gridDim.y),
                (int64_t)gridDim.x, (int64_t)gridDim.y, bm, bn);
  } else {
    bn = blockIdx.x;
    bm = blockIdx.y;
  }
  const int64_t r0 = bm * TILE, c0 = bn * TILE;
  Acc<T> acc;
  acc.zero();
  int64_t kEnd = tri_b ? ((c0 + TILE) < K ? (c0 + TILE) : K) : K;
  gemm_tile<T, KC, KC, KG>(A + r0 * lda, lda, B + c0 * ldb, ldb, 0, kEnd, nullptr, acc, smem);
  if (KG > 1 && threadIdx.x >= NTHREADS) return;  // the second k-group has handed its partial sums over
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
  if (EPI == EPI_HK) {
    // this thread's values of a column: rows (lane >> 4) + 4 r + 16 mi of its wave's 32; column sums: over r and mi in the thread, then
    // over the four lanes that share a column (xor 16, 32) -- a fixed order
    const int lane = threadIdx.x & 63, wave = (threadIdx.x >> 6) & 3, wm = wave >> 1, wn = wave & 1;
    T ucol[2] = {T(0), T(0)};
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = wm * 32 + mi * 16 + Mfma<T>::row(lane, r), col = wn * 32 + ni * 16 + (lane & 15);
          const int64_t i = r0 + row, j = c0 + col;
          T g = T(0);
          if (i < hk.B) {
            const T k = E[i * lde + j], sg = hk.rho * hk.gs[i], gm = hk.gmu[i];
            const T h = hk.rho * gm * v[j] + sg * (T(2) * acc.a[mi][ni][r] - k);
            g = h - sg * k;
            ucol[ni] += gm * k;
          }
          part1[i * ldc + j] = g;
        }
    // one row of partial sums per (tile row, row wave): 2 M / 64 rows of `part0`, which the consumer adds in order (no barrier here:
    // with two k-groups half of the workgroup has left already)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      T u = ucol[ni];
      u += __shfl_xor(u, 16);
      u += __shfl_xor(u, 32);
      if (lane < 16) part0[(bm * 2 + wm) * ldp + c0 + wn * 32 + ni * 16 + lane] = u;
    }
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

// The same product on 128 x 64 C tiles (gemm_tile_tall, agp_device.h): EPI_STORE and EPI_KAPPA only, M a multiple of 128.
// grid = (N / 64, M / 128).  The K~ partial slices keep their layout (slice = 2 bn + wn, one value per row).
template <typename T, int EPI>
__global__ __launch_bounds__(NTHREADS, 2) void k_gemm_nt_tall(const T* __restrict__ A, int64_t lda, const T* __restrict__ B, int64_t ldb,
                                                              int64_t K, int tri_b, T* __restrict__ C, int64_t ldc,
                                                              const T* __restrict__ E, int64_t lde, T* __restrict__ part0,
                                                              T* __restrict__ part1, int64_t ldp) {
  static_assert(EPI == EPI_STORE || EPI == EPI_KAPPA, "tall tiles: plain store and the kappa epilogue");
  __shared__ __attribute__((aligned(16))) T smem[TallSlab<T>::SMEM];
  int64_t bn, bm;
  banded_tile(xcd_contiguous((int64_t)blockIdx.x + (int64_t)blockIdx.y * gridDim.x, (int64_t)gridDim.x * gridDim.y),
              (int64_t)gridDim.x, (int64_t)gridDim.y, bm, bn);
  const int64_t r0 = bm * 2 * TILE, c0 = bn * TILE;
  AccTall<T> acc;
  acc.zero();
  const int64_t kEnd = tri_b ? ((c0 + TILE) < K ? (c0 + TILE) : K) : K;
  gemm_tile_tall<T>(A + r0 * lda, lda, B + c0 * ldb, ldb, 0, kEnd, acc, smem);
  acc_foreach_tall<T>(acc, [&](int r, int c, T val) {
    C[(r0 + r) * ldc + c0 + c] = val;
    if (EPI == EPI_KAPPA && part1) part1[(r0 + r) * ldc + c0 + c] = val;
  });
  if (EPI == EPI_KAPPA) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wm = wave >> 1, wn = wave & 1;
    T* p0 = part0 + (bn * 2 + wn) * ldp;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = wm * 64 + mi * 16 + Mfma<T>::row(lane, r);
        T s0 = T(0);
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
          const int col = wn * 32 + ni * 16 + (lane & 15);
          s0 += acc.a[mi][ni][r] * E[(r0 + row) * lde + c0 + col];  // (the order of acc_row_reduce: ni ascending, then the 16 lanes)
        }
        s0 = row16_sum(s0);
        if ((lane & 15) == 0) p0[r0 + row] = s0;
      }
  }
}

// C = E - A B^T for a result known to be SYMMETRIC (A = K^-1 - K^-1 (Sigma K^-1): predictions.jl:38 and the hyper-gradient's G_K):
// only the nt (nt + 1) / 2 lower tiles are formed -- half the flops of k_gemm_nt<EPI_EMINUS> -- and mirrored on the way out
// (diagonal tiles: the lower half is the truth).  grid = nt (nt + 1) / 2, row-major triangle order.
template <typename T, int KG = 1>
__global__ __launch_bounds__(NTHREADS * KG) void k_gemm_nt_eminus_sym(const T* __restrict__ A, int64_t lda, const T* __restrict__ B,
                                                                       int64_t ldb, int64_t K, T* __restrict__ C, int64_t ldc,
                                                                       const T* __restrict__ E, int64_t lde) {
  __shared__ __attribute__((aligned(16))) T smem[KG * smem_elems<T>()];
  int64_t ta, tb;
  tri_index(blockIdx.x, ta, tb);
  const int64_t r0 = ta * TILE, c0 = tb * TILE;
  Acc<T> acc;
  acc.zero();
  gemm_tile<T, KC, KC, KG>(A + r0 * lda, lda, B + c0 * ldb, ldb, 0, K, nullptr, acc, smem);
  if (KG > 1 && threadIdx.x >= NTHREADS) return;
  acc_foreach<T>(acc, [&](int r, int c, T val) {
    const int64_t gr = r0 + r, gc = c0 + c;
    if (ta != tb || gc <= gr) {
      const T v = E[gr * lde + gc] - val;
      C[gr * ldc + gc] = v;
      C[gc * ldc + gr] = v;
    }
  });
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
//   SY_PACK  : S -> out as packed lower tiles, block column by block column: tile (ta, tb <= ta) at out + pack_index(ta, tb, nt)
//              * 64*64 (agp_chol.h), row-major inside the tile (the batch-parallel statistics buffer: one triangle travels over
//              xGMI; nt = ldo / 64)
enum { SY_STORE = 0, SY_ETA2 = 1, SY_PACK = 2 };

// what a finished tile (ta, tb) of S does, by mode
template <typename T, int MODE>
__device__ __forceinline__ void syrk_epilogue(Acc<T>& acc, int64_t ta, int64_t tb, T* __restrict__ out, int64_t ldo,
                                              T* __restrict__ eta2, const T* __restrict__ Kinv, int64_t ldm, T lr) {
  const int64_t a0 = ta * TILE, b0 = tb * TILE;
  if (MODE == SY_PACK) {
    T* tp = out + pack_index(ta, tb, ldo / TILE) * (TILE * TILE);  // (ldo = the statistic's order: nt = ldo / 64)
    acc_foreach<T>(acc, [&](int r, int c, T val) { tp[r * TILE + c] = val; });
  } else if (MODE == SY_STORE) {
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

// workgroups of a launch: [0, ntri): lower tiles | riders (eta1 step) | hand-over refill.  (A tail split of multi-round launches --
// the remainder tiles of the last round as k-slices behind the full tiles, finished by a second launch -- was built and measured in
// round 4 and removed in round 5: C3 141 -> 144 us + the finishing launch; docs/DESIGN_LOG.md section 12.)
template <typename T, int MODE, int KG = 1>
__device__ __forceinline__ void syrk_tn_body(const T* __restrict__ A, int64_t lda, int64_t Kdim, const T* __restrict__ w,
                                             int lower_a, T* __restrict__ out, int64_t ldo, T* __restrict__ eta2,
                                             const T* __restrict__ Kinv, int64_t ldm, T lr, int64_t ntri_,
                                             const T* __restrict__ rvec, T* __restrict__ eta1,
                                             const T* __restrict__ kinv_mu0, int64_t nrider, T* __restrict__ fillp,
                                             int64_t fill_used, int64_t fill_stride, int fill_nb, T* smem) {
  const int64_t ntri = ntri_;
  if (fillp && (int64_t)blockIdx.x >= ntri + nrider) {
    // second kind of rider: refill the hand-over slots the factorisation before this launch wrote (agp_chol.h, "self-validating
    // hand-over") with the sentinel, in the shadow of the tile workgroups -- half the chip is idle during this launch anyway
    const int64_t nfb = (int64_t)gridDim.x - ntri - nrider, fb = (int64_t)blockIdx.x - ntri - nrider;
    const T sv = __builtin_bit_cast(T, Sent<T>::bits);
    for (int q = 0; q < fill_nb; ++q) {
      T* dst = fillp + q * fill_stride;
      for (int64_t i = fb * blockDim.x + threadIdx.x; i < fill_used; i += nfb * blockDim.x) dst[i] = sv;
    }
    return;
  }
  if (rvec && (int64_t)blockIdx.x >= ntri) {
    // rider of the fused step: workgroup ntri + j also takes the natural-gradient step on eta1[64 j .. 64 j + 63]
    //   t = A' r (column sums, analyticVI.jl:168) ; eta1 += lr (t + K^-1 mu0 - eta1)
    // (these used to be two kernels of their own, and two launch gaps, on the step's critical path)
    const int64_t c0 = ((int64_t)blockIdx.x - ntri) * TILE;
    const int c = threadIdx.x & 63, grp = threadIdx.x >> 6, ngrp = blockDim.x >> 6;
    T sum = T(0);
    for (int64_t k = grp; k < Kdim; k += ngrp) sum += A[k * lda + c0 + c] * rvec[k];
    smem[grp * TILE + c] = sum;
    __syncthreads();
    if (grp == 0) {
      T t = T(0);
      for (int q = 0; q < ngrp; ++q) t += smem[q * TILE + c];
      if (MODE == SY_PACK) {  // batch-parallel statistics: t itself travels (it is all-reduced before the step is taken)
        eta1[c0 + c] = t;
      } else {
        const T e = eta1[c0 + c];
        eta1[c0 + c] = e + lr * (t + (kinv_mu0 ? kinv_mu0[c0 + c] : T(0)) - e);
      }
    }
    return;
  }
  // (launch order = row-major triangle order.  An XCD-aware order -- contiguous id ranges per XCD over 4 x 4 blocks of tiles --
  // was measured: 25 % less fabric traffic, but the f32 m = 2048 product got 9 % SLOWER and f64 m = 1024 did not move; the
  // operands sit in the Infinity Cache either way, and with the plain order all XCDs stream the same panels at the same time)
  int64_t ta, tb;
  tri_index((int64_t)blockIdx.x, ta, tb);
  const int64_t a0 = ta * TILE, b0 = tb * TILE;
  Acc<T> acc;
  acc.zero();
  gemm_tile<T, RC, RC, KG>(A + a0, lda, A + b0, lda, lower_a ? a0 : 0, Kdim, w, acc, smem);
  if (KG > 1 && threadIdx.x >= NTHREADS) return;
  syrk_epilogue<T, MODE>(acc, ta, tb, out, ldo, eta2, Kinv, ldm, lr);
}

template <typename T, int MODE, int KG = 1>
__global__ __launch_bounds__(NTHREADS * KG) void k_syrk_tn(const T* __restrict__ A, int64_t lda, int64_t Kdim,
                                                      const T* __restrict__ w, int lower_a, T* __restrict__ out,
                                                      int64_t ldo, T* __restrict__ eta2, const T* __restrict__ Kinv,
                                                      int64_t ldm, T lr, int64_t ntri = 0,
                                                      const T* __restrict__ rvec = nullptr, T* __restrict__ eta1 = nullptr,
                                                      const T* __restrict__ kinv_mu0 = nullptr, int64_t nrider = 0,
                                                      T* __restrict__ fillp = nullptr, int64_t fill_used = 0,
                                                      int64_t fill_stride = 0, int fill_nb = 0) {
  // KG = 4 in f64 takes ALL of gfx950's 160 KB of LDS (4 x 40 KB staging areas): nothing else in this kernel may be __shared__,
  // and the instantiation does not exist for smaller-LDS targets
  static_assert((size_t)KG * smem_elems<T>() * sizeof(T) <= 160 * 1024, "k_syrk_tn: staging areas exceed the 160 KB LDS of gfx950");
  __shared__ __attribute__((aligned(16))) T smem[KG * smem_elems<T>()];
  syrk_tn_body<T, MODE, KG>(A, lda, Kdim, w, lower_a, out, ldo, eta2, Kinv, ldm, lr, ntri, rvec, eta1, kinv_mu0, nrider, fillp,
                            fill_used, fill_stride, fill_nb, smem);
}

// ---------------------------------------------------------------------------------------------------
// out = X' X for LOWER-TRIANGULAR X (K^-1 from L^-1, Sigma from chol(-2 eta2)^-1), balanced (round 4).
// k_syrk_tn<SY_STORE> with lower_a gives every lower tile (ta, tb) to one workgroup: tile (0, 0) carries nt 64-deep k-blocks, tile
// (nt-1, .) one -- the launch lasts as long as its longest tile (a CU forms one k-block in ~2.7 us of its fp64 MFMA pipes: 16 of
// them = the 42 us measured at m = 1024), although the whole product is only nt (nt + 1)(nt + 2) / 6 k-blocks (816: 8.6 us of the
// chip).  Here a workgroup takes a UNIT = at most `ch` consecutive k-blocks of one tile.  A tile of one unit is stored at once
// (mirrored); a tile of several units leaves their partial tiles in a workspace and a SECOND launch (k_xtx_bal_reduce) adds them in
// unit order.  (A single launch in which the last unit to arrive -- device-scope counter -- adds the partial tiles was built first:
// every unit then needs a device-scope release, an L2 write-back on this chip, and 444 of them serialise: 121 us.)  Units are
// numbered row of tiles by row of tiles:
//   row ta: (ta + 1) tiles x nu(ta) = ceil((nt - ta) / ch) units, unit index = [rows before] + tb * nu + u.
// ws: nunits x 64 x 64 elements (thread-major partial tiles).  Further workgroups of the first launch (blockIdx.x >= nunits) refill a
// dirty hand-over set like the riders of k_syrk_tn.
// ---------------------------------------------------------------------------------------------------
__host__ __device__ __forceinline__ int64_t xtx_bal_units(int64_t nt, int ch) {
  int64_t s = 0;
  for (int64_t ta = 0; ta < nt; ++ta) s += (ta + 1) * ((nt - ta + ch - 1) / ch);
  return s;
}
template <typename T, typename F>
__device__ __forceinline__ void xtx_store_mirrored(Acc<T>& acc, int64_t ta, int64_t tb, T* __restrict__ out, int64_t ldo, F) {
  const int64_t a0 = ta * TILE, b0 = tb * TILE;
  acc_foreach<T>(acc, [&](int rr, int c, T val) {
    const int64_t gr = a0 + rr, gc = b0 + c;
    if (ta != tb || gc <= gr) {  // diagonal tile: the lower half is the truth, mirrored
      out[gr * ldo + gc] = val;
      out[gc * ldo + gr] = val;
    }
  });
}
template <typename T, int KG>
__global__ __launch_bounds__(NTHREADS * KG) void k_xtx_bal(const T* __restrict__ X, int64_t ld, int64_t n, T* __restrict__ out,
                                                           int64_t ldo, T* __restrict__ ws, int ch, int64_t nunits,
                                                           T* __restrict__ fillp, int64_t fill_used, int64_t fill_stride,
                                                           int fill_nb) {
  __shared__ __attribute__((aligned(16))) T smem[KG * smem_elems<T>()];
  if ((int64_t)blockIdx.x >= nunits) {
    const int64_t nfb = (int64_t)gridDim.x - nunits, fb = (int64_t)blockIdx.x - nunits;
    const T sv = __builtin_bit_cast(T, Sent<T>::bits);
    for (int q = 0; q < fill_nb; ++q) {
      T* dst = fillp + q * fill_stride;
      for (int64_t i = fb * blockDim.x + threadIdx.x; i < fill_used; i += nfb * blockDim.x) dst[i] = sv;
    }
    return;
  }
  const int64_t nt = n / TILE;
  int64_t r = blockIdx.x, ta = 0, nu = 1;
  for (;; ++ta) {
    nu = (nt - ta + ch - 1) / ch;
    const int64_t inrow = (ta + 1) * nu;
    if (r < inrow) break;
    r -= inrow;
  }
  const int64_t tb = r / nu, u = r % nu;
  const int64_t k0 = (ta + u * ch) * TILE, k1 = (k0 + (int64_t)ch * TILE) < n ? (k0 + (int64_t)ch * TILE) : n;
  Acc<T> acc;
  acc.zero();
  gemm_tile<T, RC, RC, KG>(X + ta * TILE, ld, X + tb * TILE, ld, k0, k1, nullptr, acc, smem);
  if (KG > 1 && threadIdx.x >= NTHREADS) return;
  if (nu > 1) {
    T* wp = ws + (int64_t)blockIdx.x * (TILE * TILE);
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int q = 0; q < 4; ++q) wp[((mi * 2 + ni) * 4 + q) * NTHREADS + threadIdx.x] = acc.a[mi][ni][q];
    return;
  }
  xtx_store_mirrored<T>(acc, ta, tb, out, ldo, 0);
}
// second launch: one workgroup per lower tile (row-major triangle order); tiles of one unit are done already.  All partial values
// of an element are fetched side by side (a loop of dependent loads per element made this launch slower than the product)
constexpr int XTX_MAXU = 8;  // units per tile the reduction is written for (the host picks ch so that nt / ch <= 8)
// (Dg != nullptr: one more workgroup, blockIdx.x = nt (nt + 1) / 2, is k_logdiag_sum -- log det from the diagonal factors, with the
//  failure-order word of the K_ZZ refresh --, a launch of its own behind the refresh until round 4)
template <typename T>
__global__ __launch_bounds__(NTHREADS) void k_xtx_bal_reduce(int64_t n, T* __restrict__ out, int64_t ldo,
                                                             const T* __restrict__ ws, int ch, const T* __restrict__ Dg = nullptr,
                                                             int64_t nvalid = 0, double* __restrict__ ld_out = nullptr,
                                                             int32_t* __restrict__ status = nullptr) {
  const int64_t nt = n / TILE;
  if ((int64_t)blockIdx.x >= nt * (nt + 1) / 2) {
    __shared__ double red[16];
    double s = 0.0;
    if (Dg)
      for (int64_t i = threadIdx.x; i < nvalid; i += blockDim.x)
        s += log((double)Dg[(i / TILE) * TILE * TILE + (i % TILE) * (TILE + 1)]);
    s = block_sum<double>(s, red);
    if (Dg && threadIdx.x == 0) {
      ld_out[0] = s;
      if (status && status[1] != 0 && status[3] == 0) status[3] = (status[0] != 0 || status[2] != 0) ? 2 : 1;
    }
    return;
  }
  int64_t ta, tb;
  tri_index(blockIdx.x, ta, tb);
  const int nu = (int)((nt - ta + ch - 1) / ch);
  if (nu <= 1) return;
  int64_t base = 0;
  for (int64_t a = 0; a < ta; ++a) base += (a + 1) * ((nt - a + ch - 1) / ch);
  const T* w0 = ws + (base + tb * nu) * (TILE * TILE) + threadIdx.x;
  Acc<T> acc;
#pragma unroll
  for (int h = 0; h < 2; ++h) {  // two halves of the thread's 16 values: 8 x XTX_MAXU loads in flight
    T p[8][XTX_MAXU];
#pragma unroll
    for (int q = 0; q < 8; ++q)
#pragma unroll
      for (int k = 0; k < XTX_MAXU; ++k) p[q][k] = k < nu ? w0[(int64_t)k * (TILE * TILE) + (h * 8 + q) * NTHREADS] : T(0);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      T v = p[q][0];
#pragma unroll
      for (int k = 1; k < XTX_MAXU; ++k) v += p[q][k];  // unit order (the zeros of absent units change nothing)
      const int e = h * 8 + q;
      acc.a[e >> 3][(e >> 2) & 1][e & 3] = v;
    }
  }
  xtx_store_mirrored<T>(acc, ta, tb, out, ldo, 0);
}

// The fused natural-gradient step of SEVERAL latents (multi-class, multi-output, heteroscedastic models on one GPU) as one
// launch: blockIdx.y selects the latent.  One latent alone fills only ~136 of 256 CUs (m = 1024: nt (nt + 1) / 2 tiles) and
// eight launches in a row cost 8 x 89 us at C4; together the tiles of all latents keep the whole chip busy.
constexpr int SYRK_MAXB = 16;
template <typename T>
struct SyrkBatch {
  const T* A[SYRK_MAXB];      // kappa
  const T* w[SYRK_MAXB];      // rho g2
  T* out[SYRK_MAXB];          // -2 eta2 for the next factorisation
  T* eta2[SYRK_MAXB];
  const T* Kinv[SYRK_MAXB];
  const T* rvec[SYRK_MAXB];   // rho g1
  T* eta1[SYRK_MAXB];
  const T* kinv_mu0[SYRK_MAXB];
};
template <typename T, int KG>
__global__ __launch_bounds__(NTHREADS * KG) void k_syrk_eta_batch(SyrkBatch<T> b, int64_t lda, int64_t Kdim, int64_t ldo,
                                                             int64_t ldm, T lr, int64_t ntri, int64_t nrider,
                                                             T* __restrict__ fillp, int64_t fill_used,
                                                             int64_t fill_stride, int fill_nb) {
  __shared__ __attribute__((aligned(16))) T smem[KG * smem_elems<T>()];
  const int q = blockIdx.y;
  // the hand-over refill riders exist once (in the slice of latent 0)
  if (q != 0 && (int64_t)blockIdx.x >= ntri + nrider) return;
  syrk_tn_body<T, SY_ETA2, KG>(b.A[q], lda, Kdim, b.w[q], 0, b.out[q], ldo, b.eta2[q], b.Kinv[q], ldm, lr, ntri, b.rvec[q],
                               b.eta1[q], b.kinv_mu0[q], nrider, q == 0 ? fillp : (T*)nullptr, fill_used, fill_stride, fill_nb,
                               smem);
}

// eta2 step from an already reduced statistic S (batch-parallel multi-GPU path: S was all-reduced), stored as packed lower
// tiles (SY_PACK layout): grid = nt(nt+1)/2 workgroups of 256 threads.  Diagonal tiles take
// their lower half as the truth and mirror it, exactly like the fused SY_ETA2 epilogue, so a one-rank run of the phase-split
// path lands on the fused path's eta2 bit for bit.
template <typename T>
__global__ __launch_bounds__(256) void k_eta2_from_packed(const T* __restrict__ Sp, int64_t ld, T* __restrict__ eta2,
                                                          const T* __restrict__ Kinv, T* __restrict__ Amat, T lr, int64_t ntri,
                                                          const T* __restrict__ t_red, const T* __restrict__ kinv_mu0,
                                                          T* __restrict__ eta1, int64_t mp) {
  // grid = 4 ntri + ceil(mp / 256): workgroup 4 q + s takes rows 16 s .. 16 s + 15 of tile q (four workgroups per tile: 136
  // tiles alone leave half the chip idle on a kernel that only streams); the last ones take the eta1 step from the reduced
  // t = kappa' (rho g1):  eta1 += lr (t + K^-1 mu0 - eta1)   (analyticVI.jl:160-169, 229-246)
  if ((int64_t)blockIdx.x >= 4 * ntri) {
    const int64_t a = ((int64_t)blockIdx.x - 4 * ntri) * 256 + threadIdx.x;
    if (a < mp) {
      const T e = eta1[a];
      eta1[a] = e + lr * (t_red[a] + (kinv_mu0 ? kinv_mu0[a] : T(0)) - e);
    }
    return;
  }
  const int64_t q = blockIdx.x >> 2;
  const int sub = blockIdx.x & 3;
  int64_t ta, tb;
  tri_index(q, ta, tb);
  const T* tp = Sp + pack_index(ta, tb, mp / TILE) * (TILE * TILE);
  for (int e = sub * 1024 + threadIdx.x; e < (sub + 1) * 1024; e += 256) {
    const int r = e >> 6, c = e & 63;
    if (ta == tb && c > r) continue;
    const int64_t gr = ta * TILE + r, gc = tb * TILE + c;
    T e2 = eta2[gr * ld + gc];
    const T g = -(tp[e] + T(0.5) * Kinv[gr * ld + gc]) - e2;
    e2 += lr * g;
    eta2[gr * ld + gc] = e2;
    eta2[gc * ld + gr] = e2;
    Amat[gr * ld + gc] = T(-2) * e2;
    Amat[gc * ld + gr] = T(-2) * e2;
  }
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

// ys = S x for symmetric dense S (waves 0 .. n - 1, like k_symv) and yt = X x for lower-triangular X (waves n .. 2 n - 1, like
// k_trmv_lower) in ONE launch: mu = Sigma eta1 and v = Xa eta1 behind a factorisation with its inverse (materialize), so that the
// next step with that inverse finds v ready instead of launching k_trmv_lower itself
template <typename T>
__device__ __forceinline__ void symv_trmv_row(int64_t w, int lane, const T* __restrict__ S, const T* __restrict__ X, int64_t ld,
                                              int64_t n, const T* __restrict__ x, T* __restrict__ ys, T* __restrict__ yt) {
  const bool tri = w >= n;
  const int64_t row = tri ? w - n : w;
  const T* M = tri ? X : S;
  const int64_t kend = tri ? row + 1 : n;
  T s = T(0);
  for (int64_t k = lane; k < kend; k += 64) s += M[row * ld + k] * x[k];
  for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o);
  if (lane == 0) (tri ? yt : ys)[row] = s;
}
template <typename T>
__global__ void k_symv_trmv(const T* __restrict__ S, const T* __restrict__ X, int64_t ld, int64_t n, const T* __restrict__ x,
                            T* __restrict__ ys, T* __restrict__ yt) {
  const int64_t w = blockIdx.x * (int64_t)(blockDim.x >> 6) + (threadIdx.x >> 6);
  if (w >= 2 * n) return;
  symv_trmv_row<T>(w, threadIdx.x & 63, S, X, ld, n, x, ys, yt);
}

// (round 4: mu = Sigma eta1 replaced the transposed triangular mat-vec mu = Xa' v, k_trmv_lower_t, which is gone)

// y[i] = sum_j M[i][j] x[j], i < rows (one wave per row)
template <typename T>
__global__ void k_gemv_rows(const T* __restrict__ M, int64_t ld, int64_t rows, int64_t cols, const T* __restrict__ x,
                            T* __restrict__ y) {
  int64_t row = blockIdx.x * (int64_t)(blockDim.x >> 6) + (threadIdx.x >> 6);
  int lane = threadIdx.x & 63;
  if (row >= rows) return;
  T s = T(0);
  for (int64_t k = lane; k < cols; k += 64) s += M[row * ld + k] * x[k];
  for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o);
  if (lane == 0) y[row] = s;
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
// status (optional; the refresh of K_ZZ inside a training loop): [info | infoK | flags | orderK].  When this refresh has latched a
// failure in infoK, orderK records ONCE whether an earlier step had already latched one of its own (2) or not (1): the host reports
// the failure that came FIRST as the root cause -- a non-SPD K_ZZ makes K~ negative afterwards, a negative K~ turns the kernel
// parameters into NaNs and K_ZZ non-SPD afterwards (agp_svgp_check_status).
template <typename T>
__global__ void k_logdiag_sum(const T* __restrict__ Dg, int64_t nvalid, double* __restrict__ out, int32_t* __restrict__ status = nullptr) {
  __shared__ double red[16];
  double s = 0.0;
  for (int64_t i = threadIdx.x; i < nvalid; i += blockDim.x)
    s += log((double)Dg[(i / TILE) * TILE * TILE + (i % TILE) * (TILE + 1)]);
  s = block_sum<double>(s, red);
  if (threadIdx.x == 0) {
    out[0] = s;
    if (status && status[1] != 0 && status[3] == 0) status[3] = (status[0] != 0 || status[2] != 0) ? 2 : 1;
  }
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
