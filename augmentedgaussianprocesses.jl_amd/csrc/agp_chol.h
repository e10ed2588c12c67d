// agp_chol.h -- the augmented, fused blocked Cholesky of the CAVI step (gfx950, wave64, f64/f32 MFMA 16x16x4).
//
//   A = L L'  (n = nt*64, lower, in place; diagonal factors in the side buffer Dg)
//   optional extension rows  E <- E L^-T   ("augmented Cholesky": E = [kappa ; eta1'] gives W = kappa L_A^-T and
//                                           v' = (L_A^-1 eta1)', all that mean_f / var_f need, latentgp.jl:179,189)
//   optional X = L^-1 (separate row kernel, only for Sigma / mu export, the ELBO's Gaussian KL and prediction)
//
// One launch of k_chol_step per block column k (nt launches), 512-thread workgroups (8 waves = 2 per SIMD):
//   P workgroups (one per block row i >= k, then one per extension block): load L_{k,k-1}, L_{i,k-1}; the pending
//       rank-64 update of the diagonal tile and of the own tile as two 64^3 MFMA products spread over the 8 waves;
//       factor the 64x64 diagonal tile (redundantly in every P workgroup: it is the critical path and redundancy removes
//       a launch); b == 0 stores L_kk (Dg) and L_kk^-1, b > 0 forms L_ik = T_ik L_kk^-T by a third MFMA product.
//   U workgroups: trailing update from column k-1 of tiles (i, j), j > k (one 64^3 MFMA product each).
//
// Diagonal-tile factorisation = Gauss-Jordan on [A | I] without pivoting, which yields L and L^-1 together.  It is a
// 64-long dependent chain, so everything is organised around latency (measured: LDS write->barrier->read 180 cycles,
// dependent f64 op ~25 cycles, v_rcp_f64 + one Newton step 1.8e-15 accurate):
//   * FOUR columns per round (16 rounds, one barrier each): owners publish the 4 raw panel columns and the 4 M rows;
//   * every thread of waves 0-3 redundantly LDL'-factors the 4x4 pivot block, transforms the panel entries it needs and
//     applies the rank-4 update to its cyclic 4x4 sub-blocks of A and of M; triangular skipping is compile-time per
//     16-column group.
#pragma once
#include "agp_cavi.h"  // LikParams, rowstats_finish: the CAVI step's task-graph launch finishes its rows itself (EpiArgs)
#include "agp_device.h"

namespace agp {

constexpr int LDP = TILE + 2;   // 66: [r][k] stride for 64-deep LDS tiles (conflict-free MFMA fragment reads)
#ifndef AGP_PIVOT_ALG
#define AGP_PIVOT_ALG 0  // how the 4x4 pivot blocks of the tile factorisation are factored (see chol_rounds)
#endif
constexpr int CHOL_THREADS = 512;

__device__ __forceinline__ double rcp1(double p) {
  double r = __builtin_amdgcn_rcp(p);
  return fma(r, fma(-p, r, 1.0), r);
}
// 1 / sqrt(p), p > 0: hardware estimate + two Newton steps (the IEEE sqrt-then-divide sequence is ~3x longer and sits on the
// critical path at the end of every 32x32 elimination); result within 1-2 ulp
__device__ __forceinline__ double rsqrt1(double p) {
  double y = __builtin_amdgcn_rsq(p);
  const double h = 0.5 * p;
  y = y * fma(-h * y, y, 1.5);
  y = y * fma(-h * y, y, 1.5);
  return y;
}
__device__ __forceinline__ float rsqrt1(float p) {
  float y = __builtin_amdgcn_rsqf(p);
  const float h = 0.5f * p;
  y = y * fmaf(-h * y, y, 1.5f);
  return y;
}
__device__ __forceinline__ float rcp1(float p) {
  float r = __builtin_amdgcn_rcpf(p);
  return fmaf(r, fmaf(-p, r, 1.0f), r);
}

// global 64x64 tile (row-major, leading dimension ld) -> LDS [r*LDP + c], 16-byte loads, NT threads
template <typename T, int NT>
__device__ __forceinline__ void load_tile_lds(const T* __restrict__ G, int64_t ld, T* S) {
  typedef typename Mfma<T>::vec_t vec_t;
  constexpr int VEC = Mfma<T>::VEC, NV = TILE / VEC;
#pragma unroll
  for (int v = 0; v < TILE * NV / NT; ++v) {
    int vi = threadIdx.x + v * NT;
    int r = vi / NV, cv = vi % NV;
    vec_t x = *reinterpret_cast<const vec_t*>(G + (int64_t)r * ld + cv * VEC);
#pragma unroll
    for (int e = 0; e < VEC; ++e) S[r * LDP + cv * VEC + e] = x[e];
  }
}

// ---- 8-wave 64x64x64 product from LDS: wave w owns rows (w>>2)*32 + {0,16} + .., cols (w&3)*16 + .. ----
template <typename T>
struct Acc8 {
  typename Mfma<T>::acc_t a[2];
  __device__ __forceinline__ void zero() {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) a[i][r] = T(0);
  }
};

// acc += As(64 x 64, [r][k] stride LDP) * Bs(64 x 64 given as [c][k] stride LDP)^T
template <typename T>
__device__ __forceinline__ void mma8(const T* As, const T* Bs, Acc8<T>& acc) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wm = wave >> 2, wn = wave & 3;
#pragma unroll 4
  for (int kk = 0; kk < TILE / 4; ++kk) {
    T a0 = As[(wm * 32 + (lane & 15)) * LDP + kk * 4 + (lane >> 4)];
    T a1 = As[(wm * 32 + 16 + (lane & 15)) * LDP + kk * 4 + (lane >> 4)];
    T b0 = Bs[(wn * 16 + (lane & 15)) * LDP + kk * 4 + (lane >> 4)];
    acc.a[0] = Mfma<T>::mma(a0, b0, acc.a[0]);
    acc.a[1] = Mfma<T>::mma(a1, b0, acc.a[1]);
  }
}

// acc -= As * Bs^T  (the subtraction rides on the negated A operand)
template <typename T>
__device__ __forceinline__ void mma8_sub(const T* As, const T* Bs, Acc8<T>& acc) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wm = wave >> 2, wn = wave & 3;
#pragma unroll 4
  for (int kk = 0; kk < TILE / 4; ++kk) {
    T a0 = -As[(wm * 32 + (lane & 15)) * LDP + kk * 4 + (lane >> 4)];
    T a1 = -As[(wm * 32 + 16 + (lane & 15)) * LDP + kk * 4 + (lane >> 4)];
    T b0 = Bs[(wn * 16 + (lane & 15)) * LDP + kk * 4 + (lane >> 4)];
    acc.a[0] = Mfma<T>::mma(a0, b0, acc.a[0]);
    acc.a[1] = Mfma<T>::mma(a1, b0, acc.a[1]);
  }
}

// visit the accumulator elements of this thread: f(row_in_tile, col_in_tile, value&) ; ext-vector lanes cannot bind to
// references, so each element goes through a scalar temporary
template <typename T, typename F>
__device__ __forceinline__ void acc8_foreach(Acc8<T>& acc, F f) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wm = wave >> 2, wn = wave & 3;
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      T t = acc.a[mi][r];
      f(wm * 32 + mi * 16 + Mfma<T>::row(lane, r), wn * 16 + (lane & 15), t);
      acc.a[mi][r] = t;
    }
}

// ---- 256-thread variant (the on-demand X row kernel keeps 4-wave workgroups because it uses gemm_tile) ----
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

// ---------------------------------------------------------------------------------------------------
// Diagonal-tile factorisation.  Waves 0-3 (256 threads, cyclic 4x4 ownership: R = ti + 16 r, C = tj + 16 c) hold BOTH
// the A sub-blocks and the sub-blocks of the running inverse M; waves 4-7 of the 512-thread workgroup only join the
// barriers (they exist for the MFMA products around the factorisation).  Per round of 4 columns, with ONE barrier:
//   owners publish the 4 raw panel columns + the 4 M rows; every active thread redundantly LDL'-factors the 4x4 pivot
//   block, transforms the panel entries of its own rows / columns / M columns, and applies the rank-4 update.
// (Measured alternatives, tools/bench_diag.py, f64 us per tile: one column per barrier 20-21; this scheme 14.5; pivot
//  block + transforms done once by a dedicated wave and shared through a second barrier 16.7 -- the serial LDL' chain in
//  a single wave costs more than the redundant work it saves.)
// Scratch: PL[2][4][64] panel columns, MW[2][4][64] M rows (double-buffered: round r+1 publishes while stragglers of
// round r may still be reading).
// ---------------------------------------------------------------------------------------------------
constexpr int SC_ELEMS = 2 * 2 * 4 * TILE;

// hook: called once per round by every thread right after the round's barrier (the idle waves 4-7 of a 512-thread workgroup
// can step a side job there -- the chain workgroup of k_chol_dag prefetches its next two tiles this way)
struct NoHook {
  __device__ __forceinline__ void operator()(int) const {}
};

// PIV selects how the 4x4 pivot block is factored (identical results up to rounding):
//   0  LDL' column by column: every reciprocal waits for the previous column's update (dependent depth ~22 operations)
//   1  fraction-free (Bareiss) minors: n_ij = d00 d_ij - d_i0 d_j0, p_ij = (n11 n_ij - n_i1 n_j1) / d00, M4 = (p22 p33 - p32^2) / n11
//      are the leading-minor numerators of the same elimination, so the four reciprocals 1/d00, 1/M2, 1/M3, 1/M4 no longer wait
//      for one another's multipliers (depth ~13); pivots = M_q / M_(q-1), multipliers l21 = n21 / M2, l32 = p32 / M3, ...
template <typename T, int J, int NS, typename H, int PIV = 0>
__device__ __forceinline__ void chol_rounds(T (&a)[NS][NS], T (&g)[NS][NS], T* sc, T* piv, const bool act, const int ti,
                                            const int tj, H& hook, const int hbase) {
#pragma unroll 1
  for (int rr = 0; rr < 4; ++rr) {
    const int jj0 = rr * 4, j0 = J * 16 + jj0;
    T* P = sc + (rr & 1) * 4 * TILE;
    T* Mw = sc + 2 * 4 * TILE + (rr & 1) * 4 * TILE;
    if (act && PIV != 4) {  // (PIV 4, 5: timing experiments -- barrier only / publish + barrier only)
      const int qc = tj - jj0, qr = ti - jj0;
      if (qc >= 0 && qc < 4) {
#pragma unroll
        for (int r = J; r < NS; ++r) {
          T v = a[r][J];
          if (r == J) v = (ti >= jj0) ? v : T(0);  // rows above the panel are dead
          P[qc * TILE + ti + 16 * r] = v;
        }
      }
      if (qr >= 0 && qr < 4) {
#pragma unroll
        for (int c = 0; c <= J; ++c) Mw[qr * TILE + tj + 16 * c] = g[J][c];
      }
    }
    __syncthreads();
    hook(hbase + rr);
    if (!act) continue;
    if (PIV == 4 || PIV == 5) continue;
    // ---- 4x4 pivot block, LDL' (every active thread, redundantly) ----
    const T d00 = P[j0], d10 = P[j0 + 1], d20 = P[j0 + 2], d30 = P[j0 + 3];
    T d11 = P[TILE + j0 + 1], d21 = P[TILE + j0 + 2], d31 = P[TILE + j0 + 3];
    T d22 = P[2 * TILE + j0 + 2], d32 = P[2 * TILE + j0 + 3], d33 = P[3 * TILE + j0 + 3];
    T r0, r1, r2, r3, l10, l20, l30, l21, l31, l32;
    if (PIV == 2) {  // TIMING EXPERIMENT ONLY (wrong results): no pivot-block factorisation at all -- what the rest of a round costs
      r0 = r1 = r2 = r3 = T(1);
      l10 = d10;
      l20 = d20;
      l30 = d30;
      l21 = d21;
      l31 = d31;
      l32 = d32;
    } else if (PIV == 1) {
      r0 = rcp1(d00);
      const T n11 = fma(d00, d11, -(d10 * d10)), n21 = fma(d00, d21, -(d20 * d10)), n31 = fma(d00, d31, -(d30 * d10));
      const T n22 = fma(d00, d22, -(d20 * d20)), n32 = fma(d00, d32, -(d30 * d20)), n33 = fma(d00, d33, -(d30 * d30));
      const T rm2 = rcp1(n11);
      const T p22 = fma(n11, n22, -(n21 * n21)) * r0, p32 = fma(n11, n32, -(n31 * n21)) * r0;
      const T p33 = fma(n11, n33, -(n31 * n31)) * r0;
      const T rm3 = rcp1(p22);
      const T m4 = fma(p22, p33, -(p32 * p32)) * rm2;
      const T rm4 = rcp1(m4);
      l10 = d10 * r0;
      l20 = d20 * r0;
      l30 = d30 * r0;
      l21 = n21 * rm2;
      l31 = n31 * rm2;
      l32 = p32 * rm3;
      r1 = d00 * rm2;
      r2 = n11 * rm3;
      r3 = p22 * rm4;
      d11 = n11 * r0;   // the pivots themselves (positivity check, final scaling)
      d22 = p22 * rm2;
      d33 = m4 * rm3;
    } else {
      r0 = rcp1(d00);
      l10 = d10 * r0;
      l20 = d20 * r0;
      l30 = d30 * r0;
      d11 = fma(-l10, d10, d11);
      d21 = fma(-l20, d10, d21);
      d31 = fma(-l30, d10, d31);
      d22 = fma(-l20, d20, d22);
      d32 = fma(-l30, d20, d32);
      d33 = fma(-l30, d30, d33);
      r1 = rcp1(d11);
      l21 = d21 * r1;
      l31 = d31 * r1;
      d22 = fma(-l21, d21, d22);
      d32 = fma(-l31, d21, d32);
      d33 = fma(-l31, d31, d33);
      r2 = rcp1(d22);
      l32 = d32 * r2;
      d33 = fma(-l32, d32, d33);
      r3 = rcp1(d33);
    }
    if (ti == 0 && tj == 0) {
      piv[j0] = d00;
      piv[j0 + 1] = d11;
      piv[j0 + 2] = d22;
      piv[j0 + 3] = d33;
    }
    if (PIV == 3) {  // TIMING EXPERIMENT ONLY (wrong results): publish + barrier + pivot-block LDL', no transforms / updates
      a[J][J] += r3 * T(1e-30) + l32 * T(1e-30);
      continue;
    }
    // ---- panel transforms for my columns / my M columns, then row by row: multipliers + rank-4 update ----
    T u[4][NS], mw[4][NS];  // [q][c]
#pragma unroll
    for (int c = J; c < NS; ++c) {
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
    for (int r = J; r < NS; ++r) {
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

// In: bufA = SPD tile [R*LDP + C] (lower triangle valid).  Out: bufA = L (strict upper zero), bufB = L^-1.
// All threads of the (256- or 512-thread) workgroup must call.  info: first non-positive pivot (1-based global column).
template <typename T>
__device__ __forceinline__ void factor_diag_tile512(T* bufA, T* bufB, T* sc, T* piv, int32_t* info, int64_t col0,
                                                    int64_t nvalid) {
  const int tid = threadIdx.x;
  const bool act = tid < 256;
  const int ti = (tid & 255) >> 4, tj = tid & 15;
  NoHook nohook;
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
  __syncthreads();
  chol_rounds<T, 0, 4, NoHook>(a, g, sc, piv, act, ti, tj, nohook, 0);
  chol_rounds<T, 1, 4, NoHook>(a, g, sc, piv, act, ti, tj, nohook, 0);
  chol_rounds<T, 2, 4, NoHook>(a, g, sc, piv, act, ti, tj, nohook, 0);
  chol_rounds<T, 3, 4, NoHook>(a, g, sc, piv, act, ti, tj, nohook, 0);
  __syncthreads();  // piv complete
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
  if (act) {
    T rsC[4], rsR[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      T pc = piv[tj + 16 * q], pr = piv[ti + 16 * q];
      rsC[q] = rsqrt1(pc > T(0) ? pc : T(1));
      rsR[q] = rsqrt1(pr > T(0) ? pr : T(1));
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        int R = ti + 16 * r, Cc = tj + 16 * c;
        bufA[R * LDP + Cc] = (R >= Cc) ? a[r][c] * rsC[c] : T(0);
        bufB[R * LDP + Cc] = (R >= Cc) ? g[r][c] * rsR[r] : T(0);
      }
  }
  __syncthreads();
}

// ---------------------------------------------------------------------------------------------------
// Two-level variant of the diagonal-tile factorisation: the 64x64 tile is split into 32x32 blocks
//   [A11 . ; A21 A22] :  (L11, X11) = elim(A11) ; L21 = A21 X11' ; S = A22 - L21 L21' ; (L22, X22) = elim(S) ;
//   X21 = -X22 (L21 X11)
// The two eliminations use the same 4-column rounds with 2x2 cyclic ownership (about half the per-round work of the
// 4x4 ownership: the redundant per-thread transforms shrink with the sub-block count); the glue is four 32^3 MFMA products.
// ---------------------------------------------------------------------------------------------------
template <typename T, bool TB, int LDB = LDP>
__device__ __forceinline__ typename Mfma<T>::acc_t mma_blk32(const T* As, const T* Bs, typename Mfma<T>::acc_t acc,
                                                             int wr, int wc, int lane, bool negA) {
  // acc(16x16 tile (wr, wc) of a 32x32 result) += sum_k A[r][k] * B(k, c) ; TB: B(k,c) = Bs[k*LDB + c] else Bs[c*LDB + k]
#pragma unroll
  for (int kk = 0; kk < 8; ++kk) {
    T a = As[(wr * 16 + (lane & 15)) * LDP + kk * 4 + (lane >> 4)];
    if (negA) a = -a;
    T b = TB ? Bs[(kk * 4 + (lane >> 4)) * LDB + wc * 16 + (lane & 15)]
             : Bs[(wc * 16 + (lane & 15)) * LDB + kk * 4 + (lane >> 4)];
    acc = Mfma<T>::mma(a, b, acc);
  }
  return acc;
}

// eliminate the 32x32 block at (o, o) of bufA (lower valid): L -> bufA block, L^-1 -> bufB block ; all threads call
template <typename T, typename H, int PIV = 0>
__device__ __forceinline__ void elim_block32(T* bufA, T* bufB, int o, T* sc, T* piv, const bool act, const int ti,
                                             const int tj, H& hook) {
  T a[2][2], g[2][2];
#pragma unroll
  for (int r = 0; r < 2; ++r)
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      int R = ti + 16 * r, Cc = tj + 16 * c;
      int lo = R >= Cc ? R : Cc, hi = R >= Cc ? Cc : R;
      a[r][c] = bufA[(o + lo) * LDP + o + hi];
      g[r][c] = (R == Cc) ? T(1) : T(0);
    }
  __syncthreads();
  chol_rounds<T, 0, 2, H, PIV>(a, g, sc, piv + o, act, ti, tj, hook, 0);
  chol_rounds<T, 1, 2, H, PIV>(a, g, sc, piv + o, act, ti, tj, hook, 4);
  __syncthreads();
  if (act) {
    T rsC[2], rsR[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      T pc = piv[o + tj + 16 * q], pr = piv[o + ti + 16 * q];
      rsC[q] = rsqrt1(pc > T(0) ? pc : T(1));
      rsR[q] = rsqrt1(pr > T(0) ? pr : T(1));
    }
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        int R = ti + 16 * r, Cc = tj + 16 * c;
        bufA[(o + R) * LDP + o + Cc] = (R >= Cc) ? a[r][c] * rsC[c] : T(0);
        bufB[(o + R) * LDP + o + Cc] = (R >= Cc) ? g[r][c] * rsR[r] : T(0);
      }
  }
  __syncthreads();
}

template <typename T, typename H, int PIV = AGP_PIVOT_ALG>
__device__ __forceinline__ void factor_diag_tile_2lvl(T* bufA, T* bufB, T* sc, T* piv, int32_t* info, int64_t col0,
                                                      int64_t nvalid, H& hook) {
  const int tid = threadIdx.x;
  const bool act = tid < 256;
  const int ti = (tid & 255) >> 4, tj = tid & 15;
  const int lane = tid & 63, wave = tid >> 6, wr = (wave >> 1) & 1, wc = wave & 1;
  const bool mw = wave < 4;  // the four waves that run the 32^3 MFMA products (one 16x16 result tile each)
  typedef typename Mfma<T>::acc_t acc_t;
  NoHook nohook;
  elim_block32<T, NoHook, PIV>(bufA, bufB, 0, sc, piv, act, ti, tj, nohook);
  // L21 = A21 X11' -> scratch in bufB[32:64, 0:32] (free until X21 is formed at the end): no barrier between reading A21 and
  // writing the result.  The other waves clear the upper-right blocks meanwhile (input garbage in bufA, nothing yet in bufB).
  acc_t acc;
#pragma unroll
  for (int r = 0; r < 4; ++r) acc[r] = T(0);
  if (mw) {
    acc = mma_blk32<T, false>(bufA + 32 * LDP, bufB, acc, wr, wc, lane, false);
#pragma unroll
    for (int r = 0; r < 4; ++r)
      bufB[(32 + wr * 16 + Mfma<T>::row(lane, r)) * LDP + wc * 16 + (lane & 15)] = acc[r];
  } else {  // (workgroups of 512 threads: all callers)
    for (int e = tid - 256; e < 32 * 32; e += (int)blockDim.x - 256) {
      bufA[(e >> 5) * LDP + 32 + (e & 31)] = T(0);
      bufB[(e >> 5) * LDP + 32 + (e & 31)] = T(0);
    }
  }
  __syncthreads();
  // L21 into its home, and S = A22 - L21 L21' in place on the three 16x16 tiles on and below the diagonal (the elimination reads
  // the lower triangle only ; every wave reads and writes its own tile of A22, L21 comes from the scratch: no barrier in between)
  if (mw) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
      bufA[(32 + wr * 16 + Mfma<T>::row(lane, r)) * LDP + wc * 16 + (lane & 15)] = acc[r];
    if (wr >= wc) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        acc[r] = bufA[(32 + wr * 16 + Mfma<T>::row(lane, r)) * LDP + 32 + wc * 16 + (lane & 15)];
      acc = mma_blk32<T, false>(bufB + 32 * LDP, bufB + 32 * LDP, acc, wr, wc, lane, true);
#pragma unroll
      for (int r = 0; r < 4; ++r)
        bufA[(32 + wr * 16 + Mfma<T>::row(lane, r)) * LDP + 32 + wc * 16 + (lane & 15)] = acc[r];
    }
  }
  __syncthreads();
  elim_block32<T, H, PIV>(bufA, bufB, 32, sc, piv, act, ti, tj, hook);  // hook rounds 0..7 of the second half
  // P = L21 X11  -> sc (32 x 32, free after the rounds)
#pragma unroll
  for (int r = 0; r < 4; ++r) acc[r] = T(0);
  if (mw) {
    acc = mma_blk32<T, true>(bufA + 32 * LDP, bufB, acc, wr, wc, lane, false);
#pragma unroll
    for (int r = 0; r < 4; ++r) sc[(wr * 16 + Mfma<T>::row(lane, r)) * 32 + wc * 16 + (lane & 15)] = acc[r];
  }
  __syncthreads();
  // X21 = -X22 P   -> bufB[32:64, 0:32] ; bad pivots reported next to it
#pragma unroll
  for (int r = 0; r < 4; ++r) acc[r] = T(0);
  if (mw) {
    acc = mma_blk32<T, true, 32>(bufB + 32 * LDP + 32, sc, acc, wr, wc, lane, true);
#pragma unroll
    for (int r = 0; r < 4; ++r) bufB[(32 + wr * 16 + Mfma<T>::row(lane, r)) * LDP + wc * 16 + (lane & 15)] = acc[r];
  }
  if (blockDim.x <= 256) {  // nobody was free to clear the upper-right blocks earlier
    for (int e = tid; e < 32 * 32; e += blockDim.x) {
      bufA[(e >> 5) * LDP + 32 + (e & 31)] = T(0);
      bufB[(e >> 5) * LDP + 32 + (e & 31)] = T(0);
    }
  }
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
  __syncthreads();
}

template <typename T>
__device__ __forceinline__ void factor_diag_tile_2lvl(T* bufA, T* bufB, T* sc, T* piv, int32_t* info, int64_t col0,
                                                      int64_t nvalid) {
  NoHook nohook;
  factor_diag_tile_2lvl<T, NoHook>(bufA, bufB, sc, piv, info, col0, nvalid, nohook);
}

// ---------------------------------------------------------------------------------------------------
// Panel variant of the diagonal-tile factorisation (round 2): the dependent chain runs in ONE wave with the tile's rows in its
// lanes, no LDS round trip and no barrier inside a 16-column panel.
//   for each 16-column panel s (c0 = 16 s):
//     wave 0, lane r = row r, a[0..15] = its entries of the panel columns.  Column j: the pivot and the entries of column j
//       that the rank-1 update needs come out of the lanes with v_readlane (wave-uniform SGPR operands of the v_fma);
//       a[c] -= (a[j] / d_j) * a_j(row c0 + c).  The raw column u_j (= L_j sqrt(d_j)) goes to bufA as soon as it is final,
//       1 / d_j to rinv.
//     barrier ; the other waves apply the panel to the remaining columns, one 16x16 MFMA tile each (k = 16) ; barrier
//   L^-1 = D^-1/2 (U D^-1)^-1: the four 16x16 unit-triangular diagonal blocks by substitution (one wave each, a column per
//   lane), the off-diagonal blocks by the usual two-level products on MFMA, then the scaling of L and L^-1.
// Same arithmetic as the rounds above up to the order of the updates (LDL' without pivoting, raw pivots in piv).
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ double lane_bcast(double v, int l) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ float lane_bcast(float v, int l) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}

// one 16x16 result tile: acc += (neg ? -1 : 1) * sum_{k < kend} A(r, k) sa[k] B(k, c)
//   A(r, k) = Ar[r*LDP + k] ; sa == nullptr: no scaling ; B(k, c) = NN ? Bp[k*LDP + c] : Bp[c*LDP + k] ; kend a multiple of 8
template <typename T, bool NN>
__device__ __forceinline__ typename Mfma<T>::acc_t mma16(const T* Ar, const T* sa, const T* Bp, int kend,
                                                         typename Mfma<T>::acc_t acc, bool neg, int lane) {
  const int lr = lane & 15, lk = lane >> 4;
  typename Mfma<T>::acc_t p1;
#pragma unroll
  for (int r = 0; r < 4; ++r) p1[r] = T(0);
#pragma unroll 2
  for (int kk = 0; kk < kend; kk += 8) {
    T a0 = Ar[lr * LDP + kk + lk], a1 = Ar[lr * LDP + kk + 4 + lk];
    if (sa) {
      a0 *= sa[kk + lk];
      a1 *= sa[kk + 4 + lk];
    }
    if (neg) {
      a0 = -a0;
      a1 = -a1;
    }
    const T b0 = NN ? Bp[(kk + lk) * LDP + lr] : Bp[lr * LDP + kk + lk];
    const T b1 = NN ? Bp[(kk + 4 + lk) * LDP + lr] : Bp[lr * LDP + kk + 4 + lk];
    acc = Mfma<T>::mma(a0, b0, acc);
    p1 = Mfma<T>::mma(a1, b1, p1);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) acc[r] += p1[r];
  return acc;
}

template <typename T>
__device__ __forceinline__ typename Mfma<T>::acc_t acc_load16(const T* P, int lane) {
  typename Mfma<T>::acc_t acc;
#pragma unroll
  for (int r = 0; r < 4; ++r) acc[r] = P[Mfma<T>::row(lane, r) * LDP + (lane & 15)];
  return acc;
}
template <typename T>
__device__ __forceinline__ void acc_store16(T* P, typename Mfma<T>::acc_t acc, int lane) {
#pragma unroll
  for (int r = 0; r < 4; ++r) P[Mfma<T>::row(lane, r) * LDP + (lane & 15)] = acc[r];
}

// sc layout of this variant: rinv[64] | rs[64]
template <typename T, typename H, int CUT = 0>
__device__ __forceinline__ void factor_diag_tile_panel(T* bufA, T* bufB, T* sc, T* piv, int32_t* info, int64_t col0,
                                                       int64_t nvalid, H& hook) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  T* rinv = sc;
  T* rs = sc + TILE;
  typedef typename Mfma<T>::acc_t acc_t;
#pragma unroll 1
  for (int s = 0; s < 4; ++s) {
    const int c0 = 16 * s;
    if (wave == 0 && CUT != 2) {
      // no masks anywhere: rows above the pivot and the upper triangle of the diagonal block carry garbage that nothing
      // reads (the lanes read below are those of rows >= c0 + j ; the final scaling selects the lower triangle)
      T a[16], dv[16], rv[16];
#pragma unroll
      for (int c = 0; c < 16; ++c) a[c] = bufA[lane * LDP + c0 + c];
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const T d = lane_bcast(a[j], c0 + j);
        const T r = rcp1(d);
        const T l = a[j] * r;
        bufA[lane * LDP + c0 + j] = a[j];
        dv[j] = d;
        rv[j] = r;
#pragma unroll
        for (int c = j + 1; c < 16; ++c) a[c] = fma(-l, lane_bcast(a[j], c0 + c), a[c]);
      }
      if (lane == 0) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          piv[c0 + j] = dv[j];
          rinv[c0 + j] = rv[j];
        }
      }
    }
    __syncthreads();
    hook(2 * s);
    if (s == 3) break;
    {  // trailing tiles (tr, tc), s < tc <= tr <= 3, next panel's first
      const int nrem = 3 - s;  // remaining 16-blocks
      const int ntile = nrem * (nrem + 1) / 2;
      if (wave < ntile) {
        int tc = s + 1, tr = s + 1 + wave;  // column-major order over the lower triangle
        if (tr > 3) {
          int w = wave - nrem;
          tc = s + 2;
          tr = s + 2 + w;
          if (tr > 3) {
            tc = s + 3;
            tr = 3;
          }
        }
        T* Ct = bufA + (tr * 16) * LDP + tc * 16;
        acc_t acc = acc_load16<T>(Ct, lane);
        acc = mma16<T, false>(bufA + (tr * 16) * LDP + c0, rinv + c0, bufA + (tc * 16) * LDP + c0, 16, acc, true, lane);
        acc_store16<T>(Ct, acc, lane);
      }
    }
    __syncthreads();
    hook(2 * s + 1);
  }
  if (CUT == 1 || CUT == 2) return;  // timing experiments: panels + trailing updates only / without the eliminations
  // ---- inverse of the unit-triangular factor U D^-1: diagonal 16x16 blocks by substitution, column lane & 15 per lane ----
  if (wave < 4) {
    const int o = 16 * wave, c = lane & 15;
    T m[16], w[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      T acc = (i == c) ? T(1) : T(0);
#pragma unroll
      for (int k = 0; k < i; ++k) acc = fma(-bufA[(o + i) * LDP + o + k], w[k], acc);
      m[i] = acc;
      w[i] = acc * rinv[o + i];
    }
    if (lane < 16) {
#pragma unroll
      for (int i = 0; i < 16; ++i) bufB[(o + i) * LDP + o + c] = m[i];
    }
  } else {
    // blocks (0,1) and (2,3) of the inverse are read as zeros by the 32-level products ; rs = 1 / sqrt(pivot)
    const int t = tid - 256;
    for (int e = t; e < 2 * 256; e += 256) {
      const int b = e >> 8, r = (e >> 4) & 15, cc = e & 15;
      bufB[(32 * b + r) * LDP + 32 * b + 16 + cc] = T(0);
    }
    if (t < TILE) {
      const T pv = piv[t];
      rs[t] = rsqrt1(pv > T(0) ? pv : T(1));
    }
  }
  __syncthreads();
  hook(7);
  if (CUT == 3) return;  // timing experiment: up to the substitutions
  // M_10 = -M_11 (L1_10 M_00), M_32 = -M_33 (L1_32 M_22) ; Q in the upper-right corner of bufB (scratch)
  acc_t acc;
  if (wave < 2) {
    const int o = 32 * wave;
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = T(0);
    acc = mma16<T, true>(bufA + (o + 16) * LDP + o, rinv + o, bufB + o * LDP + o, 16, acc, false, lane);
    acc_store16<T>(bufB + 32 + 16 * wave, acc, lane);
  }
  __syncthreads();
  if (wave < 2) {
    const int o = 32 * wave;
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = T(0);
    acc = mma16<T, true>(bufB + (o + 16) * LDP + o + 16, (const T*)nullptr, bufB + 32 + 16 * wave, 16, acc, true, lane);
    acc_store16<T>(bufB + (o + 16) * LDP + o, acc, lane);
  }
  __syncthreads();
  // P = L1_[2:4][0:2] M_[0:2][0:2] -> bufB[0:32, 32:64] ; M_[2:4][0:2] = -M_[2:4][2:4] P
  const int wr = (wave >> 1) & 1, wc = wave & 1;
  if (wave < 4) {
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = T(0);
    acc = mma16<T, true>(bufA + (32 + 16 * wr) * LDP, rinv, bufB + 16 * wc, 32, acc, false, lane);
    acc_store16<T>(bufB + (16 * wr) * LDP + 32 + 16 * wc, acc, lane);
  }
  __syncthreads();
  if (wave < 4) {
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = T(0);
    acc = mma16<T, true>(bufB + (32 + 16 * wr) * LDP + 32, (const T*)nullptr, bufB + 32 + 16 * wc, 32, acc, true, lane);
    acc_store16<T>(bufB + (32 + 16 * wr) * LDP + 16 * wc, acc, lane);
  }
  __syncthreads();
  if (CUT == 4) return;  // timing experiment: without the final scaling
  // scaling: L = U D^-1/2 (columns), L^-1 = D^-1/2 M (rows) ; strict upper parts zero ; bad pivots
  for (int e = tid; e < TILE * TILE; e += CHOL_THREADS) {
    const int R = e >> 6, Cc = e & 63;
    const bool low = R >= Cc;
    bufA[R * LDP + Cc] = low ? bufA[R * LDP + Cc] * rs[Cc] : T(0);
    bufB[R * LDP + Cc] = low ? bufB[R * LDP + Cc] * rs[R] : T(0);
  }
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
  __syncthreads();
}

__device__ __forceinline__ void tri_index(int64_t idx, int64_t& ti, int64_t& tj) {
  // idx = ti*(ti+1)/2 + tj, tj <= ti
  int64_t t = (int64_t)((sqrt(8.0 * (double)idx + 1.0) - 1.0) * 0.5);
  while (t * (t + 1) / 2 > idx) --t;
  while ((t + 1) * (t + 2) / 2 <= idx) ++t;
  ti = t;
  tj = idx - t * (t + 1) / 2;
}

// Packed lower tiles of the batch-parallel statistics (SY_PACK): block column by block column -- tile (ta, tb <= ta) of an nt x nt
// grid at index tb nt - tb (tb - 1) / 2 + (ta - tb) -- so that a range of block columns is one contiguous range of the buffer: the
// all-reduce can travel, and arrive, in column groups (AGP_SPLIT_OVERLAP), and block column 0 is what the factorisation needs first.
__host__ __device__ __forceinline__ int64_t pack_index(int64_t ta, int64_t tb, int64_t nt) {
  return tb * nt - tb * (tb - 1) / 2 + (ta - tb);
}

// ---- XCD-aware workgroup -> tile maps (speed only: the dispatcher is observed to place workgroup b on XCD b % 8, and each XCD
// has its own 4 MiB L2; nothing depends on it for correctness).  With the launch order as tile order every XCD ends up reading
// ALL operand panels (C2's kappa GEMM: 72 MB through the fabric for a 16 MB operand set); giving each XCD a contiguous range of
// a locality-preserving tile order cuts that to what a compact block of tiles needs (used by k_gemm_nt; measured neutral to
// slightly faster there, and slower for the symmetric product, which keeps its launch order -- see syrk_tn_body).
constexpr int N_XCD = 8;
// logical id of workgroup `bid` of `nwg`: XCD x = bid % 8 gets the contiguous range of ids it would own in a blocked split
// (bijective for any nwg)
__device__ __forceinline__ int64_t xcd_contiguous(int64_t bid, int64_t nwg) {
  const int64_t q = nwg / N_XCD, r = nwg % N_XCD, x = bid % N_XCD, s = bid / N_XCD;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + s;
}
// id -> tile (bm, bn) of a gy x gx grid in bands of GM tile rows, column-major inside a band: a contiguous id range is a compact
// rectangle of about GM x (range / GM) tiles
// (GM: the power of two with GM^2 >= tiles per XCD / 2, so an XCD's range is about square: 4 for 256 tiles, 8 for 1024)
__device__ __forceinline__ void banded_tile(int64_t id, int64_t gx, int64_t gy, int64_t& bm, int64_t& bn) {
  int64_t GM = 2;
  while (2 * GM * GM * N_XCD < gx * gy) GM *= 2;
  const int64_t per = GM * gx, band = id / per, first = band * GM, loc = id - band * per;
  const int64_t h = (gy - first) < GM ? (gy - first) : GM;
  bm = first + loc % h;
  bn = loc / h;
}
// Independent factorisations of equal shape (the latent GPs of a multi-class / multi-output / heteroscedastic model) share
// the launches: blockIdx.y selects the problem, so their latency-bound chains overlap instead of queueing.
constexpr int CHOL_MAXB = 16;
template <typename T>
struct CholBatch {
  T* A[CHOL_MAXB];
  T* X[CHOL_MAXB];
  T* Dg[CHOL_MAXB];
  T* E[CHOL_MAXB];
  const T* R[CHOL_MAXB];  // task graph only, optional: row 0 of the problem's LAST extension block ([eta1' ; 0], see erow)
};

// ---------------------------------------------------------------------------------------------------
// launch S(k), k = 0..nt-1: grid = (nP + nU, n_problems), 512 threads
//   nP = nt - k + ne  (block rows k..nt-1 of A, then the ne extension blocks)
//   nU = k >= 1 ? T(nt-k-1) + ne*(nt-k-1) : 0
// ---------------------------------------------------------------------------------------------------
// one workgroup's share of launch S(k): `bid` in [0, nP + nU) selects the role and the tile.  A device function so that the
// per-column launches (k_chol_step) and the single-launch fallback with grid barriers (k_chol_safe) run the same code.
// Panel groups (large matrices): with k0 <= k < jmax the launches S(k0) .. S(jmax - 1) only touch block columns [k0, jmax) -- the
// pending update of column k - 1 exists only for k > k0, and the U role is restricted to columns (k, jmax) -- and everything to the
// right of the group receives the group's jmax - k0 columns at once from k_chol_trail (one read-modify-write of the trailing
// matrix per group instead of one per column).  k0 = 0, jmax = nt is the plain right-looking algorithm.
__host__ __device__ __forceinline__ int64_t chol_nU(int64_t k, int64_t k0, int64_t jmax, int64_t nt, int64_t ne) {
  if (k <= k0) return 0;
  int64_t n = 0;
  for (int64_t j = k + 1; j < jmax; ++j) n += (nt - j) + ne;  // tiles (i >= j, j) and the extension tiles of column j
  return n;
}
template <typename T>
__device__ __forceinline__ void chol_step_body(T* __restrict__ A, T* __restrict__ X, T* __restrict__ Dg, T* __restrict__ E,
                                               int64_t bid, int64_t ld, int64_t ldx, int64_t lde, int64_t ne, int do_x,
                                               int64_t k, int64_t nt, int32_t* __restrict__ info, int64_t nvalid, T* sm, T* sc,
                                               T* piv, int64_t k0 = 0, int64_t jmax = -1, T* __restrict__ li = nullptr) {
  if (jmax < 0) jmax = nt;
  T* bufA = sm;
  T* bufB = sm + TILE * LDP;
  T* bufC = sm + 2 * TILE * LDP;
  const int tid = threadIdx.x;
  const int64_t nP = nt - k + ne;
  const bool pending = k > k0;
  const int64_t d0 = k * TILE, p0 = (k - 1) * TILE;
  if (bid < nP) {
    // ---------------- P: panel of block column k ----------------
    const int64_t b = bid;
    const bool ext = b >= (nt - k);
    T* rowp = ext ? E + (b - (nt - k)) * TILE * lde : A + (k + b) * TILE * ld;  // first row of this block row
    const int64_t ldr = ext ? lde : ld;
    Acc8<T> accD, accT;
    // the tiles to be updated are fetched first (MFMA accumulator layout) so their HBM/L2 latency overlaps the loads
    // and products of the pending update instead of following them
    acc8_foreach<T>(accD, [&](int r, int c, T& val) { val = A[(d0 + r) * ld + d0 + c]; });
    if (b > 0) acc8_foreach<T>(accT, [&](int r, int c, T& val) { val = rowp[r * ldr + d0 + c]; });
    if (pending) {  // pending rank-64 update from column k-1: acc = tile - L L'
      load_tile_lds<T, CHOL_THREADS>(A + d0 * ld + p0, ld, bufA);
      if (b > 0) load_tile_lds<T, CHOL_THREADS>(rowp + p0, ldr, bufC);
      __syncthreads();
      mma8_sub<T>(bufA, bufA, accD);
      if (b > 0) mma8_sub<T>(bufC, bufA, accT);
      __syncthreads();
    }
    acc8_foreach<T>(accD, [&](int r, int c, T& val) { bufA[r * LDP + c] = val; });
    if (b > 0)  // own tile with the pending update applied, parked in LDS across the factorisation
      acc8_foreach<T>(accT, [&](int r, int c, T& val) { bufC[r * LDP + c] = val; });
    __syncthreads();
    factor_diag_tile_2lvl<T>(bufA, bufB, sc, piv, info, d0, nvalid);
    if (b == 0) {
      for (int e = tid; e < TILE * TILE; e += CHOL_THREADS) {
        int R = e >> 6, Cc = e & 63;
        Dg[k * TILE * TILE + e] = bufA[R * LDP + Cc];
        if (do_x) X[(d0 + R) * ldx + d0 + Cc] = bufB[R * LDP + Cc];
        if (li) li[e] = bufB[R * LDP + Cc];  // the tile's inverse, for the panel solve of a blocked factorisation
      }
      return;
    }
    Acc8<T> acc;  // L_ik = T_ik * Linv^T
    acc.zero();
    mma8<T>(bufC, bufB, acc);
    acc8_foreach<T>(acc, [&](int r, int c, T& val) { rowp[r * ldr + d0 + c] = val; });
    return;
  }
  bid -= nP;
  {
    // ---------------- U: trailing update from column k-1, tiles (i, j) with k < j < jmax ----------------
    // column by column: nt - j matrix tiles (i = j .. nt-1), then the ne extension tiles
    int64_t j = k + 1;
    while (j < jmax && bid >= (nt - j) + ne) {
      bid -= (nt - j) + ne;
      ++j;
    }
    if (j >= jmax) return;
    T* rowp;
    int64_t ldr;
    const int64_t j0 = j * TILE;
    if (bid < nt - j) {
      rowp = A + (j + bid) * TILE * ld;
      ldr = ld;
    } else {
      rowp = E + (bid - (nt - j)) * TILE * lde;
      ldr = lde;
    }
    Acc8<T> acc;
    acc8_foreach<T>(acc, [&](int r, int c, T& val) { val = rowp[r * ldr + j0 + c]; });
    load_tile_lds<T, CHOL_THREADS>(rowp + p0, ldr, bufA);
    load_tile_lds<T, CHOL_THREADS>(A + j0 * ld + p0, ld, bufB);
    __syncthreads();
    mma8_sub<T>(bufA, bufB, acc);
    acc8_foreach<T>(acc, [&](int r, int c, T& val) { rowp[r * ldr + j0 + c] = val; });
  }
}

template <typename T>
__global__ __launch_bounds__(CHOL_THREADS) void k_chol_step(CholBatch<T> bt, int64_t ld, int64_t ldx, int64_t lde,
                                                            int64_t ne, int do_x, int64_t k, int64_t nt,
                                                            int32_t* __restrict__ info, int64_t nvalid, int64_t k0 = 0,
                                                            int64_t jmax = -1, T* __restrict__ li = nullptr,
                                                            int64_t li_stride = 0) {
  __shared__ __attribute__((aligned(16))) T sm[3 * TILE * LDP];
  __shared__ __attribute__((aligned(16))) T sc[SC_ELEMS];
  __shared__ T piv[TILE];
  chol_step_body<T>(bt.A[blockIdx.y], bt.X[blockIdx.y], bt.Dg[blockIdx.y], bt.E[blockIdx.y], (int64_t)blockIdx.x, ld, ldx, lde,
                    ne, do_x, k, nt, info, nvalid, sm, sc, piv, k0, jmax,
                    li ? li + blockIdx.y * li_stride + (k - k0) * TILE * TILE : nullptr);
}

// ---------------------------------------------------------------------------------------------------
// Blocked factorisation of large matrices (nt beyond the task graph): block columns in groups of G,
//   D(g): the G x G diagonal block alone       -- G small launches of k_chol_step restricted to the block (nt := k1, ne := 0),
//                                                 which also leave the inverses of the G diagonal tiles in `li`
//   P(g): the rows below it (and the extension) -- k_chol_panel: one workgroup per block row solves its G tiles against the block
//   T(g): everything right of the group         -- k_chol_trail: each tile receives the group's G rank-64 updates in one pass
// so that the serial part touches G tiles per group and the O(m^3) part is the trail kernel, which runs at the MFMA ceiling.
// The host splits T(g) into the next group's columns and the rest, and runs the rest on a second stream next to D(g+1), P(g+1).
//
// k_chol_panel: X(R, k0..k1-1) = A(R, k0..k1-1) Lblk^-T by forward substitution over the block's columns,
//   X_c = (A_c - sum_{c' < c} X_c' L(c, c')') Linv_c',
// right-looking inside the workgroup: the G tiles of the row stay in MFMA accumulators, X_c goes through LDS into the updates of
// the tiles right of it.  grid = (nt - k1 + ne, n_problems).
template <typename T, int G>
__global__ __launch_bounds__(CHOL_THREADS) void k_chol_panel(CholBatch<T> bt, int64_t ld, int64_t lde, int64_t ne, int64_t k0,
                                                             int64_t nt, const T* __restrict__ li, int64_t li_stride, int g) {
  T* __restrict__ A = bt.A[blockIdx.y];
  T* __restrict__ E = bt.E[blockIdx.y];
  li += blockIdx.y * li_stride;
  __shared__ __attribute__((aligned(16))) T sm[2 * TILE * LDP];
  T* bufA = sm;
  T* bufB = sm + TILE * LDP;
  const int64_t k1 = k0 + g, b = blockIdx.x;
  const bool ext = b >= nt - k1;
  T* rowp = ext ? E + (b - (nt - k1)) * TILE * lde : A + (k1 + b) * TILE * ld;
  const int64_t ldr = ext ? lde : ld;
  Acc8<T> acc[G];
#pragma unroll
  for (int c = 0; c < G; ++c)
    if (c < g) acc8_foreach<T>(acc[c], [&](int r, int cc, T& val) { val = rowp[r * ldr + (k0 + c) * TILE + cc]; });
#pragma unroll
  for (int c = 0; c < G; ++c) {
    if (c >= g) break;
    if (c > 0) __syncthreads();
    acc8_foreach<T>(acc[c], [&](int r, int cc, T& val) { bufA[r * LDP + cc] = val; });
    load_tile_lds<T, CHOL_THREADS>(li + c * TILE * TILE, TILE, bufB);
    __syncthreads();
    Acc8<T> x;
    x.zero();
    mma8<T>(bufA, bufB, x);
    acc8_foreach<T>(x, [&](int r, int cc, T& val) { rowp[r * ldr + (k0 + c) * TILE + cc] = val; });
    if (c + 1 < g) {
      __syncthreads();
      acc8_foreach<T>(x, [&](int r, int cc, T& val) { bufA[r * LDP + cc] = val; });
#pragma unroll
      for (int c2 = c + 1; c2 < G; ++c2) {
        if (c2 >= g) break;
        if (c2 > c + 1) __syncthreads();
        load_tile_lds<T, CHOL_THREADS>(A + (k0 + c2) * TILE * ld + (k0 + c) * TILE, ld, bufB);
        __syncthreads();
        mma8_sub<T>(bufA, bufB, acc[c2]);
      }
    }
  }
}

// Trailing update of a panel group: every tile (i, j) with block column j in [j_lo, j_hi) (matrix tiles i >= j, then the ne
// extension rows, column by column) receives
//   tile -= sum_{c = k0}^{k1 - 1} L(i, c) L(j, c)'
// in ONE pass: the tile stays in the MFMA accumulators across the k1 - k0 rank-64 products, so the trailing matrix is read and
// written once per group instead of once per block column (at m = 4096 the per-column launches are bound by exactly that
// stream: ~0.4 GB per launch at the start of a C5 factorisation).  grid = (sum_j (nt - j + ne), n_problems), 512 threads.
__host__ __device__ __forceinline__ int64_t chol_trail_tiles(int64_t j_lo, int64_t j_hi, int64_t nt, int64_t ne) {
  const int64_t w = j_hi - j_lo;  // sum_{j = j_lo}^{j_hi - 1} (nt - j + ne)
  return w <= 0 ? 0 : w * (nt + ne) - (j_lo + j_hi - 1) * w / 2;
}
template <typename T>
__global__ __launch_bounds__(CHOL_THREADS) void k_chol_trail(CholBatch<T> bt, int64_t ld, int64_t lde, int64_t ne, int64_t k0,
                                                             int64_t k1, int64_t nt, int64_t j_lo, int64_t j_hi) {
  T* __restrict__ A = bt.A[blockIdx.y];
  T* __restrict__ E = bt.E[blockIdx.y];
  __shared__ __attribute__((aligned(16))) T sm[2 * TILE * LDP];  // 68 KB in f64: two workgroups per CU hide each other's loads
  int64_t bid = blockIdx.x, j = j_lo;
  while (j < j_hi && bid >= (nt - j) + ne) {
    bid -= (nt - j) + ne;
    ++j;
  }
  if (j >= j_hi) return;
  T* rowp;
  int64_t ldr;
  if (bid < nt - j) {
    rowp = A + (j + bid) * TILE * ld;
    ldr = ld;
  } else {
    rowp = E + (bid - (nt - j)) * TILE * lde;
    ldr = lde;
  }
  const int64_t j0 = j * TILE;
  const T* colp = A + j0 * ld;  // block row j of L: the second operand
  Acc8<T> acc;
  acc8_foreach<T>(acc, [&](int r, int c, T& val) { val = rowp[r * ldr + j0 + c]; });
  for (int64_t c = k0; c < k1; ++c) {
    if (c > k0) __syncthreads();
    load_tile_lds<T, CHOL_THREADS>(rowp + c * TILE, ldr, sm);
    load_tile_lds<T, CHOL_THREADS>(colp + c * TILE, ld, sm + TILE * LDP);
    __syncthreads();
    mma8_sub<T>(sm, sm + TILE * LDP, acc);
  }
  acc8_foreach<T>(acc, [&](int r, int c, T& val) { rowp[r * ldr + j0 + c] = val; });
}

// ---------------------------------------------------------------------------------------------------
// k_chol_safe: the fallback behind every task-graph launch of the CAVI step.  The task graph (k_chol_dag below) relies on
// workgroups being dispatched in index order; when that assumption breaks (a second PROCESS running its own task graph on the
// same device is the known way) a dependency never arrives, the bounded spin gives up and latches info = -1.  This kernel is
// enqueued right behind the task graph on the same stream: its workgroups look at the latch and return at once when all is
// well (one empty launch per step, ~2 us; in a single-latent step it is the prologue of the row-statistics launch instead,
// k_safe_rowstats); on -1 it restores the inputs from their sources (A = -2 eta2, E = [kappa ; eta1' ; 0])
// and runs the per-column algorithm (chol_step_body) for all columns in this ONE launch, separated by grid barriers.  The grid
// is at most one workgroup per CU, so every workgroup gets a slot without any assumption about order, and the barriers cannot
// deadlock.  The step itself never fails; `retries` counts the events so that the host can warn and pause the task graph.
// ---------------------------------------------------------------------------------------------------
template <typename T>
struct SafeSrc {
  const T* kappa[CHOL_MAXB];  // Bq x n rows of E (leading dimension lde)
  const T* eta1[CHOL_MAXB];   // first row of the last extension block (the others are zero)
  const T* eta2[CHOL_MAXB];   // A = -2 eta2
  int64_t Bq;
  // round 3: launches that also deliver X = L^-1 (the factor of the updated -2 eta2 for Sigma / mu, K_ZZ's at a kernel refresh)
  // have an in-stream fallback too, so that no host check (= stream synchronisation) sits behind them:
  int want_x = 0;             // after the factorisation, X = L^-1 by forward substitution, one thread per column (slow, rare)
  // kz: A is not -2 eta2 but the kernel matrix of the inducing points, recomputed from its inputs (the factor overwrote it):
  //   A[i][j] = variance * base(|| s .* (z_i - z_j) ||^2) + jitter [i == j], identity in the padding (i, j >= mz)
  const T* kz = nullptr;      // Z (mz x D, leading dimension ldz); nullptr: A = -2 eta2
  const T* kscales = nullptr; // [scale_0 .. scale_{D-1} | variance]
  int64_t mz = 0, ldz = 0, Dz = 0;
  int kkind = 0;
  T kvariance = T(1), kjitter = T(0);  // kvariance < 0: read kscales[Dz]
  // split launch (k_chol_dag ROLE 1 / 2): the chain kernel(s) of the launch have counted themselves out when *chain_done >= chain_want
  const int32_t* chain_done = nullptr;
  int32_t chain_want = 0;
  // the launch also delivers P = X' X (ProdArgs): the fallback forms it from its own X (problem 0; leading dimension ldpo)
  T* pout = nullptr;
  int64_t ldpo = 0;
  double* ld_out = nullptr;      // ... and log det of the factor over the first ld_n rows (ProdArgs::ld_out), with its status words
  int32_t* ld_status = nullptr;
  int64_t ld_n = 0;
};

// the kernel functions of agp_cavi.h restated for the fallback (agp_chol.h is included first): base(d2), d2 = squared scaled distance
template <typename T>
__device__ __forceinline__ T safe_kernel_base(int kind, T d2) {
  d2 = d2 > T(0) ? d2 : T(0);
  if (kind == 0) return exp(T(-0.5) * d2);  // SqExponential
  const T r = sqrt(d2);
  if (kind == 1) {  // Matern 5/2
    const T s5 = T(2.23606797749978969641);
    return (T(1) + s5 * r + T(5) / T(3) * d2) * exp(-s5 * r);
  }
  if (kind == 2) {  // Matern 3/2
    const T s3 = T(1.73205080756887729353);
    return (T(1) + s3 * r) * exp(-s3 * r);
  }
  return exp(-r);  // Exponential
}

// Bounded (round 6): a workgroup of the fallback that cannot become resident (its grid is <= n_cu - CHOL_MAXB workgroups of ~110 KB
// LDS, dealt to the XCDs round-robin; ONE further long-lived large-LDS workgroup of somebody else in an XCD is enough) used to make
// this barrier spin forever.  Now the wait is limited on the device's constant 100 MHz clock; when it runs out the workgroup latches
// info = -3 (agp_svgp_check_status: AGP_ERR_HIP, "the in-stream fallback could not become resident"; the host clears the barrier
// words) and every workgroup leaves at its next barrier.  Returns false (workgroup-uniform) when the barrier is broken.
constexpr long long GRID_BARRIER_TICKS = 8LL * 100000000LL;  // 8 s (a phase with the inverse's forward substitution is ~0.2 s)
__device__ __forceinline__ bool grid_barrier(unsigned* ctr, unsigned target, int32_t* info) {
  __shared__ int gb_ok;
  __syncthreads();
  if (threadIdx.x == 0) {
    int ok = 1;
    __threadfence();  // agent-scope release: this workgroup's plain stores reach memory before it is counted
    atomicAdd(ctr, 1u);
    const long long t0 = (long long)wall_clock64();
    unsigned spins = 0;
    while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(8);
      if ((++spins & 255u) == 0) {
        if (__hip_atomic_load(info, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == -3) {
          ok = 0;
          break;
        }
        if ((long long)wall_clock64() - t0 > GRID_BARRIER_TICKS) {
          atomicExch(info, -3);
          ok = 0;
          break;
        }
      }
    }
    __threadfence();  // agent-scope acquire: drop what this XCD's L2 may still hold of the others' tiles
    gb_ok = ok;
  }
  __syncthreads();
  return gb_ok != 0;
}

// (a device function so that the kernel which follows the task graph in a single-latent step -- row statistics and local
//  update -- can carry the fallback itself instead of waiting behind an extra launch: k_safe_rowstats in agp_capi.hip.
//  Returns whether the fallback ran; every thread of every workgroup of a <= n_cu grid must call.)
template <typename T>
__device__ __forceinline__ bool chol_safe_body(const CholBatch<T>& bt, const SafeSrc<T>& src, int nb, int64_t ld, int64_t ldx,
                                               int64_t lde, int64_t ne, int64_t nt, int32_t* __restrict__ info, int64_t nvalid,
                                               unsigned* __restrict__ bar, int32_t* __restrict__ retries, T* sm, T* sc, T* piv) {
  if (__hip_atomic_load(info, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != -1) return false;
  if (src.chain_done) {  // a split launch: its chain kernel may still be writing diagonal factors for a few microseconds (DagSync::done)
    if (threadIdx.x == 0) {
      long spins = 0;
      while ((int32_t)(__hip_atomic_load(src.chain_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - src.chain_want) < 0 &&
             ++spins < (1L << 24))
        __builtin_amdgcn_s_sleep(16);
    }
    __syncthreads();
  }
  const unsigned nwg = gridDim.x;
  const int64_t n = nt * TILE, gsz = (int64_t)nwg * CHOL_THREADS, g0 = (int64_t)blockIdx.x * CHOL_THREADS + threadIdx.x;
  unsigned phase = 0;
  for (int q = 0; q < nb; ++q) {  // inputs back from their sources (the aborted launch left E half overwritten)
    T* A = bt.A[q];
    T* E = bt.E[q];
    if (src.kz) {  // K_ZZ + jitter I from the inducing points (q == 0 only: one latent per launch)
      const T var = src.kvariance < T(0) ? src.kscales[src.Dz] : src.kvariance;
      for (int64_t e = g0; e < n * n; e += gsz) {
        const int64_t i = e / n, j = e % n;
        T v = i == j ? T(1) : T(0);
        if (i < src.mz && j < src.mz) {
          T d2 = T(0);
          for (int64_t d = 0; d < src.Dz; ++d) {
            const T t = (src.kscales ? src.kscales[d] : T(1)) * (src.kz[i * src.ldz + d] - src.kz[j * src.ldz + d]);
            d2 += t * t;
          }
          v = var * safe_kernel_base<T>(src.kkind, d2) + (i == j ? src.kjitter : T(0));
        }
        A[i * ld + j] = v;
      }
    } else {
      for (int64_t e = g0; e < n * n; e += gsz) A[(e / n) * ld + (e % n)] = T(-2) * src.eta2[q][(e / n) * ld + (e % n)];
    }
    if (ne > 0) {
      for (int64_t e = g0; e < src.Bq * n; e += gsz) E[(e / n) * lde + (e % n)] = src.kappa[q][(e / n) * lde + (e % n)];
      for (int64_t e = g0; e < TILE * n; e += gsz)
        E[(src.Bq + e / n) * lde + (e % n)] = (e / n) == 0 ? src.eta1[q][e % n] : T(0);
    }
  }
  if (!grid_barrier(bar, ++phase * nwg, info)) return true;
  if (blockIdx.x == 0 && threadIdx.x == 0) {  // every workgroup has seen the -1 by now: the latch goes back to "no failure",
    atomicCAS(info, -1, 0);                   // so that a non-positive pivot of THIS run is reported like any other (CAS: a
    atomicAdd(retries, 1);                    // broken barrier's -3 stays)
  }
  if (!grid_barrier(bar, ++phase * nwg, info)) return true;
  for (int64_t k = 0; k < nt; ++k) {
    const int64_t nP = nt - k + ne;
    const int64_t nU = chol_nU(k, 0, nt, nt, ne);
    for (int64_t v = blockIdx.x; v < (nP + nU) * nb; v += nwg) {
      const int q = (int)(v % nb);
      chol_step_body<T>(bt.A[q], bt.X[q], bt.Dg[q], bt.E[q], v / nb, ld, ldx, lde, ne, 0, k, nt, info, nvalid, sm, sc, piv);
      __syncthreads();  // the next share reuses the LDS tiles
    }
    if (!grid_barrier(bar, ++phase * nwg, info)) return true;
  }
  if (src.want_x) {
    // X = L^-1 (lower triangular) column by column: thread j solves L x = e_j by forward substitution.  L's strictly-lower tiles sit
    // in A, its diagonal tiles in Dg.  O(n^2) dependent steps per thread: milliseconds -- this is the path of a lost dependency,
    // which the bounded spin in front of it has already made ~0.1 s long.
    for (int q = 0; q < nb; ++q) {
      const T* A = bt.A[q];
      const T* Dg = bt.Dg[q];
      T* X = bt.X[q];
      auto Lel = [&](int64_t i, int64_t k) -> T {
        return (i / TILE) == (k / TILE) ? Dg[(i / TILE) * TILE * TILE + (i % TILE) * TILE + (k % TILE)] : A[i * ld + k];
      };
      for (int64_t j = g0; j < n; j += gsz) {
        const int64_t t0 = (j / TILE) * TILE;
        for (int64_t i = t0; i < j; ++i) X[i * ldx + j] = T(0);  // above the diagonal inside the diagonal tile
        X[j * ldx + j] = T(1) / Lel(j, j);
        for (int64_t i = j + 1; i < n; ++i) {
          T sacc = T(0);
          for (int64_t k = j; k < i; ++k) sacc += Lel(i, k) * X[k * ldx + j];
          X[i * ldx + j] = -sacc / Lel(i, i);
        }
      }
    }
    if (!grid_barrier(bar, ++phase * nwg, info)) return true;
    if (src.pout) {  // P = X' X, one thread per lower element, k ascending (X is lower triangular: the sum starts at row i)
      const T* X = bt.X[0];
      for (int64_t e = g0; e < n * n; e += gsz) {
        const int64_t i = e / n, j = e % n;
        if (j > i) continue;
        T sacc = T(0);
        for (int64_t k = i; k < n; ++k) sacc += X[k * ldx + i] * X[k * ldx + j];
        src.pout[i * src.ldpo + j] = sacc;
        src.pout[j * src.ldpo + i] = sacc;
      }
      if (src.ld_out && blockIdx.x == 0) {  // log det from the diagonal factors (as k_logdiag_sum reads them)
        const T* Dg = bt.Dg[0];
        double lsum = 0.0;
        for (int64_t i = threadIdx.x; i < src.ld_n; i += CHOL_THREADS) lsum += log((double)Dg[(i / TILE) * TILE * TILE + (i % TILE) * (TILE + 1)]);
        lsum = block_sum<double>(lsum, reinterpret_cast<double*>(sc));
        if (threadIdx.x == 0) {
          src.ld_out[0] = lsum;
          int32_t* st_ = src.ld_status;
          if (st_ && st_[1] != 0 && st_[3] == 0) st_[3] = (st_[0] != 0 || st_[2] != 0) ? 2 : 1;
        }
      }
      if (!grid_barrier(bar, ++phase * nwg, info)) return true;
    }
  }
  // leave the barrier words at zero for the next use: arrivals are counted on a second word, the last one to arrive resets both
  if (threadIdx.x == 0) {
    if (atomicAdd(bar + 1, 1u) == nwg - 1) {
      __hip_atomic_store(bar, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(bar + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  return true;
}

template <typename T>
__global__ __launch_bounds__(CHOL_THREADS) void k_chol_safe(CholBatch<T> bt, SafeSrc<T> src, int nb, int64_t ld, int64_t ldx,
                                                            int64_t lde, int64_t ne, int64_t nt,
                                                            int32_t* __restrict__ info, int64_t nvalid,
                                                            unsigned* __restrict__ bar, int32_t* __restrict__ retries) {
  __shared__ __attribute__((aligned(16))) T sm[3 * TILE * LDP];
  __shared__ __attribute__((aligned(16))) T sc[SC_ELEMS];
  __shared__ T piv[TILE];
  (void)chol_safe_body<T>(bt, src, nb, ld, ldx, lde, ne, nt, info, nvalid, bar, retries, sm, sc, piv);
}

// ---------------------------------------------------------------------------------------------------
// k_chol_dag: the augmented factorisation as ONE launch of a tile task graph -- one 512-thread workgroup per 64x64 tile
// (r, c), c <= r (matrix rows, then the extension row blocks), numbered column-major so that every dependency points to a
// lower workgroup index (in-order dispatch => no deadlock even when the grid is not fully resident).  The tile lives in the
// MFMA accumulators for its whole life -- no read-modify-write of the trailing matrix in global memory:
//     for j < c :  wait L(r,j), L(c,j)          acc -= L(r,j) L(c,j)'
//     r == c    :  factor the tile -> L_cc (Dg), X_c = L_cc^-1 (X) ; publish X_c
//     r >  c    :  wait X_c ; L(r,c) = acc X_c' ; publish L(r,c)
// Hand-over between workgroups WITHOUT cache-wide fences, through a sentinel-filled hand-over area whose elements validate
// themselves (see "self-validating hand-over" below): the producer writes the tile with agent-scope relaxed atomic stores (sc1:
// write-through to the memory side), waits for their acknowledgement (s_waitcnt), barrier, then stores the flag; consumers poll
// the flag with agent-scope loads and read the tile with agent-scope (sc1) loads, re-loading what is not there yet.  Measured
// (tools/ubench/hop.hip): 1.6 us per hop (0.4 publish + 0.4 flag + 0.8 fetch of 32 KB), independent of how much dirty data
// the other workgroups keep in the L2s; release / acquire fences (an earlier row-per-workgroup version)
// cost 3.8 - 10 us for the same hop.
// FUSED (the version in use): workgroup 0 is the CHAIN -- it carries the critical path through all columns
//     factor(k) -> L(k+1,k) = T X_k' -> S = D - L L' -> factor(k+1)
// so that no hop is left on it; T = tile (k+1,k) and D = tile (k+1,k+1) are parked with all their other updates applied by
// two feeder workgroups (the tile workgroups of those two positions) and prefetched into LDS by the chain's idle waves while
// the elimination of column k is still running (ChainPrefetch, a hook called once per elimination round).
// Flags hold the launch's epoch (never reset); a bounded spin turns a lost dependency into info = -1 instead of a hang.
// flags: int32 [(nt + ne + nx) * nt] tile-ready | [nt] x-ready | [nt] T parked | [nt] D parked | [1] abort, each on its own
// 256-byte line (stride DAG_FS)
// ---------------------------------------------------------------------------------------------------
constexpr long DAG_SPIN_LIMIT = 1L << 17;  // ~50-100 ms of polling: far beyond any real wait (a whole factorisation takes < 15 ms
                                           // at the largest size the task graph is used for), short enough that a lost dependency is
                                           // handed to k_chol_safe before anybody notices
// workgroup index of diagonal tile `col`: column c holds nt - c matrix tiles, ne extension tiles and, with the inverse
// requested (nx), c + 1 identity-row tiles
__device__ __forceinline__ int64_t chain_slot(int64_t col, int64_t nt, int64_t ne, int64_t nx) {
  return nx ? col * (nt + ne + 1) : col * (nt + ne) - col * (col - 1) / 2;
}
constexpr int DAG_FS = 64;  // flag stride in int32: one 256-byte line per flag, so the pollers spread over the memory channels

// ---- self-validating hand-over -------------------------------------------------------------------------------------
// Tiles travel between workgroups through a hand-over area H (one contiguous 64x64 slot per tile) that the host fills with a
// SENTINEL bit pattern (a signalling-NaN payload no arithmetic produces) before the launch (and again after it: see Dirty in
// agp_capi.hip).  The producer stores the tile there
// with coherent (sc1) 8-byte stores and then raises a flag; the flag is only a HINT that the data is on its way: the consumer
// loads the slot with coherent loads and re-loads any element that still reads as the sentinel.  8-byte stores are single-copy
// atomic, so an element is either the sentinel or final -- no ordering between the data stores and the flag store is needed,
// hence no cache-wide release fence (measured: the fence costs 18 % of the step; without fence AND without validation a consumer
// read a stale tile about once in 10^4 launches under load).
template <typename T>
struct Sent;
template <>
struct Sent<double> {
  typedef unsigned long long U;
  static constexpr U bits = 0x7FF4DEADBEEF1234ull;
};
template <>
struct Sent<float> {
  typedef unsigned int U;
  static constexpr U bits = 0x7FA5F00Du;
};
template <typename T>
__device__ __forceinline__ bool is_sent(T v) {
  return __builtin_bit_cast(typename Sent<T>::U, v) == Sent<T>::bits;
}
// re-load while the element still holds the sentinel (bounded: a producer that never arrives ends in the abort path)
template <typename T>
__device__ __forceinline__ T hv_settle(const T* p, T v) {
  long spins = 0;
  while (is_sent(v) && ++spins < (1L << 22)) {
    __builtin_amdgcn_s_sleep(1);
    v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  return v;
}

// hand-over slot (4096 contiguous elements) -> LDS tile; all loads are issued before the first use (the compiler keeps an
// atomic load ordered against a following store: the interleaved form ran the eight loads back to back, 2.5 us instead of 0.8)
template <typename T>
__device__ __forceinline__ void load_tile_lds_hv(const T* Hs, T* S) {
  constexpr int Q = TILE * TILE / CHOL_THREADS;
  T v[Q];
#pragma unroll
  for (int q = 0; q < Q; ++q) v[q] = __hip_atomic_load(Hs + threadIdx.x + q * CHOL_THREADS, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    const int e = threadIdx.x + q * CHOL_THREADS;
    S[(e >> 6) * LDP + (e & 63)] = hv_settle<T>(Hs + e, v[q]);
  }
}

// two slots at once (both operands of a pending update): 16 loads in flight per thread
template <typename T>
__device__ __forceinline__ void load_tiles_lds_hv2(const T* H0, T* S0, const T* H1, T* S1) {
  constexpr int Q = TILE * TILE / CHOL_THREADS;
  T v[2 * Q];
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    v[q] = __hip_atomic_load(H0 + threadIdx.x + q * CHOL_THREADS, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    v[Q + q] = __hip_atomic_load(H1 + threadIdx.x + q * CHOL_THREADS, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    const int e = threadIdx.x + q * CHOL_THREADS;
    S0[(e >> 6) * LDP + (e & 63)] = hv_settle<T>(H0 + e, v[q]);
    S1[(e >> 6) * LDP + (e & 63)] = hv_settle<T>(H1 + e, v[Q + q]);
  }
}

#ifdef AGP_DEBUG_PTRS
// development builds only (docs/DESIGN_LOG.md section 14, "which wait expires"): what the FIRST bounded wait of a launch that ran out was
// waiting for -- [0] number of expired waits since the library was loaded, [1] address of the flag, [2] epoch, [3] workgroup index,
// [4] 1 = spin limit / 2 = saw the launch's abort flag, [5] chain kernels whose release word never came.  Read by agp_debug_dag_diag.
__device__ unsigned long long agp_dag_diag[8];
#endif
// one thread polls up to two flags for `epoch`; returns false (workgroup-uniform) when the run was aborted
__device__ __forceinline__ bool dag_wait(const int32_t* f0, const int32_t* f1, int32_t epoch, int32_t* abort_flag,
                                         int32_t* info, int* lds_ok) {
  if (threadIdx.x == 0) {
    long spins = 0;
    int ok = 1;
    for (int w = 0; w < 2 && ok; ++w) {
      const int32_t* f = w == 0 ? f0 : f1;
      if (!f) continue;
      while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > DAG_SPIN_LIMIT ||
            ((spins & 63) == 0 && __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == epoch)) {
#ifdef AGP_DEBUG_PTRS
          if (spins > DAG_SPIN_LIMIT && atomicAdd(&agp_dag_diag[0], 1ull) == 0) {  // (the first one is the cause, the others follow it)
            agp_dag_diag[1] = (unsigned long long)f;
            agp_dag_diag[2] = (unsigned long long)epoch;
            agp_dag_diag[3] = (unsigned long long)blockIdx.x;
            agp_dag_diag[4] = 1;
          }
#endif
          __hip_atomic_store(abort_flag, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          atomicExch(info, -1);
          ok = 0;
          break;
        }
      }
    }
    *lds_ok = ok;
  }
  __syncthreads();
  return *lds_ok != 0;
}

// Hand-over from the CAVI step's task-graph launch to the look-ahead stream through a word in signal memory instead of an event
// record on the step's own stream (which costs the in-order queue ~10 us per step): the chain workgroup stores `seq` as soon
// as the launch runs -- everything enqueued before it is complete by then, so the look-ahead stream (k_wait_ge) may overwrite
// the kappa buffers of the step before.  (The other direction stays a stream-level wait: letting the extension-row workgroups
// poll for the look-ahead's kappa themselves saves 3 more us but can fill every CU with pollers while the look-ahead's GEMM
// still needs one -- seen once in ~20 000 steps as a spin-limit abort)
struct DagSync {
  int32_t* started = nullptr;
  int32_t seq = 0;
  // split launch (ROLE 1 / 2 of k_chol_dag): the chain kernel sits on its own stream and may be dispatched before the kernels that
  // precede the tile kernel on the step's stream are done; it waits for `go` == go_val, which the tile kernel stores when it starts
  int32_t* go = nullptr;
  int32_t go_val = 0;
  // ... and counts itself out on `done` (one per chain workgroup, at every exit): the in-stream fallback of an ABORTED launch waits
  // for that count before it rebuilds the inputs -- the only place where the tile kernel can end before the chain kernel has stopped
  // writing (a chain that is in the middle of a tile factorisation notices the abort at its next wait, up to ~17 us later).  No
  // event joins the two streams: the tile kernel cannot end before the chain's last publish, after which the chain writes nothing.
  int32_t* done = nullptr;
  // round 6: ... and counts itself IN on `here` (one per chain workgroup, the first thing it does).  The step's stream does not start
  // the tile kernel before every chain workgroup of the launch is resident (k_wait_here in front of the tile kernel): a tile kernel
  // that fills every CU with workgroups waiting for a chain kernel that has not found a place yet was the one way a split launch
  // lost a dependency on its own (docs/DESIGN_LOG.md section 14, defect 5: about once in 10 000 launches at fp32 m = 1024 / B = 2048)
  int32_t* here = nullptr;
};
// the step stream's side of DagSync::here: one wave in front of the tile kernel.  In the steady state the chain kernel has been
// resident for the whole of the kernels that precede the tile kernel (it is enqueued first, on a stream of higher priority, and
// becomes dispatchable when the chain kernel of the launch before exits -- when that launch's tile kernel ends), so this costs a
// launch slot and no wait.  Bounded (2 s): past that the tile kernel starts anyway and is protected by its own bounded waits.
__global__ void k_wait_here(const int32_t* __restrict__ here, int32_t want) {
  if (threadIdx.x != 0) return;
  const long long t0 = (long long)wall_clock64();
  unsigned spins = 0;
  while ((int32_t)(__hip_atomic_load(here, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - want) < 0) {
    __builtin_amdgcn_s_sleep(2);
    if ((++spins & 1023u) == 0 && (long long)wall_clock64() - t0 > 200000000LL) break;
  }
}
// (ROLE 1) every thread's stores acknowledged, then one count
__device__ __forceinline__ void chain_count_out(const DagSync& sync) {
  if (!sync.done) return;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_fetch_add(sync.done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Self-test for the hand-over below, run once per handle on the two streams at the same time: each side raises its own word
// and waits (bounded, ~2 ms) for the other's.  Both succeed only if kernels of the two streams really are in flight together;
// where dispatches are serialised (counter collection by rocprofv3 --pmc, AMD_SERIALIZE_KERNEL, a single hardware queue) the
// one that runs first times out -- and a polling kernel on one stream could then keep the launch it waits for on the other
// from ever starting, so such a handle keeps the events.
__global__ void k_handshake(int32_t* mine, const int32_t* theirs, int32_t* ok) {
  if (threadIdx.x != 0) return;
  __hip_atomic_store(mine, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  long spins = 0;
  int seen = 0;
  while (!(seen = __hip_atomic_load(theirs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) && ++spins < 4000) __builtin_amdgcn_s_sleep(16);
  *ok = seen;
}

// the look-ahead stream's side: one wave polls `started` at a leisurely rate (hipStreamWaitValue32's polling kernel slowed the
// task graph next to it when the wait was long: C3, 450 us).  The launch it waits for is always enqueued already.
__global__ void k_wait_ge(const int32_t* __restrict__ p, int32_t want, int32_t* __restrict__ info) {
  if (threadIdx.x != 0) return;
  long spins = 0;
  while ((int32_t)(__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - want) < 0) {
    __builtin_amdgcn_s_sleep(127);
    if (++spins > (1L << 24)) {  // about a minute: the step this waits for never ran
      atomicExch(info, -2);
      break;
    }
  }
}

// the step stream's side of the other direction (round 3): the look-ahead's "done" word instead of a cross-stream event wait.
// Short naps: in the steady state the word is set long before this kernel runs, and when it is not, the step is waiting.
__global__ void k_wait_ge_fast(const int32_t* __restrict__ p, int32_t want, int32_t* __restrict__ info) {
  if (threadIdx.x != 0) return;
  long spins = 0;
  while ((int32_t)(__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - want) < 0) {
    __builtin_amdgcn_s_sleep(8);
    if (++spins > (1L << 27)) {
      atomicExch(info, -2);
      break;
    }
  }
}

// every thread's hand-over stores acknowledged by the L2 -> barrier -> flag
__device__ __forceinline__ void dag_signal(int32_t* flag, int32_t epoch) {
  // the flag is a hint (see "self-validating hand-over"): waiting for the L2's acknowledgement of the stores just makes it a
  // good one -- the consumer validates every element, so no agent-scope release is needed
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(flag, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// one 16x16 result tile over k in [0, kend), kend a multiple of 8: acc += (neg ? -1 : 1) * Ar(16 rows, [r][k]) * Bc(16
// columns given as [c][k])' ; two partial accumulators keep dependent MFMAs apart
template <typename T>
__device__ __forceinline__ typename Mfma<T>::acc_t mma_tile16(const T* Ar, const T* Bc, int kend,
                                                              typename Mfma<T>::acc_t acc, bool neg, int lane) {
  const T* pa = Ar + (lane & 15) * LDP + (lane >> 4);
  const T* pb = Bc + (lane & 15) * LDP + (lane >> 4);
  typename Mfma<T>::acc_t p1;
#pragma unroll
  for (int r = 0; r < 4; ++r) p1[r] = T(0);
#pragma unroll 2
  for (int kk = 0; kk < kend / 4; kk += 2) {
    T a0 = pa[kk * 4], a1 = pa[kk * 4 + 4];
    const T b0 = pb[kk * 4], b1 = pb[kk * 4 + 4];
    if (neg) {
      a0 = -a0;
      a1 = -a1;
    }
    acc = Mfma<T>::mma(a0, b0, acc);
    p1 = Mfma<T>::mma(a1, b1, p1);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) acc[r] += p1[r];
  return acc;
}

// side job of the chain workgroup, stepped by the idle waves 4-7 during the second half of a tile factorisation: look once
// whether the two feeder tiles of the next column are parked, then move them into LDS one tile per round
template <typename T>
struct ChainPrefetch {
  const T* src;  // hand-over slot of the parked tile (k+1, k) ; the parked diagonal tile (k+1, k+1) is the next slot
  T *dT, *dD;    // LDS destinations
  int* bad;      // LDS: set when a prefetched element still read as the sentinel (the chain then fetches again, blocking)
  const int32_t *f1, *f2;
  int32_t epoch;
  int* ok;       // LDS
  const T* dgsrc;  // LDS: factor of the previous column, still to be stored to Dg (nullptr: none)
  T* gDg;
  const T* lsrc;   // LDS: L(k, k-1), still to be stored to its real home in A (nullptr: none)
  T* gL;           // A + (k * TILE) * ld + (k - 1) * TILE
  int ld_i;
  bool coh = false;  // the chain runs as a kernel of its own: its plain stores would only become visible when THAT kernel ends
  T v[TILE * TILE / 256];
  __device__ __forceinline__ void operator()(int round) {
    int t = (int)threadIdx.x - 256;
    if (t < 0) return;
    // (opaque to the optimiser: otherwise the LDS offsets (e >> 6) * LDP + (e & 63) of all sixteen e = t + 256 q are hoisted out of
    //  the chain's loop and live through every tile factorisation -- the launch with prologue and identity rows, at the 256-register
    //  cap of a 512-thread workgroup, spilled two of them to scratch)
    // fp64 only: the fp32 launches are far from their cap, and their register allocation is left as it was.
    if (sizeof(T) == 8) asm volatile("" : "+v"(t));
    if (round < 2) {  // L_{k-1,k-1} -> Dg, off the critical path (plain stores, read by later kernels only)
      if (dgsrc) {
#pragma unroll
        for (int q = 0; q < TILE * TILE / 512; ++q) {
          const int e = t + (2 * q + round) * 256;
          if (coh) __hip_atomic_store(gDg + e, dgsrc[(e >> 6) * LDP + (e & 63)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          else gDg[e] = dgsrc[(e >> 6) * LDP + (e & 63)];
        }
      }
      return;
    }
    if (round < 4 && lsrc) {  // L(k, k-1) -> A, half per round
#pragma unroll
      for (int q = 0; q < TILE * TILE / 512; ++q) {
        const int e = t + (2 * q + (round - 2)) * 256;
        gL[(e >> 6) * ld_i + (e & 63)] = lsrc[(e >> 6) * LDP + (e & 63)];
      }
    }
    if (!src) return;
    if (round == 3) {
      if (t == 0)
        *ok = (__hip_atomic_load(f1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == epoch) &&
              (__hip_atomic_load(f2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == epoch);
      return;
    }
    if (round < 4 || round > 6 || !*ok) return;
    if (round >= 5) {
      T* d = round == 5 ? dT : dD;
#pragma unroll
      for (int q = 0; q < TILE * TILE / 256; ++q) {
        const int e = t + q * 256;
        if (is_sent<T>(v[q])) *bad = 1;
        d[(e >> 6) * LDP + (e & 63)] = v[q];
      }
    }
    if (round <= 5) {
      const T* g = round == 4 ? src : src + TILE * TILE;
#pragma unroll
      for (int q = 0; q < TILE * TILE / 256; ++q)
        v[q] = __hip_atomic_load(g + t + q * 256, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
};

// ---- PRO: the natural-gradient step of the PREVIOUS minibatch as the prologue of the task graph (round 3) -------------------------
// Until round 2 the fused product  S = kappa' diag(w) kappa  + eta2 step ran as a kernel of its own between two task graphs
// (k_syrk_tn<SY_ETA2>, 53 us at C2, all 136 tiles finishing together) although the chain only needs block column 0 of the new
// -2 eta2 at first and one more column every ~18 us.  With PRO the launch of step t+1 takes the pending step of minibatch t itself:
//   * every matrix-tile workgroup (R, c) forms its own tile of S -- split along k over ks[c] workgroups for the first block
//     columns (HELPER workgroups, dispatched right before the column's tile workgroups; they wait for nothing, so the dispatch-
//     order argument of the task graph is untouched), whole for the later ones, which have tens of microseconds of slack;
//   * it then applies  eta2 += lr (-(S + K^-1/2) - eta2)  (analyticVI.jl:172-180, 229-246), writes eta2 home (both triangles) and
//     keeps A = -2 eta2 in its accumulators: the matrix never travels through memory between the step and the factorisation;
//   * the tile of the [eta1' ; 0] extension row in block column c takes  eta1 += lr (kappa' r + K^-1 mu0 - eta1)  for its 64
//     entries (analyticVI.jl:160-169) the same way.
// The prologue of a workgroup finishes before its first abortable wait, and helpers never wait: eta1 / eta2 are completely
// updated even when the factorisation behind them is aborted, so the in-stream fallback (k_chol_safe) finds its sources intact.
// Partial tiles travel through sentinel-validated hand-over slots like every other tile; summation order is fixed (own slice,
// then helpers 1, 2, ...), so results are bitwise reproducible.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
// 16-byte device-coherent load (the compiler has no such builtin: its agent-scope atomic loads stop at 8 bytes).  Every aligned
// 4- / 8-byte part of the result is a single-copy-atomic read, which is all the sentinel validation needs.  The caller waits with
// wait_vmcnt0() before touching the result (the compiler does not count these loads).
__device__ __forceinline__ u32x4 load16_sc1(const void* p) {
  u32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
  return v;
}
__device__ __forceinline__ void wait_vmcnt0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
template <typename T>
__device__ __forceinline__ T vec16_elem(u32x4 v, int i);
template <>
__device__ __forceinline__ double vec16_elem<double>(u32x4 v, int i) {
  const unsigned long long u = ((unsigned long long)v[2 * i + 1] << 32) | v[2 * i];
  return __builtin_bit_cast(double, u);
}
template <>
__device__ __forceinline__ float vec16_elem<float>(u32x4 v, int i) {
  return __builtin_bit_cast(float, (unsigned int)v[i]);
}
// where thread `tid` keeps its e-th accumulator element (Acc8 order) inside a helper's slot: consecutive threads own consecutive
// 16-byte groups, so both the helper's stores and the tile workgroup's 16-byte loads are contiguous across a wave
template <typename T>
__device__ __forceinline__ int pro_slot_off(int tid, int e) {
  constexpr int VEC = 16 / (int)sizeof(T);
  return (e / VEC) * (CHOL_THREADS * VEC) + tid * VEC + (e % VEC);
}

// ---- EPI: the row statistics + local update of the step as the EPILOGUE of its task-graph launch (round 3) -------------------------
// mean_f = W v and var_f = rowsum(W .* W) + K~ (latentgp.jl:179,189) need a whole row of W = kappa L^-T, i.e. the tiles (R, 0 ..
// nt-1) of an extension block row R and the matching pieces of v = L^-1 eta1.  The workgroup of the LAST tile of the row, (R, nt-1),
// has fetched every W(R, j), j < nt-1, anyway (for its own rank-64 updates): it accumulates the two row sums on the way, adds its own
// tile when that is solved, and finishes its 64 rows -- K~, mean_f, var_f, the likelihood's local update and the expectation
// gradients r = rho grad_E_mu, w = rho grad_E_Sigma (rowstats_finish, agp_cavi.h) -- so that no row-statistics kernel follows the
// launch.  r and w are written to the buffers the NEXT launch's prologue reads (this launch's prologue reads the previous pair).
template <typename T>
struct EpiArgs {
  int on = 0;
  int64_t B = 0;
  int nslices = 0;
  const T* pk = nullptr;  // K~ partial slices [nslices][ldp]
  int64_t ldp = 0;
  T kdiag = T(0);
  const T* kd_ptr = nullptr;  // kdiag < 0: device-resident kernel variance
  int use_kt = 0;
  T jitter = T(0), rho = T(0);
  LikParams<T> lp{};
  const T* y = nullptr;
  const int64_t* idx = nullptr;
  T *Kt = nullptr, *muf = nullptr, *varf = nullptr, *cb = nullptr, *theta = nullptr, *r = nullptr, *w = nullptr;
  int* flags = nullptr;
  const T* lam = nullptr;
  T* gamma = nullptr;
};

template <typename T>
struct ProArgs {
  const T* kap = nullptr;  // kappa of the minibatch whose natural-gradient step is pending (Kdim x ldk)
  int64_t ldk = 0, Kdim = 0;
  const T* w = nullptr;    // rho grad_E_Sigma   [Kdim]
  const T* r = nullptr;    // rho grad_E_mu      [Kdim]
  T* eta2 = nullptr;
  const T* Kinv = nullptr;
  int64_t ldm = 0;
  T* eta1 = nullptr;
  const T* kinv_mu0 = nullptr;
  T lr = T(0);
  T* HS = nullptr;           // hand-over slots of the helpers' partial tiles (sentinel-filled like the rest of the area)
  int32_t* sflags = nullptr; // one flag per helper (stride DAG_FS)
  T* fill = nullptr;         // hand-over set used by the PREVIOUS launch: refilled with the sentinel by the trailing workgroups
  int64_t fill_n = 0;
  int32_t nfill = 0;
  // batch-parallel step: the statistics arrive REDUCED over the ranks (all-reduced packed lower tiles, SY_PACK layout, and the
  // vector t = kappa' r); the prologue then only takes the eta step from them (Kdim = 0: no product, no helpers)
  const T* packed = nullptr;
  const T* tred = nullptr;
  // AGP_SPLIT_OVERLAP: those statistics arrive in groups of block columns, all-reduced one after the other on the communicator's
  // own stream; arrive[g * ARRIVE_STRIDE] holds the number of the last step whose group g is complete (t travels with group 0).
  // A workgroup waits for the group of its block column before it reads its tile (pro_arrival_gate).
  const int32_t* arrive = nullptr;
  int32_t arrive_want = 0;
  unsigned char grp[32] = {};
  unsigned char ks[32] = {};
  // k-split of the PRO_NEAR tiles next to the diagonal of block column c (the chain's two feeders and the two tiles the NEXT
  // feeders wait for): their product has to be done while the chain is at most a column or two away, whatever the column
  unsigned char kf[32] = {};  // k-slices per block column (1: the tile workgroup forms the whole product itself)
  // hyper-parameter iteration (round 4): C = S + K^-1 / 4 of the step, S = kappa' diag(w) kappa, both triangles (ld = ldm).  The
  // hyper-gradient's G_K then needs ONE m^3 product, C (Sigma K^-1), where it took kappa' H and K^-1 Sigma K^-1 (hypergrad)
  T* Cout = nullptr;
};

// The gate of AGP_SPLIT_OVERLAP.  It is NOT one of the task graph's abortable waits: what it waits for was enqueued before this
// launch and depends on nothing in it, and a workgroup that gave up here would leave its tile of eta2 un-stepped for the in-stream
// fallback (which relies on every prologue having completed).  Its limit (seconds) is a hang guard: info = -4.
constexpr int ARRIVE_STRIDE = 16;  // one word per 64 bytes
__device__ __forceinline__ void pro_arrival_gate(const int32_t* arrive, int g, int32_t want, int32_t* info) {
  if (threadIdx.x == 0) {
    long spins = 0;
    while ((int32_t)(__hip_atomic_load(arrive + g * ARRIVE_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - want) < 0) {
      __builtin_amdgcn_s_sleep(8);
      if (++spins > (1L << 23)) {
        atomicExch(info, -4);
        break;
      }
    }
  }
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // the collective's kernel wrote the tile after this launch started
}

// acc += sum over the 64-row chunks [q0, q1) of kappa of  (w .* kappa[:, R0 + .])' kappa[:, c0 + .]  -- 64-deep chunks staged in
// LDS as [r][k] tiles (the layout mma8 reads), two buffer pairs alternating, the next chunk's global loads in flight during the
// product.  sm: 4 * TILE * LDP elements.  All 512 threads; ends with a barrier (sm is free again).
constexpr int PRO_NEAR = 4;
// helpers of block column c (nn = its near-diagonal tiles): [near tiles: kf - 1 each][the others: ks - 1 each]
__host__ __device__ __forceinline__ int64_t pro_nhelp(int64_t nt, int64_t c, int ks, int kf) {
  const int64_t ntc = nt - c, nn = ntc < PRO_NEAR ? ntc : PRO_NEAR;
  return nn * (kf - 1) + (ntc - nn) * (ks - 1);
}
template <typename T>
__device__ __forceinline__ void pro_slice(const T* __restrict__ kap, int64_t ldk, const T* __restrict__ w, int64_t R0, int64_t c0,
                                          int64_t q0, int64_t q1, T* sm, Acc8<T>& acc) {
  constexpr int Q = TILE * TILE / CHOL_THREADS;  // 8
  const int r = threadIdx.x & 63, kb = threadIdx.x >> 6;
  const bool dg = R0 == c0;
  T va[Q], vb[Q], wv[Q];
  auto ldg = [&](int64_t qc) {
    const T* __restrict__ p = kap + (qc * TILE) * ldk;
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      const int k = kb + 8 * q;
      va[q] = p[k * ldk + R0 + r];
      vb[q] = dg ? T(0) : p[k * ldk + c0 + r];
      wv[q] = w[qc * TILE + k];
    }
  };
  if (q0 < q1) ldg(q0);
  int cur = 0;
  for (int64_t qc = q0; qc < q1; ++qc) {
    T* As = sm + cur * 2 * TILE * LDP;
    T* Bs = As + TILE * LDP;
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      const int k = kb + 8 * q;
      As[r * LDP + k] = va[q] * wv[q];
      Bs[r * LDP + k] = dg ? va[q] : vb[q];
    }
    __syncthreads();
    if (qc + 1 < q1) ldg(qc + 1);
    mma8<T>(As, Bs, acc);
    cur ^= 1;
  }
  __syncthreads();
}

// STEP: the launch of a CAVI step (no identity rows, X_k and L not wanted in their real homes) with those three facts known at
// compile time -- the general form carries the code and the registers of all of them through the chain.
// ROLE (round 4): 0 = the whole task graph in one kernel.  1 / 2 = the same graph as TWO kernels -- the chain workgroup(s) alone
// (ROLE 1, one workgroup per problem, on a high-priority stream of its own) and every other tile (ROLE 2, on the step's stream) --
// with the same flags, sentinels and epochs.  The chain's tile elimination needs 176-255 VGPRs, which held the merged kernel to one
// 512-thread workgroup per CU; the tile kernel alone compiles to <= 128 (f64) / <= 80 (f32) and runs 2 - 3 workgroups per CU, which
// is what a launch of 1500+ tiles (C3, C4) is short of.  The chain kernel may start before the work in front of the tile kernel
// has finished: it polls DagSync::go (stored by the tile kernel's first workgroup) and then drops its caches.
constexpr long long CHAIN_GO_TICKS = 4LL * 100000000LL;
#ifndef AGP_TILES_WAVES_F64
#define AGP_TILES_WAVES_F64 4
#endif
#ifndef AGP_TILES_WAVES_F32
#define AGP_TILES_WAVES_F32 6
#endif
// (with the prologue a tile workgroup stages four LDS tiles: 135 KB in f64 = one workgroup per CU whatever the registers,
//  68 KB in f32 = two)
template <typename T, int ROLE, bool PRO>
constexpr int dag_min_waves() {
  return ROLE != 2 ? 1 : PRO ? (sizeof(T) == 8 ? 2 : 4) : (sizeof(T) == 8 ? AGP_TILES_WAVES_F64 : AGP_TILES_WAVES_F32);
}
// ---- P = X' X as SINK work of the launch that produces X = L^-1 (round 5) ------------------------------------
// A launch with the nt identity block rows (nx = nt) leaves X in the hand-over slots of those rows: slot (nt + ne + i, k) holds
// E(i, k) = X(k, i)' once block column k has been processed.  Both consumers of such a launch need X' X next -- K^-1 = X' X at a
// kernel refresh, Sigma = Xa' Xa for the hyper-gradient -- and until round 4 that was two more launches (k_xtx_bal 20.7 us +
// k_xtx_bal_reduce 8.7 us, twice per hyper-on iteration) BEHIND a launch in which 255 CUs wait for the chain.  Now the launch ends
// with nt (nt + 1) / 2 PRODUCT workgroups, one per lower tile (i, j):
//     P(i, j) = sum_{k >= i} E(i, k) E(j, k)'                         (k ascending: fixed order, bitwise reproducible)
// They are the LAST workgroups of the grid: they only get a slot when everything else has been dispatched, by which time most of
// their operands are there; each waits for the flags of its two operand tiles like any tile workgroup (abortable), and what they
// wait for has lower workgroup indices (the dispatch-order argument is untouched).  The tile is stored to `out` in both
// triangles (diagonal tiles: the lower half is the truth).  Before it starts, a product workgroup also stores its share of the
// sentinels of the OTHER hand-over set (fill): the launches whose riders did that (k_xtx_bal, k_syrk_tn) are gone from this path.
// If the launch is aborted the in-stream fallback forms P itself (SafeSrc::pout).
template <typename T>
struct ProdArgs {
  T* out = nullptr;     // n x n, leading dimension ld; nullptr: no product workgroups
  int64_t ld = 0;
  int64_t base = 0;     // index of the first product workgroup in the grid
  T* fill = nullptr;    // (nullable) hand-over set to refill with the sentinel
  int64_t fill_n = 0;
  // (nullable) log det of the factored matrix' Cholesky factor, sum_i log L_ii over the first nvalid rows, by the LAST product
  // workgroup: it has waited for the last block column, so every X_k = L_kk^-1 sits in its hand-over slot, and the diagonal of a
  // triangular inverse is the reciprocal of the factor's (the launch behind a kernel refresh that did this, k_logdiag_sum /
  // k_xtx_bal_reduce's rider, is gone).  status: the [info | infoK | flags | orderK] words of k_logdiag_sum, same bookkeeping.
  double* ld_out = nullptr;
  int32_t* status = nullptr;
};

template <typename T, bool FUSED, bool BATCH = false, bool TRACE = false, bool STEP = false, bool PRO = false, int ROLE = 0>
__global__ __launch_bounds__(CHOL_THREADS, (dag_min_waves<T, ROLE, PRO>()))
void k_chol_dag(CholBatch<T> bt, int nb, int64_t fstride, int64_t ld, int64_t ldx,
                                                           int64_t lde, int64_t ne, int64_t nt,
                                                           int32_t* __restrict__ info, int64_t nvalid, int32_t* flags,
                                                           int32_t epoch, unsigned long long* trace, T* H, int64_t hstride,
                                                           int64_t nx_, const T* __restrict__ erow, int opts, DagSync sync,
                                                           ProArgs<T> pro = ProArgs<T>{}, EpiArgs<T> epi = EpiArgs<T>{},
                                                           ProdArgs<T> prod = ProdArgs<T>{}) {
  static_assert(!PRO || (FUSED && !BATCH), "the prologue exists for single-problem launches with a chain workgroup only");
  static_assert(ROLE == 0 || (FUSED && STEP && !TRACE), "the split launch exists for the CAVI step's launches");
  // opts bit 0: the chain also stores X_k to its real home (a single block column wanting its inverse: no identity rows run)
  //      bit 1: the factor L is wanted in its real home A as well (K's factor, the potrf entry points); the CAVI step only
  //             consumes the extension rows W, v and never reads L itself, so its launches skip those stores
  const int64_t nx = STEP ? 0 : nx_;
  const int write_x = STEP ? 0 : (opts & 1);
  const bool store_l = STEP ? false : (opts & 2) != 0;
  // nb > 1: nb independent problems of the same shape (the latents of a small multi-class model) in ONE launch, their
  // workgroups interleaved (linear index = tile * nb + problem) so that the chains of all problems start at once and the
  // per-XCD dispatch order stays a topological order of every graph.  Each problem has its own flags (fstride apart).  The
  // chains publish X_k before they block on a feeder of a later index (late_feed below), so no residency bound is needed.
  // (BATCH is a template parameter so that the single-problem instantiation keeps constant kernel-argument offsets)
  const int prob = BATCH ? (int)(blockIdx.x % (unsigned)nb) : 0;
  const int64_t bidx = BATCH ? (int64_t)(blockIdx.x / (unsigned)nb) : (int64_t)blockIdx.x;
  T* A = bt.A[prob];
  T* X = bt.X[prob];
  T* __restrict__ Dg = bt.Dg[prob];
  T* E = bt.E[prob];
  if (BATCH) erow = bt.R[prob];
  flags += prob * fstride;
  // hand-over area of this problem (sentinel-filled by the host): slots of 64x64 elements
  //   HX[k] = X_k  |  HP[2k] = parked (k, k-1), HP[2k + 1] = parked (k, k)  |  HL[(R * nt + c)] tile (R, c), R in [0, nt + ne + nx)
  // (the part a launch can touch is a prefix of the problem's region: only that much is refilled afterwards)
  constexpr int64_t SLOT = TILE * TILE;
  T* HX = H + prob * hstride;
  T* HP = HX + nt * SLOT;
  T* HL = HP + 2 * nt * SLOT;
  // erow (optional): row 0 of the LAST extension block is taken from this vector and its rows 1-63 as zero, instead of being
  // read from E (the CAVI step appends [eta1' ; 0]: saves the launch that used to write them)
  // nx = nt: also X = L^-1 in full.  L^-T = I L^-T, so nt more extension block rows holding the identity give X' column by
  // column with the same task graph and off the critical path (row i: tiles (i, c), c >= i; the others stay zero and have
  // no workgroup).  They recurse through their hand-over slots (rows nt + ne + i) and are stored transposed into X.
  __shared__ __attribute__((aligned(16))) T sm[((FUSED && ROLE != 2) || PRO ? 4 : 2) * TILE * LDP];
  __shared__ __attribute__((aligned(16))) T sc[(ROLE == 2 && !PRO) ? TILE : SC_ELEMS];
  __shared__ T piv[TILE];
  __shared__ int wait_ok, pf_ok, pf_bad;
  if (ROLE == 1) {  // the chain as a kernel of its own: wait until the tile kernel runs (= everything before it on its stream is done)
    if (threadIdx.x == 0) {
      wait_ok = 1;
      if (sync.here) __hip_atomic_fetch_add(sync.here, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);  // resident: the tile kernel may start
      unsigned spins = 0;
      const long long t0 = (long long)wall_clock64();
      // ordered compare, not equality: the release word is one per context and every split launch stores its own sequence number
      // into it.  A chain kernel that is dispatched late (its launch was aborted and the fallback has run, the next tile kernel has
      // already stored go_val + 1) must not wait for a value that has come and gone -- it would spin out its limit and latch -1 on a
      // launch that has nothing to do with it.  If the word has moved PAST go_val this launch is over: count out, do no work.
      while (sync.go) {
        const int32_t d = (int32_t)(__hip_atomic_load(sync.go, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - sync.go_val);
        if (d == 0) break;
        if (d > 0) {  // a later launch has been released already
          wait_ok = 0;
          break;
        }
        __builtin_amdgcn_s_sleep(4);
        // limit on the device's 100 MHz clock (round 6; it was 2^26 polls, about 100 s): what sits between this kernel's start and
        // the tile kernel's is at most the kernels of one step plus a fallback with the inverse (~0.2 s); CHAIN_GO_TICKS = 4 s
        if ((++spins & 1023u) == 0 && (long long)wall_clock64() - t0 > CHAIN_GO_TICKS) {  // the tile kernel never started.  Treated
          atomicExch(info, -1);      // like a lost dependency: the latch sends the launch to the in-stream fallback (the tiles, should
          wait_ok = 0;               // they still come, give up on their own bounded waits for the chain)
#ifdef AGP_DEBUG_PTRS
          atomicAdd(&agp_dag_diag[5], 1ull);
#endif
          break;
        }
      }
    }
    __syncthreads();
    if (!wait_ok) {
      chain_count_out(sync);
      return;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // this kernel's caches may predate what those kernels wrote
  }
  T* bufA = sm;
  T* bufB = sm + TILE * LDP;
  const int tid = threadIdx.x;
  // column-major tile numbering: column c holds its diagonal tile, rows c+1..nt-1, then the ne extension blocks
  // (PRO: the (nt - c)(ks[c] - 1) helper workgroups of a column come right before its tiles; refill workgroups after everything)
  int64_t b = bidx, c = 0;
  bool helper = false;
  int64_t hbase = 0;  // PRO: helper slots of the columns before c
  const bool is_prod = !BATCH && !STEP && ROLE != 1 && prod.out != nullptr && bidx >= prod.base;  // P = X' X tile (see ProdArgs)
  //  the host clears the dirty mark of the OTHER hand-over set when it enqueues a launch with product workgroups and no
  // prologue, trusting THESE workgroups to refill it (prod.fill, first thing they do below).  That holds only while no exit can
  // precede the fill for an is_prod workgroup -- true for ROLE 0 (the chain-kernel exits above are ROLE 1, the STEP forms have no
  // product workgroups); the hand-over validates itself by sentinels, so a missed refill would be accepted as data, not reported
  static_assert(ROLE == 0 || STEP, "product workgroups (ProdArgs) must reach their sentinel refill unconditionally: ROLE 0 only");
  if (is_prod) {
    b = 0;  // (decoded below, once the flag pointers are in place)
  } else if (PRO && ROLE == 1) {
    // the chain kernel: tile (0, 0), whose place in the tile kernel comes right after the helpers of block column 0 (hbase = 0)
  } else if (PRO) {
    for (;;) {
      if (c == nt) {  // trailing workgroups: the hand-over set the launch before this one used gets its sentinels back
        const T sv = __builtin_bit_cast(T, Sent<T>::bits);
        for (int64_t i = b * CHOL_THREADS + tid; i < pro.fill_n; i += (int64_t)pro.nfill * CHOL_THREADS) pro.fill[i] = sv;
        return;
      }
      const int64_t nh = pro_nhelp(nt, c, pro.ks[c], pro.kf[c]), ntile = nt - c + ne + (nx ? c + 1 : 0);
      if (b < nh) {
        helper = true;
        break;
      }
      b -= nh;
      if (b < ntile) break;
      b -= ntile;
      hbase += nh;
      ++c;
    }
  } else {
    while (b >= nt - c + ne + (nx ? c + 1 : 0)) {
      b -= nt - c + ne + (nx ? c + 1 : 0);
      ++c;
    }
  }
  // development aid (AGP_PRO_TRACE): wall-clock stamps of the prologue -- chain [0..7], tiles (R, c) with R, c < 4 at
  // 64 + 8 (4 R + c), end of factor(k) at 512 + k, the first helper of tile (0, 0) at 1024
#define PRO_TS(i) \
  if (PRO && trace && tid == 0) trace[i] = wall_clock64()
  if (PRO && helper) {  // one k-slice of S(R, c): no dependencies, one store of the partial tile, done
    const int64_t ntc = nt - c, nn = ntc < PRO_NEAR ? ntc : PRO_NEAR;
    const int kfc = pro.kf[c], kso = pro.ks[c];
    const bool near = b < nn * (kfc - 1);
    const int ksc = near ? kfc : kso;
    const int64_t bo = near ? b : b - nn * (kfc - 1);
    const int64_t tb = (near ? 0 : nn) + bo / (ksc - 1), sl = 1 + bo % (ksc - 1), nq = pro.Kdim / TILE;
    Acc8<T> S;
    S.zero();
    if (c == 0 && b == 0) PRO_TS(1024);
    pro_slice<T>(pro.kap, pro.ldk, pro.w, (c + tb) * TILE, c * TILE, sl * nq / ksc, (sl + 1) * nq / ksc, sm, S);
    if (c == 0 && b == 0) PRO_TS(1025);
    T* hs = pro.HS + (hbase + b) * (TILE * TILE);
    {
      int e = 0;
      acc8_foreach<T>(S, [&](int r, int cc, T& val) {
        (void)r;
        (void)cc;
        __hip_atomic_store(hs + pro_slot_off<T>(tid, e++), val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      });
    }
    dag_signal(pro.sflags + (hbase + b) * DAG_FS, epoch);
    return;
  }
  const bool diag = b == 0, ext = b >= nt - c, idr = b >= nt - c + ne;  // idr: identity row i = b - (nt - c + ne) <= c
  const int64_t R = idr ? nt + ne + (b - (nt - c + ne)) : ext ? nt + (b - (nt - c)) : c + b;  // block row in [0, nt+ne+nx)
  T* rowp = idr ? nullptr : ext ? E + (R - nt) * TILE * lde : A + R * TILE * ld;  // the block row's real home (none for idr)
  const int64_t ldr = ext ? lde : ld, c0 = c * TILE;
  int32_t* ready = flags;
  int32_t* xready = flags + (nt + ne + nx) * nt * DAG_FS;
  int32_t* pre1 = xready + nt * DAG_FS;  // FUSED: tile (c, c-1) with all its pending updates is parked in place for the chain
  int32_t* pre2 = pre1 + nt * DAG_FS;    // FUSED: diagonal tile (c, c) with the updates of columns < c-1 parked in place
  int32_t* abortf = pre2 + nt * DAG_FS;
#define DAG_TR(slot) \
  if (TRACE && tid == 0) trace[bidx * 8 + (slot)] = wall_clock64()
#define DAG_TRC(col, slot) \
  if (TRACE && tid == 0) trace[chain_slot(col, nt, ne, nx) * 8 + (slot)] = wall_clock64()
  if (is_prod) {
    const int64_t pidx = bidx - prod.base, nprod = nt * (nt + 1) / 2;
    if (prod.fill) {  // this workgroup's share of the other hand-over set's sentinels (fire-and-forget)
      const T sv = __builtin_bit_cast(T, Sent<T>::bits);
      for (int64_t i = pidx * CHOL_THREADS + tid; i < prod.fill_n; i += nprod * CHOL_THREADS) prod.fill[i] = sv;
    }
    int64_t pi, pj;
    tri_index(pidx, pi, pj);  // lower tile (pi, pj), pj <= pi
    const bool pdiag = pi == pj;
    const int64_t Ri = nt + ne + pi, Rj = nt + ne + pj;
    Acc8<T> pacc;
    pacc.zero();
    for (int64_t k = pi; k < nt; ++k) {
      if (!dag_wait(ready + (Ri * nt + k) * DAG_FS, pdiag ? nullptr : ready + (Rj * nt + k) * DAG_FS, epoch, abortf, info, &wait_ok))
        return;
      if (pdiag) load_tile_lds_hv<T>(HL + (Ri * nt + k) * SLOT, bufA);
      else load_tiles_lds_hv2<T>(HL + (Ri * nt + k) * SLOT, bufA, HL + (Rj * nt + k) * SLOT, bufB);
      __syncthreads();
      mma8<T>(bufA, pdiag ? bufA : bufB, pacc);
      __syncthreads();
    }
    const int64_t i0 = pi * TILE, j0 = pj * TILE;
    if (pdiag) {  // the lower half is the truth, the upper half its mirror image
      acc8_foreach<T>(pacc, [&](int r, int cc, T& val) { bufA[r * LDP + cc] = val; });
      __syncthreads();
      acc8_foreach<T>(pacc, [&](int r, int cc, T& val) {
        prod.out[(i0 + r) * prod.ld + i0 + cc] = cc > r ? bufA[cc * LDP + r] : val;
      });
    } else {
      acc8_foreach<T>(pacc, [&](int r, int cc, T& val) {
        prod.out[(i0 + r) * prod.ld + j0 + cc] = val;
        prod.out[(j0 + cc) * prod.ld + i0 + r] = val;
      });
    }
    if (prod.ld_out && pidx == nprod - 1) {  // tile (nt - 1, nt - 1): every xready flag is up
      __syncthreads();
      double lsum = 0.0;
      for (int64_t i = tid; i < nvalid; i += CHOL_THREADS) {
        const T* px = HX + (i / TILE) * SLOT + (i % TILE) * (TILE + 1);
        const T xv = hv_settle<T>(px, __hip_atomic_load(px, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        lsum -= log((double)xv);
      }
      lsum = block_sum<double>(lsum, reinterpret_cast<double*>(sc));
      if (tid == 0) {
        prod.ld_out[0] = lsum;
        int32_t* st_ = prod.status;
        if (st_) {
          const int32_t i0_ = __hip_atomic_load(st_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          const int32_t i1_ = __hip_atomic_load(st_ + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          const int32_t i2_ = __hip_atomic_load(st_ + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (i1_ != 0 && st_[3] == 0) st_[3] = (i0_ != 0 || i2_ != 0) ? 2 : 1;
        }
      }
    }
    return;
  }
  DAG_TR(0);
  const bool chain = FUSED && c == 0 && b == 0;  // (PRO: no longer workgroup 0 -- the helpers of column 0 come first)
  if (STEP && !BATCH && sync.started && chain && tid == 0 && ROLE != 1)
    __hip_atomic_store(sync.started, sync.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  if (ROLE == 2 && chain) {  // the chain's place in the tile kernel: it is the first workgroup to run, so it releases the chain kernel
    if (tid == 0 && sync.go) __hip_atomic_store(sync.go, sync.go_val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    // PRO: this workgroup -- not the chain kernel -- takes the prologue of tile (0, 0) and PARKS A(0, 0) = -2 eta2 for the chain
    // like any feeder (the unused diagonal park slot of block column 0): every store of the natural-gradient step then belongs to
    // the tile kernel's stream, and "eta is completely stepped even when the factorisation is aborted" holds for split launches too
    if (!PRO) return;
  }
  if (ROLE == 1 && !chain) return;
  Acc8<T> acc;
  if (PRO && ROLE == 1) {
    // (tile (0, 0) arrives parked, see above)
  } else if (PRO && !ext) {
    // ---- prologue of a matrix tile: S(R, c) (own k-slice + the helpers' partial tiles), the eta2 step, A = -2 eta2 -> acc
    const int64_t ntc_ = nt - c, nn_ = ntc_ < PRO_NEAR ? ntc_ : PRO_NEAR;
    const int ksc = b < nn_ ? pro.kf[c] : pro.ks[c];
    const int64_t nq = pro.Kdim / TILE;
    const int tsb = chain ? 0 : (R < 4 && c < 4) ? 64 + 8 * (4 * (int)R + (int)c) : 2040;
    PRO_TS(tsb);
    acc.zero();
    pro_slice<T>(pro.kap, pro.ldk, pro.w, R * TILE, c0, 0, nq / ksc, sm, acc);
    if (pro.packed) {  // the reduced statistic of this tile (what k_eta2_from_packed reads)
      if (pro.arrive) pro_arrival_gate(pro.arrive, pro.grp[c], pro.arrive_want, info);
      const T* __restrict__ tp = pro.packed + pack_index(R, c, nt) * (TILE * TILE);
      acc8_foreach<T>(acc, [&](int r, int cc, T& val) { val = tp[r * TILE + cc]; });
    }
    PRO_TS(tsb + 1);
    // eta2 and K^-1 of the tile: in flight while the helpers' tiles are fetched
    Acc8<T> e2v, kiv;
    acc8_foreach<T>(e2v, [&](int r, int cc, T& val) { val = pro.eta2[(R * TILE + r) * pro.ldm + c0 + cc]; });
    acc8_foreach<T>(kiv, [&](int r, int cc, T& val) { val = pro.Kinv[(R * TILE + r) * pro.ldm + c0 + cc]; });
    if (ksc > 1) {
      const int64_t h0 = hbase + (b < nn_ ? b * (pro.kf[c] - 1) : nn_ * (pro.kf[c] - 1) + (b - nn_) * (pro.ks[c] - 1));
      if (tid < ksc - 1) {  // one poller per helper; helpers have lower workgroup indices and wait for nothing: they always arrive
        long spins = 0;
        while (__hip_atomic_load(pro.sflags + (h0 + tid) * DAG_FS, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch) {
          __builtin_amdgcn_s_sleep(1);
          if (++spins > (1L << 28)) {  // minutes: the device is gone
            atomicExch(info, -3);
            break;
          }
        }
      }
      __syncthreads();
      PRO_TS(tsb + 2);
      // the helpers' tiles: thread-major slots (pro_slot_off), fetched with 16-byte coherent loads, all of them in flight at once
      // (8-byte loads, four tiles at a time, took the chain 4.9 us for seven tiles), then added in helper order
      constexpr int VEC = 16 / (int)sizeof(T), NV = 8 / VEC;
      u32x4 pv[7][NV];
#pragma unroll
      for (int q = 0; q < 7; ++q)
        if (q < ksc - 1) {
          const T* hs = pro.HS + (h0 + q) * (TILE * TILE);
#pragma unroll
          for (int j = 0; j < NV; ++j) pv[q][j] = load16_sc1(hs + pro_slot_off<T>(tid, j * VEC));
        }
      wait_vmcnt0();
#pragma unroll
      for (int q = 0; q < 7; ++q)
        if (q < ksc - 1) {
          const T* hs = pro.HS + (h0 + q) * (TILE * TILE);
          int e = 0;
          acc8_foreach<T>(acc, [&](int r, int cc, T& val) {
            (void)r;
            (void)cc;
            const T got = vec16_elem<T>(pv[q][e / VEC], e % VEC);
            val += hv_settle<T>(hs + pro_slot_off<T>(tid, e), got);
            ++e;
          });
        }
    }
    PRO_TS(tsb + 3);
    if (pro.Cout) {
      int e = 0;
      acc8_foreach<T>(acc, [&](int r, int cc, T& val) {
        const T cv = val + T(0.25) * kiv.a[e >> 2][e & 3];
        ++e;
        pro.Cout[(R * TILE + r) * pro.ldm + c0 + cc] = cv;
        if (!diag) pro.Cout[(c0 + cc) * pro.ldm + R * TILE + r] = cv;
      });
    }
    {
      int e = 0;
      const T lr = pro.lr;
      acc8_foreach<T>(acc, [&](int r, int cc, T& val) {  // the step of the fused epilogue (k_syrk_tn<SY_ETA2>), same operations
        const int mi = e >> 2, rr = e & 3;
        ++e;
        T e2 = e2v.a[mi][rr];
        const T g = -(val + T(0.5) * kiv.a[mi][rr]) - e2;
        e2 += lr * g;
        val = e2;
      });
    }
    if (diag) {  // the lower half is the truth, the upper half its mirror image (as the fused epilogue does it)
      acc8_foreach<T>(acc, [&](int r, int cc, T& val) { bufA[r * LDP + cc] = val; });
      __syncthreads();
      acc8_foreach<T>(acc, [&](int r, int cc, T& val) {
        if (cc > r) val = bufA[cc * LDP + r];
        pro.eta2[(c0 + r) * pro.ldm + c0 + cc] = val;
        val *= T(-2);
      });
      __syncthreads();
    } else {
      acc8_foreach<T>(acc, [&](int r, int cc, T& val) {
        pro.eta2[(R * TILE + r) * pro.ldm + c0 + cc] = val;
        pro.eta2[(c0 + cc) * pro.ldm + R * TILE + r] = val;
        val *= T(-2);
      });
    }
    PRO_TS(tsb + 4);
  } else if (idr) {
    const bool on_diag = (R - nt - ne) == c;
    acc8_foreach<T>(acc, [&](int r, int cc, T& val) { val = (on_diag && r == cc) ? T(1) : T(0); });
  } else if (PRO && ext && R == nt + ne - 1) {
    // ---- prologue of the [eta1' ; 0] tile: t = kappa[:, c0 ..]' r (eight row groups, fixed order), eta1 step, row 0 <- eta1
    if (pro.tred && pro.arrive) pro_arrival_gate(pro.arrive, 0, pro.arrive_want, info);
    const int col = tid & 63, grp = tid >> 6;
    T s0 = T(0), s1 = T(0), s2 = T(0), s3 = T(0);
    int64_t k = grp;
    for (; k + 24 < pro.Kdim; k += 32) {
      const T a0 = pro.kap[k * pro.ldk + c0 + col], a1 = pro.kap[(k + 8) * pro.ldk + c0 + col];
      const T a2 = pro.kap[(k + 16) * pro.ldk + c0 + col], a3 = pro.kap[(k + 24) * pro.ldk + c0 + col];
      s0 += a0 * pro.r[k];
      s1 += a1 * pro.r[k + 8];
      s2 += a2 * pro.r[k + 16];
      s3 += a3 * pro.r[k + 24];
    }
    for (; k < pro.Kdim; k += 8) s0 += pro.kap[k * pro.ldk + c0 + col] * pro.r[k];
    sc[grp * TILE + col] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (grp == 0) {
      T t = T(0);
#pragma unroll
      for (int q = 0; q < CHOL_THREADS / 64; ++q) t += sc[q * TILE + col];
      if (pro.tred) t = pro.tred[c0 + col];  // (pro.arrive: group 0 was waited for above)
      T e = pro.eta1[c0 + col];
      e += pro.lr * (t + (pro.kinv_mu0 ? pro.kinv_mu0[c0 + col] : T(0)) - e);
      pro.eta1[c0 + col] = e;
      sc[CHOL_THREADS + col] = e;
    }
    __syncthreads();
    acc8_foreach<T>(acc, [&](int r, int cc, T& val) { val = r == 0 ? sc[CHOL_THREADS + cc] : T(0); });
    __syncthreads();
  } else if (erow && ext && R == nt + ne - 1) {
    acc8_foreach<T>(acc, [&](int r, int cc, T& val) { val = r == 0 ? erow[c0 + cc] : T(0); });
  } else {
    acc8_foreach<T>(acc, [&](int r, int cc, T& val) { val = rowp[r * ldr + c0 + cc]; });
  }
  if (PRO && ROLE == 2 && chain) {  // A(0, 0) of the stepped eta2 -> the chain kernel
    T* park = HP + SLOT;
    acc8_foreach<T>(acc, [&](int r, int cc, T& val) {
      __hip_atomic_store(park + r * TILE + cc, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    });
    dag_signal(pre2, epoch);
    return;
  }
  if (ROLE != 2 && chain) {
    // ---- the chain: ONE workgroup carries the critical path through all columns, so that per column only the tile
    // factorisation and two 64^3 products are serial:  factor(c) -> L(c+1,c) = T X_c' -> S = D - L L' -> factor(c+1).
    // T = tile (c+1, c) and D = tile (c+1, c+1) arrive with all their other updates already applied by feeder workgroups.
    if (PRO && ROLE == 1) {
      if (!dag_wait(pre2, nullptr, epoch, abortf, info, &wait_ok)) {
        chain_count_out(sync);
        return;
      }
      load_tile_lds_hv<T>(HP + SLOT, bufA);
    } else {
      acc8_foreach<T>(acc, [&](int r, int cc, T& val) { bufA[r * LDP + cc] = val; });
    }
    __syncthreads();
    T* bufC = sm + (FUSED ? 2 : 0) * TILE * LDP;
    T* bufD = sm + (FUSED ? 3 : 0) * TILE * LDP;
    for (int64_t k = 0; k < nt; ++k) {
      const int64_t k0 = k * TILE;
      T* trow = A + (k + 1) * TILE * ld;  // block row k+1 (only touched while k + 1 < nt)
      ChainPrefetch<T> pf;
      pf.src = k + 1 < nt ? HP + 2 * (k + 1) * SLOT : nullptr;
      pf.bad = &pf_bad;
      pf.dT = bufC;
      pf.dD = bufD;
      pf.f1 = pre1 + (k + 1) * DAG_FS;
      pf.f2 = pre2 + (k + 1) * DAG_FS;
      pf.epoch = epoch;
      pf.ok = &pf_ok;
      pf.dgsrc = k >= 1 ? bufD : nullptr;  // after the swap below bufD is where L_{k-1,k-1} still sits
      pf.gDg = Dg + (k - 1) * TILE * TILE;
      pf.lsrc = (k >= 1 && store_l) ? bufC : nullptr;  // L(k, k-1) stays in bufC until the prefetch of round 5 overwrites it
      pf.gL = A + k0 * ld + (k0 - TILE);
      pf.ld_i = (int)ld;
      pf.coh = ROLE == 1;
      if (tid == 0) pf_ok = pf_bad = 0;
      factor_diag_tile_2lvl<T, ChainPrefetch<T>>(bufA, bufB, sc, piv, info, k0, nvalid, pf);
      DAG_TRC(k, 2);
      PRO_TS(512 + k);
      {  // X_k out (coherent): all LDS reads first, then the stores back to back (a read-store-read-store loop exposed the
         // LDS latency eight times: 0.6 us on the chain)
        T xv[TILE * TILE / CHOL_THREADS];
        int tl = tid;
        asm volatile("" : "+v"(tl));  // keeps these addresses out of the registers that live through the factorisation
#pragma unroll
        for (int q = 0; q < TILE * TILE / CHOL_THREADS; ++q) {
          const int e = tl + q * CHOL_THREADS;
          xv[q] = bufB[(e >> 6) * LDP + (e & 63)];
        }
#pragma unroll
        for (int q = 0; q < TILE * TILE / CHOL_THREADS; ++q) {
          const int e = tl + q * CHOL_THREADS;
          __hip_atomic_store(HX + k * SLOT + e, xv[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (write_x) X[(k0 + (e >> 6)) * ldx + k0 + (e & 63)] = xv[q];
        }
        // (the real X block is only read when the full inverse was requested, and then the identity-row tile (k, k) writes it)
      }
      if (k + 1 == nt) {
        for (int e = tid; e < TILE * TILE; e += CHOL_THREADS) {
          if (ROLE == 1) __hip_atomic_store(Dg + k * TILE * TILE + e, bufA[(e >> 6) * LDP + (e & 63)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          else Dg[k * TILE * TILE + e] = bufA[(e >> 6) * LDP + (e & 63)];
        }
        dag_signal(xready + k * DAG_FS, epoch);
        DAG_TRC(k, 3);
        if (ROLE == 1) chain_count_out(sync);
        return;
      }
      const bool late_feed = !pf_ok || pf_bad;
      if (late_feed) {  // the feeders were not done (or not yet visible) when the side job looked: fetch now
        // X_k is published BEFORE this wait: the second feeder, tile (k+1, k+1), is the first workgroup of block column k + 1, and
        // it may not have a slot yet while all of column k's workgroups sit waiting for X_k (a launch whose block column does not
        // fit into the resident window: 8 problems of 34 tiles each).  With it, every resident workgroup can finish on what the
        // chains have already published, so the dispatch order alone guarantees progress -- no residency bound is needed.
        dag_signal(xready + k * DAG_FS, epoch);
        DAG_TRC(k, 3);
        if (!dag_wait(pf.f1, pf.f2, epoch, abortf, info, &wait_ok)) {
          if (ROLE == 1) chain_count_out(sync);
          return;
        }
        load_tiles_lds_hv2<T>(HP + 2 * (k + 1) * SLOT, bufC, HP + (2 * (k + 1) + 1) * SLOT, bufD);
        __syncthreads();
      }
      if (TRACE && tid == 0) trace[chain_slot(k + 1, nt, ne, nx) * 8 + 7] = (unsigned long long)pf_ok;
      DAG_TRC(k + 1, 4);
      // both products skip what the structure makes zero or redundant (a 64^3 f64 product is MFMA-throughput bound on one CU:
      // 2.2 us; these take 40/64 and 48/64 of it):
      //   L = T X_k'  with X_k lower triangular: column tile j only needs k < 16 (j + 1); wave -> row tile w >> 1 and the column
      //   tile pair {3, 0} or {2, 1} (20 MFMAs each way)
      typedef typename Mfma<T>::acc_t acc_t;
      const int lane = tid & 63, wave = tid >> 6;
      {
        const int ri = wave >> 1, cA = (wave & 1) ? 2 : 3, cB = 3 - cA;
        acc_t oa, ob;
#pragma unroll
        for (int r = 0; r < 4; ++r) oa[r] = ob[r] = T(0);
        oa = mma_tile16<T>(bufC + 16 * ri * LDP, bufB + 16 * cA * LDP, 16 * (cA + 1), oa, false, lane);
        ob = mma_tile16<T>(bufC + 16 * ri * LDP, bufB + 16 * cB * LDP, 16 * (cB + 1), ob, false, lane);
        // X_k went out before the product: its stores are acknowledged by now, so publishing it here costs the chain one
        // barrier instead of a store round trip (the column's other tiles see X_k ~2 us later; they have ~10 us of slack)
        if (!late_feed) {
          dag_signal(xready + k * DAG_FS, epoch);
          DAG_TRC(k, 3);
        } else {
          __syncthreads();  // (every wave has read T before it is overwritten in place)
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = 16 * ri + Mfma<T>::row(lane, r);
          bufC[row * LDP + 16 * cA + (lane & 15)] = oa[r];  // in place of T (every wave is past the barrier above)
          bufC[row * LDP + 16 * cB + (lane & 15)] = ob[r];
          T* hl = HL + ((k + 1) * nt + k) * SLOT + row * TILE + (lane & 15);
          __hip_atomic_store(hl + 16 * cA, oa[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(hl + 16 * cB, ob[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          // (its real home in A is written by the idle waves during the next factorisation: ChainPrefetch rounds 2, 3)
        }
      }
      __syncthreads();
      DAG_TRC(k + 1, 5);
      //   S = D - L L': only the ten 16x16 tiles on and below the diagonal (the factorisation never reads above it); every
      //   wave takes one, waves 0 and 1 a second (SIMDs 0, 1: three tiles, SIMDs 2, 3: two).  In place in bufD.
      {
        // tiles in row-major lower order: (0,0) (1,0) (1,1) (2,0) (2,1) (2,2) (3,0) (3,1) (3,2) (3,3)
        const int ti_[10] = {0, 1, 1, 2, 2, 2, 3, 3, 3, 3}, tj_[10] = {0, 0, 1, 0, 1, 2, 0, 1, 2, 3};
#pragma unroll
        for (int rep = 0; rep < 2; ++rep) {
          const int tl = rep == 0 ? 2 + wave : wave;  // second pass: tiles 0, 1 on waves 0, 1
          if (rep == 1 && wave >= 2) break;
          int i2 = 0, j2 = 0;
#pragma unroll
          for (int q = 0; q < 10; ++q)
            if (q == tl) {
              i2 = ti_[q];
              j2 = tj_[q];
            }
          acc_t a;
#pragma unroll
          for (int r = 0; r < 4; ++r) a[r] = bufD[(16 * i2 + Mfma<T>::row(lane, r)) * LDP + 16 * j2 + (lane & 15)];
          a = mma_tile16<T>(bufC + 16 * i2 * LDP, bufC + 16 * j2 * LDP, TILE, a, true, lane);
#pragma unroll
          for (int r = 0; r < 4; ++r) bufD[(16 * i2 + Mfma<T>::row(lane, r)) * LDP + 16 * j2 + (lane & 15)] = a[r];
        }
      }
      dag_signal(ready + ((k + 1) * nt + k) * DAG_FS, epoch);
      {  // the tile to factor next is in bufD; L_kk stays where it is (now called bufD) until the side job has stored it
        T* t0 = bufA;
        bufA = bufD;
        bufD = t0;
      }
      DAG_TRC(k + 1, 1);
    }
    if (ROLE == 1) chain_count_out(sync);
    return;
  }
  if (ROLE == 1) return;
  const bool f1 = FUSED && b == 1 && !ext;  // tile (c+1, c): feeds the chain instead of waiting for X_c itself
  const bool f2 = FUSED && diag;            // diagonal tile (c, c), c >= 1: the chain applies the last update itself
  const int64_t jend = f2 ? c - 1 : c;
  // EPI: the last tile of a kappa block row gathers the row's statistics while it fetches the tiles left of it
  const bool epi_row = PRO && epi.on && ext && !idr && c == nt - 1 && R < nt + ne - 1;
  T epi_ss = T(0), epi_dt = T(0), epi_sk = T(0), epi_y = T(0);
  const int er = tid >> 3, es = (tid & 7) * 8;  // thread -> row er, columns es .. es + 7 of a 64 x 64 tile
  if (epi_row) {  // what the rows need besides W and v is fetched now, far from the launch's tail: the K~ slices of row er
    const int64_t i = (R - nt) * TILE + er;
    if (!epi.use_kt && i < epi.B)
      for (int q = (tid & 7); q < epi.nslices; q += 8) epi_sk += epi.pk[q * epi.ldp + i];
    if ((tid & 7) == 0 && i < epi.B) epi_y = epi.y[epi.idx ? epi.idx[i] : i];  // ... and its target
  }
  auto epi_acc = [&](const T* Wt /* LDS [r][k] */, const T* vslot /* hand-over slot of the v tile of this block column */) {
    if (tid < TILE) piv[tid] = hv_settle<T>(vslot + tid, __hip_atomic_load(vslot + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const T a = Wt[er * LDP + es + q];
      epi_ss += a * a;
      epi_dt += a * piv[es + q];
    }
  };
  for (int64_t j = idr ? R - nt - ne : 0; j < jend; ++j) {  // row i of the identity is zero left of block column i
    if (!dag_wait(ready + (R * nt + j) * DAG_FS, diag ? nullptr : ready + (c * nt + j) * DAG_FS, epoch, abortf, info, &wait_ok))
      return;
    if (j == c - 1) DAG_TR(4);
    if (diag) load_tile_lds_hv<T>(HL + (R * nt + j) * SLOT, bufA);
    else load_tiles_lds_hv2<T>(HL + (R * nt + j) * SLOT, bufA, HL + (c * nt + j) * SLOT, bufB);
    __syncthreads();
    if (j == c - 1) DAG_TR(5);
    if (epi_row) epi_acc(bufA, HL + ((nt + ne - 1) * nt + j) * SLOT);  // (ends behind a barrier of its own: piv is stable below)
    mma8_sub<T>(bufA, diag ? bufA : bufB, acc);
    __syncthreads();
  }
  if (!f2) DAG_TR(1);
  if (f1 || f2) {
    T* park = f1 ? HP + 2 * (c + 1) * SLOT : HP + (2 * c + 1) * SLOT;
    acc8_foreach<T>(acc, [&](int r, int cc, T& val) {
      __hip_atomic_store(park + r * TILE + cc, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    });
    dag_signal((f1 ? pre1 + (c + 1) * DAG_FS : pre2 + c * DAG_FS), epoch);
    if (R < 4 && c < 4) PRO_TS(64 + 8 * (4 * (int)R + (int)c) + 5);
    if (f2) {
      DAG_TR(6);
    } else {
      DAG_TR(3);
    }
    return;
  }
  acc8_foreach<T>(acc, [&](int r, int cc, T& val) { bufA[r * LDP + cc] = val; });
  if (!FUSED && diag) {  // (FUSED: the chain factors every diagonal tile; tile (c, c) left above as its feeder)
    __syncthreads();
    factor_diag_tile_2lvl<T>(bufA, bufB, sc, piv, info, c0, nvalid);
    DAG_TR(2);
    for (int e = tid; e < TILE * TILE; e += CHOL_THREADS) {
      const int r = e >> 6, cc = e & 63;
      __hip_atomic_store(HX + c * SLOT + e, bufB[r * LDP + cc], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      X[(c0 + r) * ldx + c0 + cc] = bufB[r * LDP + cc];
    }
    dag_signal(xready + c * DAG_FS, epoch);
    for (int e = tid; e < TILE * TILE; e += CHOL_THREADS) Dg[c * TILE * TILE + e] = bufA[(e >> 6) * LDP + (e & 63)];
    DAG_TR(3);
    return;
  }
  if (!dag_wait(xready + c * DAG_FS, nullptr, epoch, abortf, info, &wait_ok)) return;
  DAG_TR(2);
  load_tile_lds_hv<T>(HX + c * SLOT, bufB);
  __syncthreads();
  DAG_TR(6);
  Acc8<T> out;
  out.zero();
  mma8<T>(bufA, bufB, out);
  {
    T* hs = HL + (R * nt + c) * SLOT;
    acc8_foreach<T>(out, [&](int r, int cc, T& val) {
      __hip_atomic_store(hs + r * TILE + cc, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    });
  }
  if (idr) {  // (L^-T)(i, c) = X(c, i)': the tile of X proper, for the kernels that follow
    const int64_t i0 = (R - nt - ne) * TILE;
    acc8_foreach<T>(out, [&](int r, int cc, T& val) { X[(c0 + cc) * ldx + i0 + r] = val; });
  } else if (ext || store_l) {  // W (always) / the factor (when wanted) in its real home, for the kernels that follow
    acc8_foreach<T>(out, [&](int r, int cc, T& val) { rowp[r * ldr + c0 + cc] = val; });
  }
  DAG_TR(7);
  dag_signal(ready + (R * nt + c) * DAG_FS, epoch);
  DAG_TR(3);
  if (epi_row) {
    // own tile W(R, nt-1) (still in the accumulators) through LDS, the last piece of v from the tile of the [eta1' ; 0] row in
    // this block column (solved by its own workgroup at about the same time: its slot validates itself), then the 64 rows
    __syncthreads();
    acc8_foreach<T>(out, [&](int r, int cc, T& val) { bufA[r * LDP + cc] = val; });
    __syncthreads();
    epi_acc(bufA, HL + ((nt + ne - 1) * nt + c) * SLOT);
    const int64_t i = (R - nt) * TILE + er;
    T sk = epi_sk;
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) {
      epi_ss += __shfl_xor(epi_ss, o);
      epi_dt += __shfl_xor(epi_dt, o);
      sk += __shfl_xor(sk, o);
    }
    if ((tid & 7) == 0 && i < epi.B) {
      const T kd = epi.kdiag < T(0) ? epi.kd_ptr[0] : epi.kdiag;
      rowstats_finish<T>(i, epi_ss, epi_dt, sk, kd, epi.use_kt, epi.jitter, epi.rho, epi.lp, epi.y, epi.idx, epi.Kt, epi.muf,
                         epi.varf, epi.cb, epi.theta, epi.r, epi.w, epi.flags, epi.lam, epi.gamma, &epi_y);
    }
  }
#undef DAG_TR
#undef DAG_TRC
#undef PRO_TS
}

// fills the hand-over area with the sentinel (normally rider workgroups of the next k_syrk_tn launch do this; this kernel is the
// fallback for callers that factor again before such a launch comes by, and the initial fill)
// blockIdx.y = problem: the first n elements of each problem's region (stride elements apart)
template <typename T>
__global__ void k_fill_sent(T* __restrict__ p, int64_t n, int64_t stride) {
  const T sv = __builtin_bit_cast(T, Sent<T>::bits);
  T* q = p + blockIdx.y * stride;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) q[i] = sv;
}

// ---------------------------------------------------------------------------------------------------
// X = L^-1 from the diagonal inverses by recursive doubling (replaces the chain of nt-1 dependent row launches):
//   level with block size bs tiles: for every pair (left block [b, b+bs), right block [b+bs, b+bs+nr)), nr <= bs:
//       T   = L21 X11        (phase 0, into scratch at the coordinates of X21)
//       X21 = -X22 T         (phase 1)
// grid (bs, bs, pairs) of 64x64 tiles, 256 threads; log2(nt) levels x 2 launches, each a batch of independent tile GEMMs.
// ---------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(NTHREADS) void k_trtri_level(const T* __restrict__ A, int64_t ld, T* __restrict__ X, int64_t ldx,
                                                          T* __restrict__ S, int64_t lds, int64_t nt, int64_t bs,
                                                          int phase) {
  __shared__ __attribute__((aligned(16))) T smem[smem_elems<T>()];
  const int64_t base = (int64_t)blockIdx.z * 2 * bs, r0 = base + bs, c0 = base;
  if (r0 >= nt) return;
  const int64_t nr = (nt - r0) < bs ? (nt - r0) : bs;
  const int64_t bx = blockIdx.x, by = blockIdx.y;
  if (by >= nr) return;
  Acc<T> acc;
  acc.zero();
  if (phase == 0) {
    // T(by, bx) = sum_k L[r0+by][c0+k] X[c0+k][c0+bx] ; X11 is lower triangular: k >= bx
    gemm_tile<T, KC, RC>(A + (r0 + by) * TILE * ld + c0 * TILE, ld, X + c0 * TILE * ldx + (c0 + bx) * TILE, ldx, bx * TILE,
                         bs * TILE, nullptr, acc, smem);
    acc_foreach<T>(acc, [&](int r, int c, T val) { S[((r0 + by) * TILE + r) * lds + (c0 + bx) * TILE + c] = val; });
  } else {
    // X21(by, bx) = -sum_k X[r0+by][r0+k] T[r0+k][c0+bx] ; X22 is lower triangular: k <= by
    gemm_tile<T, KC, RC>(X + (r0 + by) * TILE * ldx + r0 * TILE, ldx, S + r0 * TILE * lds + (c0 + bx) * TILE, lds, 0,
                         (by + 1) * TILE, nullptr, acc, smem);
    acc_foreach<T>(acc, [&](int r, int c, T val) { X[((r0 + by) * TILE + r) * ldx + (c0 + bx) * TILE + c] = -val; });
  }
}

// copy the diagonal factors from Dg into the diagonal tiles of an n x n matrix (state export / building blocks)
template <typename T>
__global__ void k_publish_diag(T* __restrict__ A, int64_t ld, const T* __restrict__ Dg) {
  const int64_t k = blockIdx.x;
  for (int e = threadIdx.x; e < TILE * TILE; e += blockDim.x)
    A[(k * TILE + (e >> 6)) * ld + k * TILE + (e & 63)] = Dg[k * TILE * TILE + e];
}

// micro-benchmark of the diagonal-tile factorisation alone (tools/bench_diag.py): `reps` factorizations per launch
template <typename T, int VAR>
__global__ __launch_bounds__(CHOL_THREADS) void k_diag_bench(const T* __restrict__ A, T* __restrict__ out, int reps,
                                                             int32_t* info) {
  __shared__ __attribute__((aligned(16))) T sm[2 * TILE * LDP];
  __shared__ __attribute__((aligned(16))) T sc[SC_ELEMS];
  __shared__ T piv[TILE];
  T* bufA = sm;
  T* bufB = sm + TILE * LDP;
  for (int it = 0; it < reps; ++it) {
    for (int e = threadIdx.x; e < TILE * TILE; e += CHOL_THREADS) bufA[(e >> 6) * LDP + (e & 63)] = A[e];
    __syncthreads();
    NoHook nh;
    if (VAR == 0) factor_diag_tile512<T>(bufA, bufB, sc, piv, info, 0, 64);
    else if (VAR == 1) factor_diag_tile_2lvl<T, NoHook, 0>(bufA, bufB, sc, piv, info, 0, 64, nh);
    else if (VAR == 2) factor_diag_tile_2lvl<T, NoHook, 1>(bufA, bufB, sc, piv, info, 0, 64, nh);
    else if (VAR == 3) factor_diag_tile_2lvl<T, NoHook, 2>(bufA, bufB, sc, piv, info, 0, 64, nh);
    else if (VAR == 4) factor_diag_tile_2lvl<T, NoHook, 3>(bufA, bufB, sc, piv, info, 0, 64, nh);
    else if (VAR == 5) factor_diag_tile_2lvl<T, NoHook, 4>(bufA, bufB, sc, piv, info, 0, 64, nh);
    else if (VAR == 6) factor_diag_tile_2lvl<T, NoHook, 5>(bufA, bufB, sc, piv, info, 0, 64, nh);
    else if (VAR == 8) factor_diag_tile_panel<T, NoHook>(bufA, bufB, sc, piv, info, 0, 64, nh);
    else if (VAR >= 9 && VAR <= 12) factor_diag_tile_panel<T, NoHook, VAR - 8>(bufA, bufB, sc, piv, info, 0, 64, nh);
    else {  // VAR 7: the harness alone (tile reload + barrier)
    }
  }
  for (int e = threadIdx.x; e < TILE * TILE; e += CHOL_THREADS) {
    out[blockIdx.x * 2 * TILE * TILE + e] = bufA[(e >> 6) * LDP + (e & 63)];
    out[blockIdx.x * 2 * TILE * TILE + TILE * TILE + e] = bufB[(e >> 6) * LDP + (e & 63)];
  }
}

}  // namespace agp
