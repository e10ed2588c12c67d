// agp_device.h -- CDNA4 (gfx950) device building blocks: f64/f32 MFMA 64x64 tile product with
// LDS-staged, register-prefetched operands.  Written for wave64 / v_mfma_{f64,f32}_16x16x4 only.
//
// Every dense contraction of the CAVI path (kappa = Knm K^-1, W = kappa L_A^-T, kappa' diag(w) kappa,
// the Cholesky trailing updates, the triangular-inverse products, predict variances) is built from
// gemm_tile<> below; kernels differ only in tile mapping, k-range and epilogue.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace agp {

typedef double d4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef double d2v __attribute__((ext_vector_type(2)));

// ---- MFMA traits -----------------------------------------------------------------------------------
// A operand: lane l holds A[i = l&15][k = l>>4] ; B operand: lane l holds B[k = l>>4][j = l&15].
// C/D: col = l&15 ; row = (l>>4) + 4r for f64 but (l>>4)*4 + r for f32 (the f64 form has its own map).
template <typename T>
struct Mfma;
template <>
struct Mfma<double> {
  typedef d4 acc_t;
  typedef d2v vec_t;  // 16-byte global vector
  static constexpr int VEC = 2;
  static __device__ __forceinline__ acc_t mma(double a, double b, acc_t c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ int row(int lane, int r) { return (lane >> 4) + 4 * r; }
};
template <>
struct Mfma<float> {
  typedef f4 acc_t;
  typedef f4 vec_t;
  static constexpr int VEC = 4;
  static __device__ __forceinline__ acc_t mma(float a, float b, acc_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ int row(int lane, int r) { return (lane >> 4) * 4 + r; }
};

constexpr int TILE = 64;       // block tile edge (rows and cols of C per workgroup)
constexpr int BK = 16;         // k-depth staged per LDS buffer (f64; and the unit host-side k ranges are checked against)
constexpr int NTHREADS = 256;  // 4 waves as 2 (M) x 2 (N), each wave owns a 32x32 sub-tile = 2x2 MFMA tiles
constexpr int LDR = TILE + 16; // RC layout row stride: rows k,k+1 land on disjoint bank halves
// The staged k-depth is a compile-time property of the element type (round 4 experiment).  An fp32 v_mfma_16x16x4 takes half the
// cycles of the fp64 one, so at the same depth the fp32 kernels spend twice the share of their time between barriers
// (k_syrk_tn<float> runs at 0.46 of the sustained fp32 MFMA rate, the fp64 instantiations at 0.95).  Measured with 32 k per fp32 slab
// (-DAGP_BK_F32=32: as many MFMA cycles and as many LDS bytes per slab as fp64 at 16): C3's symmetric product 138 -> 166 us, step
// 0.672 -> 0.719 ms -- the 80 KB of a two-k-group workgroup allow two workgroups per CU instead of three, which costs more than the
// longer slabs return.  The default stays 16 for both types.
#ifndef AGP_BK_F32
#define AGP_BK_F32 16
#endif
template <typename T>
struct BkOf {
  static constexpr int v = sizeof(T) == 4 ? AGP_BK_F32 : BK;
};
template <int BKK>
struct Slab {
  static constexpr int LDK = BKK + 2;  // KC layout row stride: 16 rows x {k,k+1} hit distinct bank slots
  static constexpr int OPER = (TILE * LDK > BKK * LDR) ? TILE * LDK : BKK * LDR;  // 1280 at BKK = 16, 2560 at 32
  static constexpr int SMEM = 2 * 2 * OPER;  // double-buffered A and B tiles
};
constexpr int LDK = Slab<BK>::LDK;
constexpr int OPER_ELEMS = Slab<BK>::OPER;
constexpr int SMEM_ELEMS = Slab<BK>::SMEM;
// elements of the staging area of one k-group for element type T
template <typename T>
constexpr int smem_elems() {
  return Slab<BkOf<T>::v>::SMEM;
}

// Operand memory layouts.  "row" is the operand's C-side index (i for A, j for B).
struct KC {};  // element (row, k) at P[row*ld + k]   (k contiguous)   e.g. A[i][k], B^T given as B[j][k]
struct RC {};  // element (row, k) at P[k*ld + row]   (row contiguous) e.g. A^T given as A[k][i], B[k][j]

template <typename T, typename L, int BKK = BkOf<T>::v>
struct TileIO;

template <typename T, int BKK>
struct TileIO<T, KC, BKK> {
  static constexpr int VEC = Mfma<T>::VEC;
  static constexpr int LDK = Slab<BKK>::LDK;
  static constexpr int NV = BKK / VEC;                  // vectors per row
  static constexpr int VPT = TILE * NV / NTHREADS;      // vectors per thread (2 f64, 1 f32)
  typedef typename Mfma<T>::vec_t vec_t;
  struct Regs {
    vec_t v[VPT];
  };
  // P points at (row0, 0) of the operand; k0 is the k offset of this tile
  static __device__ __forceinline__ void load(Regs& r, const T* __restrict__ P, int64_t ld, int64_t k0, int tid) {
#pragma unroll
    for (int v = 0; v < VPT; ++v) {
      int vi = tid + v * NTHREADS;
      int row = vi / NV, kv = vi % NV;
      r.v[v] = *reinterpret_cast<const vec_t*>(P + (int64_t)row * ld + k0 + kv * VEC);
    }
  }
  static __device__ __forceinline__ void store(const Regs& r, T* S, int tid, const T* wscale, int64_t k0) {
    (void)wscale;
    (void)k0;
#pragma unroll
    for (int v = 0; v < VPT; ++v) {
      int vi = tid + v * NTHREADS;
      int row = vi / NV, kv = vi % NV;
#pragma unroll
      for (int e = 0; e < VEC; ++e) S[row * LDK + kv * VEC + e] = r.v[v][e];
    }
  }
  static __device__ __forceinline__ T frag(const T* S, int rbase, int kk, int lane) {
    return S[(rbase + (lane & 15)) * LDK + kk * 4 + (lane >> 4)];
  }
};

template <typename T, int BKK>
struct TileIO<T, RC, BKK> {
  static constexpr int VEC = Mfma<T>::VEC;
  static constexpr int NV = TILE / VEC;                 // vectors per k-row
  static constexpr int VPT = BKK * NV / NTHREADS;       // 2 f64, 1 f32 (2 at BKK = 32)
  typedef typename Mfma<T>::vec_t vec_t;
  struct Regs {
    vec_t v[VPT];
  };
  // P points at (k = 0, row0)
  static __device__ __forceinline__ void load(Regs& r, const T* __restrict__ P, int64_t ld, int64_t k0, int tid) {
#pragma unroll
    for (int v = 0; v < VPT; ++v) {
      int vi = tid + v * NTHREADS;
      int krow = vi / NV, rv = vi % NV;
      r.v[v] = *reinterpret_cast<const vec_t*>(P + (k0 + krow) * ld + rv * VEC);
    }
  }
  // wscale != nullptr : multiply row k of the tile by wscale[k] (diag(w) folded into the operand)
  static __device__ __forceinline__ void store(const Regs& r, T* S, int tid, const T* __restrict__ wscale,
                                               int64_t k0) {
#pragma unroll
    for (int v = 0; v < VPT; ++v) {
      int vi = tid + v * NTHREADS;
      int krow = vi / NV, rv = vi % NV;
      T w = wscale ? wscale[k0 + krow] : T(1);
#pragma unroll
      for (int e = 0; e < VEC; ++e) S[krow * LDR + rv * VEC + e] = wscale ? r.v[v][e] * w : r.v[v][e];
    }
  }
  static __device__ __forceinline__ T frag(const T* S, int rbase, int kk, int lane) {
    return S[(kk * 4 + (lane >> 4)) * LDR + rbase + (lane & 15)];
  }
};

template <typename T>
struct Acc {
  typename Mfma<T>::acc_t a[2][2];
  __device__ __forceinline__ void zero() {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) a[i][j][r] = T(0);
  }
};

// MFMA over one staged BK slab
template <typename T, typename LA, typename LB>
__device__ __forceinline__ void mma_slab(const T* As, const T* Bs, Acc<T>& acc, int wm, int wn, int lane) {
  constexpr int BKK = BkOf<T>::v;
#pragma unroll
  for (int kk = 0; kk < BKK / 4; ++kk) {
    T a0 = TileIO<T, LA>::frag(As, wm * 32, kk, lane);
    T a1 = TileIO<T, LA>::frag(As, wm * 32 + 16, kk, lane);
    T b0 = TileIO<T, LB>::frag(Bs, wn * 32, kk, lane);
    T b1 = TileIO<T, LB>::frag(Bs, wn * 32 + 16, kk, lane);
    acc.a[0][0] = Mfma<T>::mma(a0, b0, acc.a[0][0]);
    acc.a[0][1] = Mfma<T>::mma(a0, b1, acc.a[0][1]);
    acc.a[1][0] = Mfma<T>::mma(a1, b0, acc.a[1][0]);
    acc.a[1][1] = Mfma<T>::mma(a1, b1, acc.a[1][1]);
  }
}

// C_tile(64x64) += sum_{k in [kBegin,kEnd)} A(row,k) * B(col,k) ; kBegin/kEnd multiples of BK.
// A, B already point at their 64-row origin (see TileIO::load).  wscaleA folds diag(w) into A (RC only).
// KG = 1: 256 threads.  KG = 2: 512 threads = two k-groups of 4 waves; group g takes slabs g, g+2, ... into its own
// LDS staging area and group 1's accumulators are added to group 0's through LDS at the end (deterministic) -- used when
// a problem has fewer 64x64 tiles than the chip has CUs, to put two waves on every SIMD.
// smem: KG*SMEM_ELEMS elements.  All threads must call; on return only group 0 (threadIdx.x < 256) holds the result.
template <typename T, typename LA, typename LB, int KG = 1>
__device__ __forceinline__ void gemm_tile(const T* __restrict__ A, int64_t lda, const T* __restrict__ B, int64_t ldb,
                                          int64_t kBegin, int64_t kEnd, const T* __restrict__ wscaleA, Acc<T>& acc,
                                          T* smem_all) {
  constexpr int BK = BkOf<T>::v;  // (shadows the f64 constant: everything below is in units of this type's slab depth)
  constexpr int OPER_ELEMS = Slab<BK>::OPER, SMEM_ELEMS = Slab<BK>::SMEM;
  const int tid = threadIdx.x & (NTHREADS - 1);
  const int grp = (KG > 1) ? (threadIdx.x >> 8) : 0;
  const int lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
  T* smem = smem_all + grp * SMEM_ELEMS;
  typename TileIO<T, LA>::Regs ra;
  typename TileIO<T, LB>::Regs rb;
  if (kBegin >= kEnd) return;
  const int64_t nslab = (kEnd - kBegin) / BK;
  const int64_t niter = (nslab + KG - 1) / KG;
  auto slab_k = [&](int64_t it) { return kBegin + (it * KG + grp) * BK; };
  bool have = grp < nslab;
  if (have) {
    TileIO<T, LA>::load(ra, A, lda, slab_k(0), tid);
    TileIO<T, LB>::load(rb, B, ldb, slab_k(0), tid);
    TileIO<T, LA>::store(ra, smem, tid, wscaleA, slab_k(0));
    TileIO<T, LB>::store(rb, smem + OPER_ELEMS, tid, nullptr, slab_k(0));
  }
  __syncthreads();
  int cur = 0;
  for (int64_t it = 0; it < niter; ++it) {
    const bool more = ((it + 1) * KG + grp) < nslab;
    if (more) {
      TileIO<T, LA>::load(ra, A, lda, slab_k(it + 1), tid);
      TileIO<T, LB>::load(rb, B, ldb, slab_k(it + 1), tid);
    }
    if (have) {
      const T* As = smem + cur * 2 * OPER_ELEMS;
      mma_slab<T, LA, LB>(As, As + OPER_ELEMS, acc, wm, wn, lane);
    }
    if (more) {
      T* Ns = smem + (cur ^ 1) * 2 * OPER_ELEMS;
      TileIO<T, LA>::store(ra, Ns, tid, wscaleA, slab_k(it + 1));
      TileIO<T, LB>::store(rb, Ns + OPER_ELEMS, tid, nullptr, slab_k(it + 1));
    }
    have = more;
    __syncthreads();
    cur ^= 1;
  }
  if (KG > 1) {
    // groups 1 .. KG-1 -> LDS -> group 0, added in group order (64x64 values fit in one staging area; group g uses area g - 1,
    // all of them free after the last barrier of the loop)
    static_assert(SMEM_ELEMS >= TILE * TILE, "one staging area must hold a 64x64 partial result");
    if (grp >= 1) {
      T* red = smem_all + (grp - 1) * SMEM_ELEMS;
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
          for (int r = 0; r < 4; ++r) red[((mi * 2 + ni) * 4 + r) * NTHREADS + tid] = acc.a[mi][ni][r];
    }
    __syncthreads();
    if (grp == 0) {
#pragma unroll
      for (int g = 1; g < KG; ++g) {
        const T* red = smem_all + (g - 1) * SMEM_ELEMS;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc.a[mi][ni][r] += red[((mi * 2 + ni) * 4 + r) * NTHREADS + tid];
      }
    }
  }
}

// visit every accumulator element of this thread: f(row_in_tile, col_in_tile, value)
template <typename T, typename F>
__device__ __forceinline__ void acc_foreach(Acc<T>& acc, F f) {
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        f(wm * 32 + mi * 16 + Mfma<T>::row(lane, r), wn * 32 + ni * 16 + (lane & 15), acc.a[mi][ni][r]);
}

// sum over the 16 lanes that share (lane>>4) -- i.e. over the 16 columns of one MFMA tile row
template <typename T>
__device__ __forceinline__ T row16_sum(T v) {
  v += __shfl_xor(v, 1);
  v += __shfl_xor(v, 2);
  v += __shfl_xor(v, 4);
  v += __shfl_xor(v, 8);
  return v;
}

// Row reductions of per-element values g0/g1 over this wave's 32 columns; lane (l&15)==0 stores
// part0/part1[row] (caller offsets the pointers to its partial slice: one slice per (tile column, wn)).
template <typename T, typename G>
__device__ __forceinline__ void acc_row_reduce(Acc<T>& acc, G g, T* part0, T* part1, int64_t row0) {
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      int row = wm * 32 + mi * 16 + Mfma<T>::row(lane, r);
      T s0 = T(0), s1 = T(0);
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        int col = wn * 32 + ni * 16 + (lane & 15);
        g(row, col, acc.a[mi][ni][r], s0, s1);
      }
      s0 = row16_sum(s0);
      s1 = row16_sum(s1);
      if ((lane & 15) == 0) {
        if (part0) part0[row0 + row] = s0;
        if (part1) part1[row0 + row] = s1;
      }
    }
}

template <typename T>
__device__ __forceinline__ T block_sum(T v, T* red /* >= 4 elems */) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
  int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  T s = T(0);
  int nw = (blockDim.x + 63) >> 6;
  for (int w = 0; w < nw; ++w) s += red[w];
  return s;
}

}  // namespace agp
