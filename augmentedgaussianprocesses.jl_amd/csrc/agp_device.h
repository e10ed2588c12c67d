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

// ---- fp64: the 4-block v_mfma_f64_4x4x4 instead of v_mfma_f64_16x16x4 (round 6) ------------------------------------------------------
// Measured on this chip (tools/mfma_ceiling.py, tools/ubench/mfma4.hip; profiles/r06_mfma_ceiling.txt): the register-only issue loop of
// v_mfma_f64_16x16x4 tops out at 47-48 TF -- 0.61 of the 78.6 TF datasheet peak, with zero operands at 585 W exactly as with random
// ones at 911 W, so it is the instruction's rate (~105 cycles), not a power limit -- while v_mfma_f64_4x4x4 (four independent 4x4x4
// block products per instruction, 512 flop) sustains 74 TF = 0.94 of peak, already with one wave per SIMD.  Its operand registers
// have the SAME lane map as the 16x16x4 form -- lane l: A[row l & 15][k = l >> 4], B[k = l >> 4][col l & 15] -- but it multiplies
// only the four DIAGONAL 4x4 block pairs: D_b = A[rows 4b..4b+3] B[cols 4b..4b+3], b = (l >> 2) & 3 (the A-broadcast modifiers
// CBSZ / ABID have no effect on the fp64 form: decoded lane by lane on the device).  A 16 x 16 x 4 block product is therefore FOUR
// instructions whose B operand is rotated by s = 0..3 blocks inside every row of 16 lanes (two v_mov_b32 with DPP row_ror each: the
// VALU is idle in these loops), instruction s covering the block pairs (b, (b + s) & 3).  Result lane d of instruction s holds
// C[4 b + (d >> 4)][4 ((b + s) & 3) + (d & 3)], b = (d >> 2) & 3: a lane's four values (s = 0..3) sit in ONE row.
// The accumulators of a tile product keep this layout through the k loop; gemm_tile converts them to the canonical 16x16x4 layout
// (the one every epilogue is written for) through LDS once per tile -- 32 KB written and read, ~1 us against >= 8 us of k loop.
// RESULT (GPU session 5, profiles/r06_mfma_4x4x4_in_kernels.txt): built for gemm_tile, gemm_tile_tall and k_kernelmatrix_mma, the whole
// GPU suite green on it -- and NO faster: C5 13.86 ms (16x16x4: 13.5-13.7), C4 1.15 (1.09-1.12), hyper iteration 1.019 (0.977).  The
// production kernels are not bound by the whole-chip issue rate of the 16x16x4 instruction as the register-only loop shows it: C4 / C5
// run at the socket power limit (1.14-1.17 kW, engine clock pulled to 2.32 GHz, bench.py step_clock) with either instruction, and the
// 4x4x4 form fetches four times the operand registers per flop, rotates B with DPP moves and pays the way back to the canonical
// layout.  Kept behind AGP_F64_MFMA4 (compile-time, default 0); the default build uses v_mfma_f64_16x16x4.
#ifndef AGP_F64_MFMA4
#define AGP_F64_MFMA4 0
#endif
template <typename T>
constexpr bool use_mfma4() {
  return sizeof(T) == 8 && AGP_F64_MFMA4 != 0;
}
template <int N>
__device__ __forceinline__ double dpp_ror(double v) {  // out[l] = in[(l & 48) | ((l - N) & 15)]
  const long long u = __builtin_bit_cast(long long, v);
  int lo = (int)u, hi = (int)(u >> 32);
  lo = __builtin_amdgcn_mov_dpp(lo, 0x120 + N, 0xf, 0xf, false);
  hi = __builtin_amdgcn_mov_dpp(hi, 0x120 + N, 0xf, 0xf, false);
  return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned long long)(unsigned)lo);
}
// c[s] += A_blocks * rot_s(B)_blocks for s = 0..3 (b1, b2, b3: B rotated so that lane l holds the value of lane (l + 4 s) of its row)
__device__ __forceinline__ d4 mma4x4(double a, double b0, double b1, double b2, double b3, d4 c) {
  c[0] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b0, c[0], 0, 0, 0);
  c[1] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b1, c[1], 0, 0, 0);
  c[2] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b2, c[2], 0, 0, 0);
  c[3] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b3, c[3], 0, 0, 0);
  return c;
}
// position of element s of a lane's accumulator inside its 16 x 16 block
__device__ __forceinline__ int g4_row(int lane) { return 4 * ((lane >> 2) & 3) + (lane >> 4); }
__device__ __forceinline__ int g4_col(int lane, int s) { return 4 * ((((lane >> 2) & 3) + s) & 3) + (lane & 3); }

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
    if constexpr (use_mfma4<T>()) {
      const double b01 = dpp_ror<12>(b0), b02 = dpp_ror<8>(b0), b03 = dpp_ror<4>(b0);
      const double b11 = dpp_ror<12>(b1), b12 = dpp_ror<8>(b1), b13 = dpp_ror<4>(b1);
      acc.a[0][0] = mma4x4(a0, b0, b01, b02, b03, acc.a[0][0]);
      acc.a[0][1] = mma4x4(a0, b1, b11, b12, b13, acc.a[0][1]);
      acc.a[1][0] = mma4x4(a1, b0, b01, b02, b03, acc.a[1][0]);
      acc.a[1][1] = mma4x4(a1, b1, b11, b12, b13, acc.a[1][1]);
    } else {
      acc.a[0][0] = Mfma<T>::mma(a0, b0, acc.a[0][0]);
      acc.a[0][1] = Mfma<T>::mma(a0, b1, acc.a[0][1]);
      acc.a[1][0] = Mfma<T>::mma(a1, b0, acc.a[1][0]);
      acc.a[1][1] = Mfma<T>::mma(a1, b1, acc.a[1][1]);
    }
  }
}
// fp64: the accumulators of a 64 x 64 tile from the 4x4x4 layout (above) to the canonical one (Mfma<double>::row / lane & 15) through
// LDS: tile[64][LDR] (LDR = 80: rows one apart are half the banks apart -- both the scattered writes and the row reads are
// conflict-free).  Every thread of the workgroup calls (two barriers); `mine`: this thread holds accumulators (k-group 0).
template <typename T>
__device__ __forceinline__ void acc_canonical(Acc<T>& acc, T* tile, bool mine) {
  if constexpr (use_mfma4<T>()) {
    static_assert(TILE * LDR <= Slab<BkOf<T>::v>::SMEM, "the conversion tile must fit into one staging area");
    const int tid = threadIdx.x & (NTHREADS - 1), lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    if (mine) {
      const int r4 = g4_row(lane);
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
          for (int s2 = 0; s2 < 4; ++s2)
            tile[(wm * 32 + mi * 16 + r4) * LDR + wn * 32 + ni * 16 + g4_col(lane, s2)] = acc.a[mi][ni][s2];
    }
    __syncthreads();
    if (mine) {
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            acc.a[mi][ni][r] = tile[(wm * 32 + mi * 16 + Mfma<T>::row(lane, r)) * LDR + wn * 32 + ni * 16 + (lane & 15)];
    }
    __syncthreads();  // (callers reuse the staging area)
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
  if constexpr (use_mfma4<T>()) {
    if (KG > 1) __syncthreads();  // (the partial sums have been read: the staging area is free again)
    acc_canonical<T>(acc, smem_all, grp == 0);
  }
}

// ---- 128 x 64 C tile (round 6) ------------------------------------------------------------------------------
// The 64 x 64 product above gives a wave a 32 x 32 sub-tile: per k-step of 4 it reads 2 + 2 operand fragments for 4 MFMAs, and a
// 16-deep slab is 16 MFMAs (1024 MFMA cycles in fp64) between two barriers.  Here the workgroup (still 256 threads, 2 x 2 waves)
// owns 128 rows x 64 columns and a wave 64 x 32 = 4 x 2 MFMA tiles: 4 + 2 fragment reads for 8 MFMAs (0.75 LDS reads per MFMA
// instead of 1), 32 MFMAs between barriers, and the B operand is staged once for twice the rows (global -> LDS traffic per flop
// 0.75 of the square tile's).  Both operands k-contiguous (KC) only: the "NT" products kappa = K_nm K^-1, W = kappa Xa', the
// hyper-gradient's products.  smem: TALL_SMEM elements.  fp64 accumulators: 8 x 4 doubles = 64 VGPRs.
template <typename T>
struct AccTall {
  typename Mfma<T>::acc_t a[4][2];
  __device__ __forceinline__ void zero() {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) a[i][j][r] = T(0);
  }
};
template <typename T>
struct TallSlab {
  static constexpr int BKK = BkOf<T>::v;
  static constexpr int LDK = Slab<BKK>::LDK;
  static constexpr int A_ELEMS = 2 * TILE * LDK, B_ELEMS = TILE * LDK;
  static constexpr int SMEM = 2 * (A_ELEMS + B_ELEMS);  // double-buffered
};
template <typename T>
__device__ __forceinline__ void gemm_tile_tall(const T* __restrict__ A, int64_t lda, const T* __restrict__ B, int64_t ldb,
                                               int64_t kBegin, int64_t kEnd, AccTall<T>& acc, T* smem) {
  constexpr int BK = BkOf<T>::v;
  constexpr int LDKK = TallSlab<T>::LDK, AE = TallSlab<T>::A_ELEMS, BUF = TallSlab<T>::A_ELEMS + TallSlab<T>::B_ELEMS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
  typedef TileIO<T, KC> IO;
  typename IO::Regs ra0, ra1, rb;
  if (kBegin >= kEnd) return;
  const int64_t nslab = (kEnd - kBegin) / BK;
  const T* A1 = A + (int64_t)TILE * lda;
  IO::load(ra0, A, lda, kBegin, tid);
  IO::load(ra1, A1, lda, kBegin, tid);
  IO::load(rb, B, ldb, kBegin, tid);
  IO::store(ra0, smem, tid, nullptr, 0);
  IO::store(ra1, smem + TILE * LDKK, tid, nullptr, 0);
  IO::store(rb, smem + AE, tid, nullptr, 0);
  __syncthreads();
  int cur = 0;
  for (int64_t it = 0; it < nslab; ++it) {
    const bool more = it + 1 < nslab;
    if (more) {
      const int64_t k1 = kBegin + (it + 1) * BK;
      IO::load(ra0, A, lda, k1, tid);
      IO::load(ra1, A1, lda, k1, tid);
      IO::load(rb, B, ldb, k1, tid);
    }
    {
      const T* As = smem + cur * BUF;
      const T* Bs = As + AE;
#pragma unroll
      for (int kk = 0; kk < BK / 4; ++kk) {
        T a[4], b[2];
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = IO::frag(As, wm * 64 + i * 16, kk, lane);
#pragma unroll
        for (int j = 0; j < 2; ++j) b[j] = IO::frag(Bs, wn * 32 + j * 16, kk, lane);
        if constexpr (use_mfma4<T>()) {  // 4x4x4 form, see mma_slab
          const double b01 = dpp_ror<12>(b[0]), b02 = dpp_ror<8>(b[0]), b03 = dpp_ror<4>(b[0]);
          const double b11 = dpp_ror<12>(b[1]), b12 = dpp_ror<8>(b[1]), b13 = dpp_ror<4>(b[1]);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            acc.a[i][0] = mma4x4(a[i], b[0], b01, b02, b03, acc.a[i][0]);
            acc.a[i][1] = mma4x4(a[i], b[1], b11, b12, b13, acc.a[i][1]);
          }
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc.a[i][j] = Mfma<T>::mma(a[i], b[j], acc.a[i][j]);
        }
      }
    }
    if (more) {
      T* Ns = smem + (cur ^ 1) * BUF;
      IO::store(ra0, Ns, tid, nullptr, 0);
      IO::store(ra1, Ns + TILE * LDKK, tid, nullptr, 0);
      IO::store(rb, Ns + AE, tid, nullptr, 0);
    }
    __syncthreads();
    cur ^= 1;
  }
  if constexpr (use_mfma4<T>()) {  // to the canonical accumulator layout, 64 rows at a time (tile[64][LDR] in the staging area)
    static_assert(TILE * LDR <= TallSlab<T>::SMEM, "conversion tile");
    const int r4 = g4_row(lane);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      if (wm == h) {
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
          for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2) smem[(mi * 16 + r4) * LDR + wn * 32 + ni * 16 + g4_col(lane, s2)] = acc.a[mi][ni][s2];
      }
      __syncthreads();
      if (wm == h) {
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
          for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 4; ++r)
              acc.a[mi][ni][r] = smem[(mi * 16 + Mfma<T>::row(lane, r)) * LDR + wn * 32 + ni * 16 + (lane & 15)];
      }
      __syncthreads();
    }
  }
}
template <typename T, typename F>
__device__ __forceinline__ void acc_foreach_tall(AccTall<T>& acc, F f) {
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        f(wm * 64 + mi * 16 + Mfma<T>::row(lane, r), wn * 32 + ni * 16 + (lane & 15), acc.a[mi][ni][r]);
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
