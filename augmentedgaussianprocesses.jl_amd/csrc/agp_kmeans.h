// agp_kmeans.h -- inducing-point selection on the device: the Lloyd iterations behind
// `inducingpoints(KmeansAlg(m), X)` (reference call sites test/testingtools.jl:66, docs/examples/gpclassification.jl:47;
// the algorithm lives in the unvendored InducingPoints.jl -> Clustering.kmeans!).
//
//   k_km_assign : nearest centre of every point.  One workgroup = 64 points x all centres; the point tile stays in LDS,
//                 centre tiles stream through; distances in GEMM form  ||c||^2 - 2 x.c  (+ ||x||^2 at the end) on
//                 v_mfma 16x16x4, running (min, argmin) in registers, ties to the smaller index.  The N x m distance
//                 matrix never exists.
//   k_km_bucket : stable bucketing of every 8192-point chunk by centre tile (index order kept inside a bucket)
//   k_km_sums   : centre sums as H' X (H = one-hot assignment) on MFMA with the one-hot operand generated from the labels
//                 on the fly, over the bucket of its centre tile only; k_km_finish reduces the chunk partials in chunk
//                 order, so the new centres are bit-reproducible (no floating-point atomics anywhere).
#pragma once
#include "agp_device.h"

namespace agp {

constexpr int KM_MAXD = 128;     // padded feature dimension limit of the LDS-resident point tile
constexpr int KM_CHUNK = 8192;   // points per partial-sum chunk

// cn[j] = ||c_j||^2 for j < m ; +huge for the padding rows so that they are never selected
template <typename T>
__global__ void k_km_cnorm(const T* __restrict__ C, int64_t ldc, int64_t m, int64_t mp, int64_t D, T* __restrict__ cn) {
  int64_t j = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (j >= mp) return;
  T s = T(0);
  if (j < m) {
    for (int64_t d = 0; d < D; ++d) s += C[j * ldc + d] * C[j * ldc + d];
  } else {
    s = sizeof(T) == 8 ? T(1e300) : T(1e30);
  }
  cn[j] = s;
}

template <typename T>
__global__ __launch_bounds__(NTHREADS) void k_km_assign(const T* __restrict__ X, int64_t ldx, int64_t N, int64_t D, int Dp,
                                                        const T* __restrict__ C, int64_t ldc, int64_t m, int64_t mp,
                                                        const T* __restrict__ cn, int32_t* __restrict__ labels,
                                                        T* __restrict__ mind) {
  extern __shared__ __attribute__((aligned(16))) unsigned char km_smem[];
  T* Xs = reinterpret_cast<T*>(km_smem);  // [64][LDX]
  const int LDX = Dp + 2;
  T* Cs = Xs + TILE * LDX;                // [64][LDX]
  T* xn = Cs + TILE * LDX;                // [64]
  T* cns = xn + TILE;                     // [64]
  T* redv = cns + TILE;                   // [2][64] cross-wave reduction
  int* redi = reinterpret_cast<int*>(redv + 2 * TILE);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
  const int64_t p0 = (int64_t)blockIdx.x * TILE;
  for (int e = tid; e < TILE * Dp; e += NTHREADS) {
    int r = e / Dp, d = e % Dp;
    int64_t p = p0 + r;
    Xs[r * LDX + d] = (p < N && d < D) ? X[p * ldx + d] : T(0);
  }
  __syncthreads();
  if (tid < TILE) {
    T s = T(0);
    for (int d = 0; d < Dp; ++d) s += Xs[tid * LDX + d] * Xs[tid * LDX + d];
    xn[tid] = s;
  }
  T best[2][4];
  int bidx[2][4];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      best[mi][r] = sizeof(T) == 8 ? T(1e308) : T(3e38);
      bidx[mi][r] = 0x7fffffff;
    }
  for (int64_t c0 = 0; c0 < mp; c0 += TILE) {
    __syncthreads();  // previous tile fully consumed
    for (int e = tid; e < TILE * Dp; e += NTHREADS) {
      int r = e / Dp, d = e % Dp;
      int64_t j = c0 + r;
      Cs[r * LDX + d] = (j < m && d < D) ? C[j * ldc + d] : T(0);
    }
    if (tid < TILE) cns[tid] = cn[c0 + tid];
    __syncthreads();
    typename Mfma<T>::acc_t acc[2][2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[mi][ni][r] = T(0);
    for (int kk = 0; kk < Dp / 4; ++kk) {
      T a0 = Xs[(wm * 32 + (lane & 15)) * LDX + kk * 4 + (lane >> 4)];
      T a1 = Xs[(wm * 32 + 16 + (lane & 15)) * LDX + kk * 4 + (lane >> 4)];
      T b0 = Cs[(wn * 32 + (lane & 15)) * LDX + kk * 4 + (lane >> 4)];
      T b1 = Cs[(wn * 32 + 16 + (lane & 15)) * LDX + kk * 4 + (lane >> 4)];
      acc[0][0] = Mfma<T>::mma(a0, b0, acc[0][0]);
      acc[0][1] = Mfma<T>::mma(a0, b1, acc[0][1]);
      acc[1][0] = Mfma<T>::mma(a1, b0, acc[1][0]);
      acc[1][1] = Mfma<T>::mma(a1, b1, acc[1][1]);
    }
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        const int cl = wn * 32 + ni * 16 + (lane & 15);
        const T cnv = cns[cl];
        const int cj = (int)c0 + cl;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          T d = cnv - T(2) * acc[mi][ni][r];
          if (d < best[mi][r] || (d == best[mi][r] && cj < bidx[mi][r])) {
            best[mi][r] = d;
            bidx[mi][r] = cj;
          }
        }
      }
  }
  // reduce over the 16 lanes that share a row, then over the two waves in the column direction
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      T v = best[mi][r];
      int ix = bidx[mi][r];
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) {
        T v2 = __shfl_xor(v, o);
        int i2 = __shfl_xor(ix, o);
        if (v2 < v || (v2 == v && i2 < ix)) {
          v = v2;
          ix = i2;
        }
      }
      if ((lane & 15) == 0) {
        const int row = wm * 32 + mi * 16 + Mfma<T>::row(lane, r);
        redv[wn * TILE + row] = v;
        redi[wn * TILE + row] = ix;
      }
    }
  __syncthreads();
  if (tid < TILE && p0 + tid < N) {
    T v = redv[tid];
    int ix = redi[tid];
    T v2 = redv[TILE + tid];
    int i2 = redi[TILE + tid];
    if (v2 < v || (v2 == v && i2 < ix)) {
      v = v2;
      ix = i2;
    }
    labels[p0 + tid] = ix;
    T dd = xn[tid] + v;
    mind[p0 + tid] = dd > T(0) ? dd : T(0);
  }
}

// Stable bucketing of one point chunk by centre tile (64 centres): list[chunk][.] holds the chunk's point indices grouped by
// tile, in index order inside every group (so the sums below add in a fixed order), off[chunk][t] the group starts.
// One workgroup (256 threads) per chunk; ntile = mp/64 <= KM_MAXTILES.
constexpr int KM_MAXTILES = 256;

__global__ __launch_bounds__(256) void k_km_bucket(const int32_t* __restrict__ labels, int64_t N, int ntile,
                                                   int32_t* __restrict__ list, int32_t* __restrict__ off) {
  __shared__ int cnt[KM_MAXTILES], base[KM_MAXTILES + 1], sub[KM_MAXTILES];
  __shared__ int tl[256];
  const int tid = threadIdx.x;
  const int64_t pbeg = (int64_t)blockIdx.x * KM_CHUNK;
  const int64_t pend = pbeg + KM_CHUNK < N ? pbeg + KM_CHUNK : N;
  for (int t = tid; t < ntile; t += 256) cnt[t] = 0;
  __syncthreads();
  for (int64_t p = pbeg + tid; p < pend; p += 256) atomicAdd(&cnt[labels[p] >> 6], 1);
  __syncthreads();
  if (tid == 0) {
    int a = 0;
    for (int t = 0; t < ntile; ++t) {
      base[t] = a;
      a += cnt[t];
    }
    base[ntile] = a;
  }
  __syncthreads();
  for (int t = tid; t <= ntile; t += 256) off[(int64_t)blockIdx.x * (KM_MAXTILES + 1) + t] = base[t];
  // placement, 256 points at a time in index order; rank among the same-tile points of the sub-block = stable
  for (int64_t p0 = pbeg; p0 < pend; p0 += 256) {
    const int64_t p = p0 + tid;
    const int t = p < pend ? (labels[p] >> 6) : -1;
    for (int q = tid; q < ntile; q += 256) sub[q] = 0;
    tl[tid] = t;
    __syncthreads();
    if (t >= 0) {
      int rank = 0;
      for (int q = 0; q < tid; ++q) rank += (tl[q] == t);
      list[(int64_t)blockIdx.x * KM_CHUNK + base[t] + rank] = (int32_t)(p - pbeg);
      atomicAdd(&sub[t], 1);
    }
    __syncthreads();
    for (int q = tid; q < ntile; q += 256) base[q] += sub[q];
    __syncthreads();
  }
}

// partial centre sums of one point chunk: part[chunk][mp][Dp + 16] ; column Dp holds the member count.
// H' X on MFMA with the one-hot operand generated from the labels on the fly, but only over the chunk's points that belong
// to this centre tile (the bucket list above) -- a point contributes to exactly one centre.
template <typename T>
__global__ __launch_bounds__(NTHREADS) void k_km_sums(const T* __restrict__ X, int64_t ldx, int64_t N, int64_t D, int Dp,
                                                      const int32_t* __restrict__ labels, int64_t mp,
                                                      const int32_t* __restrict__ list, const int32_t* __restrict__ off,
                                                      T* __restrict__ part) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lr = lane & 15, lk = lane >> 4;
  const int c0 = blockIdx.x * TILE + wave * 16;  // this wave's 16 centres
  const int64_t pbeg = (int64_t)blockIdx.y * KM_CHUNK;
  const int32_t* lst = list + (int64_t)blockIdx.y * KM_CHUNK;
  const int qbeg = off[(int64_t)blockIdx.y * (KM_MAXTILES + 1) + blockIdx.x];
  const int qend = off[(int64_t)blockIdx.y * (KM_MAXTILES + 1) + blockIdx.x + 1];
  const int NT = Dp / 16;
  typename Mfma<T>::acc_t acc[KM_MAXD / 16 + 1];
#pragma unroll
  for (int t = 0; t <= KM_MAXD / 16; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[t][r] = T(0);
  for (int q = qbeg; q < qend; q += 4) {
    const bool ok = q + lk < qend;
    const int64_t pk = ok ? pbeg + lst[q + lk] : 0;
    const int lab = ok ? labels[pk] : -1;
    const T a = (lab == c0 + lr) ? T(1) : T(0);
#pragma unroll
    for (int t = 0; t < KM_MAXD / 16; ++t) {
      if (t < NT) {
        const int col = t * 16 + lr;
        const T b = (ok && col < D) ? X[pk * ldx + col] : T(0);
        acc[t] = Mfma<T>::mma(a, b, acc[t]);
      }
    }
    acc[KM_MAXD / 16] = Mfma<T>::mma(a, T(1), acc[KM_MAXD / 16]);
  }
  const int64_t LDPt = Dp + 16;
  T* out = part + ((int64_t)blockIdx.y * mp) * LDPt;
#pragma unroll
  for (int t = 0; t < KM_MAXD / 16; ++t) {
    if (t < NT) {
#pragma unroll
      for (int r = 0; r < 4; ++r) out[(int64_t)(c0 + Mfma<T>::row(lane, r)) * LDPt + t * 16 + lr] = acc[t][r];
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) out[(int64_t)(c0 + Mfma<T>::row(lane, r)) * LDPt + Dp + lr] = acc[KM_MAXD / 16][r];
}

// new centre = (sum over chunks, in chunk order) / count ; a cluster that lost all its points keeps its centre
template <typename T>
__global__ void k_km_finish(const T* __restrict__ part, int nchunks, int64_t mp, int Dp, int64_t m, int64_t D,
                            T* __restrict__ C, int64_t ldc, int32_t* __restrict__ counts) {
  int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (e >= m * D) return;
  const int64_t j = e / D, d = e % D, LDPt = Dp + 16;
  T s = T(0), cnt = T(0);
  for (int c = 0; c < nchunks; ++c) {
    s += part[((int64_t)c * mp + j) * LDPt + d];
    cnt += part[((int64_t)c * mp + j) * LDPt + Dp];
  }
  if (cnt > T(0)) C[j * ldc + d] = s / cnt;
  if (d == 0 && counts) counts[j] = (int32_t)(cnt + T(0.5));
}

// deterministic sum of n values (two stages: per-block partials, then one block)
template <typename T>
__global__ void k_km_sum_partial(const T* __restrict__ v, int64_t n, double* __restrict__ part) {
  __shared__ double red[16];
  double s = 0.0;
  const int64_t per = (n + gridDim.x - 1) / gridDim.x;
  const int64_t b = blockIdx.x * per, e = b + per < n ? b + per : n;
  for (int64_t i = b + threadIdx.x; i < e; i += blockDim.x) s += (double)v[i];
  s = block_sum<double>(s, red);
  if (threadIdx.x == 0) part[blockIdx.x] = s;
}
__global__ void k_km_sum_final(const double* __restrict__ part, int n, double* __restrict__ out) {
  __shared__ double red[16];
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) s += part[i];
  s = block_sum<double>(s, red);
  if (threadIdx.x == 0) out[0] = s;
}

}  // namespace agp
