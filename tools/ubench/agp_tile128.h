// agp_tile128.h -- the fp32 GEMM-shaped kernels of the CAVI path on 128 x 128 C tiles (round 5).
//
// Until round 4 the fp32 instantiations ran on the fp64 kernels' geometry: 64 x 64 x 16 tiles, v_mfma_f32_16x16x4, one scalar LDS
// read per operand per MFMA -- 16 flop per operand byte through the L2, 0.40 of the fp32 MFMA peak at C3 (m = B = 2048:
// k_syrk_tn<float> 137 us for 8.6 GF, 280 MB of fabric traffic for 34 MB of operands).  Here:
//   * one workgroup = 512 threads = 8 waves as 2 (M) x 4 (N), two per SIMD (these launches have one workgroup per CU: with one wave
//     per SIMD every LDS wait and every barrier is exposed -- measured 84 TF for the kappa GEMM against 97 TF on the 64-tiles); a
//     wave owns 64 x 32 of C as two v_mfma_f32_32x32x2 tiles (32 accumulator registers): one operand register feeds 4096 flop
//     instead of 2048, a 128 x 128 x 32 slab moves 32 flop per operand byte;
//   * operands staged in LDS as [k][row] (row contiguous): an MFMA operand fetch is one ds_read_b32 of 32 consecutive floats per
//     half-wave -- conflict-free by construction; slabs of 32 k, double-buffered, the next slab's global loads (16-byte) in flight
//     under the 32 MFMAs per wave of the current one, one barrier per slab;
//   * the symmetric product  S = A' diag(w) A  (analyticVI.jl:172-180; src/functions/utils.jl:70-72) has only nt (nt + 1) / 2 lower
//     tiles -- 136 at m = 2048 for 256 CUs -- so it is STREAM-K: the tile x slab iteration space is cut into one equal range per
//     workgroup (one workgroup per CU); a tile that spans several workgroups is finished by the workgroup that holds its first
//     slabs, which adds the others' partial tiles IN SLAB ORDER (bitwise reproducible).  Partial tiles travel through
//     sentinel-validated slots like every other hand-over of this library (agp_chol.h, "self-validating hand-over"): write-through
//     stores, coherent loads, every element checked against the sentinel -- no agent-scope release fence.  A workgroup posts the
//     partial tile of its first range (the tail of a tile) BEFORE it can wait for anything, so waits only ever point at workgroups
//     that need nothing to get there.
// Reference operations replaced: kappa = Knm / K (latentgp.jl:205-215, k_gemm128_nt), rho kappa' diag(theta) kappa (utils.jl:70-72,
// k_syrk128_tn).
#pragma once
#include "../../augmentedgaussianprocesses.jl_amd/csrc/agp_linalg.h"

namespace agp {

typedef float f16v __attribute__((ext_vector_type(16)));

constexpr int T128 = 128;       // C tile edge
constexpr int BK128 = 64;       // k per staged slab (32: one barrier per 32 MFMAs per wave cost the kappa GEMM 20 % of its time)
constexpr int NT128 = 512;      // threads per workgroup
constexpr int LDR128 = 132;     // [k][row] row stride, row-contiguous operands (16-byte aligned rows for ds_write_b128)
constexpr int LDK128 = 129;     // [k][row] row stride, k-contiguous operands (transposed ds_write_b32: 4 k x 8 rows hit 32 banks)
constexpr int OPER128 = BK128 * LDR128;          // floats per staged operand (the larger stride)
constexpr int SMEM128 = 2 * 2 * OPER128;         // double-buffered A and B: 135168 bytes (one workgroup per CU)
constexpr int LDT128 = 129;     // transposed epilogue tile [col][row]
static_assert(SMEM128 >= T128 * LDT128, "the epilogue's transposed tile must fit into the staging area");

struct Acc128 {
  f16v a[2];  // rows wm 64 + mi 32 .., columns wn 32 ..
  __device__ __forceinline__ void zero() {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) a[i][r] = 0.0f;
  }
};

// f(row_in_tile, col_in_tile, value&) over this thread's 32 accumulator elements.  v_mfma_f32_32x32x2 C/D layout: col = lane & 31,
// row = 8 (r >> 2) + 4 (lane >> 5) + (r & 3), r in [0, 16)
template <typename F>
__device__ __forceinline__ void acc128_foreach(Acc128& acc, F f) {
  const int lane = threadIdx.x & 63, wave = (threadIdx.x >> 6) & 7, wm = wave >> 2, wn = wave & 3;
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float v = acc.a[mi][r];  // (a vector element does not bind to a reference)
      f(wm * 64 + mi * 32 + 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3), wn * 32 + (lane & 31), v);
      acc.a[mi][r] = v;
    }
}

// ---- operand staging ---------------------------------------------------------------------------------------------------------
// RC128: element (row, k) at P[k * ld + row] (row contiguous; P at (k = 0, row0)).  KC128: element (row, k) at P[row * ld + k].
struct RC128 {
  static constexpr int LD = LDR128;
  static constexpr int NV = BK128 * T128 / 4 / NT128;  // 16-byte vectors per thread and slab (4)
  static constexpr int NCHUNK = NV;                    // LDS store chunks per slab: one ds_write_b128 each
  f4 v[NV];
  float wv[NV];
  // thread t: 16-byte vector rv = t & 31 (rows 4 rv ..), k = (t >> 5) + 16 j; the diag(w) entries of its k travel with the slab
  __device__ __forceinline__ void load_one(int j, const float* __restrict__ P, int64_t ld, int64_t k0, int tid, const float* __restrict__ wscale) {
    const int rv = tid & 31, kq = tid >> 5;
    v[j] = *reinterpret_cast<const f4*>(P + (k0 + kq + 16 * j) * ld + 4 * rv);
    wv[j] = wscale ? wscale[k0 + kq + 16 * j] : 1.0f;
  }
  __device__ __forceinline__ void store_chunk(int j, float* S, int tid, bool scaled) const {
    const int rv = tid & 31, kq = tid >> 5;
    f4 x = v[j];
    if (scaled) {
      const float w = wv[j];
      x[0] *= w, x[1] *= w, x[2] *= w, x[3] *= w;
    }
    *reinterpret_cast<f4*>(S + (kq + 16 * j) * LDR128 + 4 * rv) = x;
  }
};
struct KC128 {
  static constexpr int LD = LDK128;
  static constexpr int NV = BK128 * T128 / 4 / NT128;
  static constexpr int NCHUNK = 2 * NV;  // LDS store chunks per slab: two ds_write_b32 each (the compiler pairs them where it can)
  f4 v[NV];
  // thread t: k vector kv = t & 15 (k = 4 kv ..), row = (t >> 4) + 32 j
  __device__ __forceinline__ void load_one(int j, const float* __restrict__ P, int64_t ld, int64_t k0, int tid, const float* __restrict__ wscale) {
    (void)wscale;
    const int kv = tid & 15, r = tid >> 4;
    v[j] = *reinterpret_cast<const f4*>(P + (int64_t)(r + 32 * j) * ld + k0 + 4 * kv);
  }
  __device__ __forceinline__ void store_chunk(int c, float* S, int tid, bool scaled) const {
    (void)scaled;
    const int kv = tid & 15, r = tid >> 4;  // a half-wave: 16 k vectors x 2 rows -> (4 kv + e) * 129 + r: two lanes per bank (free)
    const int j = c >> 1, e0 = (c & 1) * 2;
    S[(4 * kv + e0) * LDK128 + r + 32 * j] = v[j][e0];
    S[(4 * kv + e0 + 1) * LDK128 + r + 32 * j] = v[j][e0 + 1];
  }
};
template <typename L>
__device__ __forceinline__ void load_slab128(L& r, const float* __restrict__ P, int64_t ld, int64_t k0, int tid, const float* __restrict__ wscale) {
#pragma unroll
  for (int j = 0; j < L::NV; ++j) r.load_one(j, P, ld, k0, tid, wscale);
}
template <typename L>
__device__ __forceinline__ void store_slab128(const L& r, float* S, int tid, bool scaled) {
#pragma unroll
  for (int c = 0; c < L::NCHUNK; ++c) r.store_chunk(c, S, tid, scaled);
}

// the 64 MFMAs per wave of one staged slab: wave (wm, wn) takes rows wm 64 .. of A and rows wn 32 .. of B.  A rolling window of 8
// k-steps of operands (24 registers): all of them are requested before the first MFMA, and the operands of k-step s + 8 are requested
// into the registers of k-step s right behind its MFMAs -- the LDS returns them in order, an MFMA waits for its own three only (left
// to itself the compiler fetched a k-step's A pair, waited, issued two MFMAs: the LDS latency on every k-step).
// `fill(s)` is issued behind the MFMAs of k-step s: the pipeline's other instructions -- the global loads of the slab after next, the
// LDS stores of the next slab -- are dealt out over the k-steps, a few per MFMA pair, into issue slots the MFMA pipe leaves free.
// (Issued in lumps -- all loads at the top, all stores in the middle -- both waves of a SIMD did them at the same time and the pipe
// idled: ~10 us each for loads, stores and the barrier's wake-up in the 2048^3 kappa GEMM.)
template <int LDA, int LDB, typename FILL>
__device__ __forceinline__ void mma128_slab(const float* As, const float* Bs, Acc128& acc, int wm, int wn, int lane, FILL fill) {
  const float* pa = As + (lane >> 5) * LDA + wm * 64 + (lane & 31);
  const float* pb = Bs + (lane >> 5) * LDB + wn * 32 + (lane & 31);
  constexpr int W = 8, NS = BK128 / 2;  // window (8 k-steps = 1024 MFMA cycles of look-ahead for a ~200-cycle LDS), k-steps per slab
  float a0[W], a1[W], b0[W];
#pragma unroll
  for (int s = 0; s < W; ++s) {
    a0[s] = pa[2 * s * LDA];
    a1[s] = pa[2 * s * LDA + 32];
    b0[s] = pb[2 * s * LDB];
  }
  __builtin_amdgcn_sched_barrier(0);  // (the machine scheduler would sink the loads back between the MFMAs)
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    acc.a[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[s % W], b0[s % W], acc.a[0], 0, 0, 0);
    acc.a[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[s % W], b0[s % W], acc.a[1], 0, 0, 0);
    if (s + W < NS) {
      a0[s % W] = pa[2 * (s + W) * LDA];
      a1[s % W] = pa[2 * (s + W) * LDA + 32];
      b0[s % W] = pb[2 * (s + W) * LDB];
    }
    fill(s);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// acc += sum over slabs [s0, s1) of A_op B_op'  (A, B at their 128-row origins; wscaleA folds diag(w) into A on its way into LDS)
// smem: SMEM128 floats.  All 512 threads; ends with a barrier (smem free again).
// Pipeline: slab s is multiplied out of LDS buffer s & 1 while slab s + 1 (in registers since the iteration before: its global loads
// have had a whole slab of MFMAs to land) is stored to the other buffer, and slab s + 2's loads are issued.  One barrier per slab.
template <typename LA, typename LB>
__device__ __forceinline__ void gemm128_slabs(const float* __restrict__ A, int64_t lda, const float* __restrict__ B, int64_t ldb,
                                              int64_t s0, int64_t s1, const float* __restrict__ wscaleA, Acc128& acc, float* smem) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 2, wn = wave & 3;
  if (s0 >= s1) return;
  const bool scaled = wscaleA != nullptr;
  LA ra0, ra1;
  LB rb0, rb1;
  load_slab128(ra0, A, lda, s0 * BK128, tid, wscaleA);
  load_slab128(rb0, B, ldb, s0 * BK128, tid, nullptr);
  if (s0 + 1 < s1) {
    load_slab128(ra1, A, lda, (s0 + 1) * BK128, tid, wscaleA);
    load_slab128(rb1, B, ldb, (s0 + 1) * BK128, tid, nullptr);
  }
  store_slab128(ra0, smem, tid, scaled);
  store_slab128(rb0, smem + OPER128, tid, false);
  __syncthreads();
  // one iteration: multiply slab s out of buffer `cur`; (ras, rbs) hold slab s + 1 -> stored to the other buffer; (raf, rbf), stored an
  // iteration ago, are refilled with slab s + 2.  Schedule over the 32 k-steps: loads at k-steps 0 .. NV-1 (A) and NV .. 2 NV - 1 (B),
  // store chunks from k-step 8 on, A's then B's, CPS per k-step, all issued by k-step 24
  constexpr int NS = BK128 / 2, NV = LA::NV, CH = LA::NCHUNK + LB::NCHUNK, S0 = 8, CPS = (CH + 15) / 16;
  static_assert(LA::NV == LB::NV && 2 * NV <= S0 && S0 + (CH + CPS - 1) / CPS <= NS, "gemm128_slabs: fill schedule");
  auto iter = [&](int64_t s, int cur, LA& ras, LB& rbs, LA& raf, LB& rbf) {
    const float* As = smem + cur * 2 * OPER128;
    float* Ns = smem + (cur ^ 1) * 2 * OPER128;
    // No branches in here: behind a conditional load the compiler's wait-count bookkeeping falls back to "wait for every load" in
    // front of the stores -- the full memory latency of the loads just issued, every slab (measured: 0.4 us per slab).  The last two
    // iterations therefore load the last slab again and store into a buffer nobody reads.
    const int64_t k2 = ((s + 2 < s1) ? s + 2 : s1 - 1) * BK128;
    mma128_slab<LA::LD, LB::LD>(As, As + OPER128, acc, wm, wn, lane, [&](int ks) {
      if (ks < NV) {
        raf.load_one(ks, A, lda, k2, tid, wscaleA);
      } else if (ks < 2 * NV) {
        rbf.load_one(ks - NV, B, ldb, k2, tid, nullptr);
      } else if (ks >= S0) {
#pragma unroll
        for (int q = 0; q < CPS; ++q) {
          const int c = (ks - S0) * CPS + q;
          if (c < LA::NCHUNK) ras.store_chunk(c, Ns, tid, scaled);
          else if (c < CH) rbs.store_chunk(c - LA::NCHUNK, Ns + OPER128, tid, false);
        }
      }
    });
    __syncthreads();
  };
  int64_t s = s0;
  for (; s + 1 < s1; s += 2) {
    iter(s, 0, ra1, rb1, ra0, rb0);
    iter(s + 1, 1, ra0, rb0, ra1, rb1);
  }
  if (s < s1) iter(s, 0, ra1, rb1, ra0, rb0);
}

// ---- hand-over of partial tiles (stream-K) --------------------------------------------------------------------------------------
// slot = 128 x 128 floats, thread-major: 16-byte group q of thread t at (q * 512 + t) * 4
__device__ __forceinline__ void store16_sc1(void* p, u32x4 v) {
  asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void acc128_post(const Acc128& acc, float* __restrict__ slot) {
  const int tid = threadIdx.x;
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    u32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = __builtin_bit_cast(unsigned int, (float)acc.a[q >> 2][(q & 3) * 4 + e]);
    store16_sc1(slot + ((int64_t)q * NT128 + tid) * 4, v);
  }
}
// acc += slot (every element validated against the sentinel, re-loaded until it has arrived), then the slot gets its sentinels back
__device__ __forceinline__ void acc128_take(Acc128& acc, float* __restrict__ slot) {
  const int tid = threadIdx.x;
  const unsigned int sb = Sent<float>::bits;
  u32x4 pv[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) pv[q] = load16_sc1(slot + ((int64_t)q * NT128 + tid) * 4);
  wait_vmcnt0();
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    float* p = slot + ((int64_t)q * NT128 + tid) * 4;
#pragma unroll
    for (int e = 0; e < 4; ++e)
      acc.a[q >> 2][(q & 3) * 4 + e] += hv_settle<float>(p + e, __builtin_bit_cast(float, (unsigned int)pv[q][e]));
    u32x4 s;
    s[0] = s[1] = s[2] = s[3] = sb;
    *reinterpret_cast<u32x4*>(p) = s;  // plain store: in memory by the end of this kernel, before the next launch posts here
  }
}

// stream-K bookkeeping shared by host and device: `total` = tiles * nslab iterations in G ranges of `per`
__host__ __device__ __forceinline__ int64_t sk_per(int64_t total, int64_t G) { return (total + G - 1) / G; }
// slots per tile: a tile of nslab iterations overlaps at most (nslab - 2) / per + 2 ranges; the first one owns it
__host__ __device__ __forceinline__ int64_t sk_slots(int64_t nslab, int64_t per) { return (nslab + per - 2) / per + 1; }

// ---------------------------------------------------------------------------------------------------------------------------------
// S = A' diag(w) A on 128-tiles, stream-K.  A: Kdim x n (row-contiguous), n a multiple of 128, Kdim a multiple of 32.
// grid = G workgroups, one range of iterations each (G = the number of CUs).  ws: partial-tile slots, tiles * sk_slots(..) * 128 * 128
// floats, sentinel-filled (the kernel leaves them sentinel-filled).
// ---------------------------------------------------------------------------------------------------------------------------------
// The eta2 / K^-1 values of the tile's lower triangle are touched (loaded into a scratch register, never used) by the owner of a split
// tile BEFORE it waits for the other ranges' partial tiles: the epilogue's own loads then find them in the L2 -- one HBM latency
// hidden behind the hand-over instead of two in front of the stores.
// (The load is inline assembly, outside the compiler's wait-count bookkeeping: its destination register must stay allocated until the
//  caller has waited -- touch_done() behind a wait_vmcnt0() -- or a late return lands in a register that has been given to
//  something else: a memory fault with a garbage address in the first version.)
__device__ __forceinline__ float touch_f32(const float* p) {
  float t;
  asm volatile("global_load_dword %0, %1, off" : "=v"(t) : "v"(p) : "memory");
  return t;
}
__device__ __forceinline__ void touch_done(float a, float b) { asm volatile("" ::"v"(a), "v"(b)); }
struct Touch128 {
  float a = 0.0f, b = 0.0f;
};
template <int MODE>
__device__ __forceinline__ Touch128 syrk128_epi_touch(int64_t ta, int64_t tb, const float* __restrict__ eta2,
                                                      const float* __restrict__ Kinv, int64_t ldm) {
  Touch128 t;
  if (MODE != SY_ETA2) return t;
  const int64_t a0 = ta * T128, b0 = tb * T128;
  // one 4-byte touch per 128-byte line: 128 rows x 4 lines x 2 matrices = 1024 lines, two per thread
  const int r = threadIdx.x >> 2, q = threadIdx.x & 3;
  t.a = touch_f32(eta2 + (a0 + r) * ldm + b0 + 32 * q);
  t.b = touch_f32(Kinv + (a0 + r) * ldm + b0 + 32 * q);
  return t;
}

template <int MODE>
__device__ __forceinline__ void syrk128_epilogue(Acc128& acc, int64_t ta, int64_t tb, float* __restrict__ out, int64_t ldo,
                                                 float* __restrict__ eta2, const float* __restrict__ Kinv, int64_t ldm, float lr,
                                                 float* smem) {
  // The tile goes through LDS ([row][col], stride 129) and both passes are plain loops over its elements -- rows of the tile for the
  // lower triangle, rows of the MIRRORED tile for the upper one (the 64-tile kernels write the mirror image as 4-byte stores a row
  // apart) -- instead of 64 unrolled element visits with their addresses in registers (252 registers + scratch).
  const int64_t a0 = ta * T128, b0 = tb * T128;
  const bool diag = ta == tb;
  const int tid = threadIdx.x;
  // element e = tid + 512 i of the tile: row (tid >> 7) + 4 i, column tid & 127; 16 elements (32 loads) per batch
  constexpr int NE = T128 * T128 / NT128, NB = 16;
  const int er = tid >> 7, ec = tid & 127;
  acc128_foreach(acc, [&](int r, int c, float& val) { smem[r * LDT128 + c] = val; });
  __syncthreads();
#pragma unroll 1
  for (int i0 = 0; i0 < NE; i0 += NB) {
    float e2v[NB], kiv[NB];
    if (MODE == SY_ETA2) {
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        const int r = er + 4 * (i0 + i);
        const bool on = !diag || ec <= r;
        e2v[i] = on ? eta2[(a0 + r) * ldm + b0 + ec] : 0.0f;
        kiv[i] = on ? Kinv[(a0 + r) * ldm + b0 + ec] : 0.0f;
      }
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int r = er + 4 * (i0 + i), c = ec;
      if (diag && c > r) continue;  // diagonal tile: the lower half is the truth
      const int64_t gr = a0 + r, gc = b0 + c;
      float val = smem[r * LDT128 + c];
      if (MODE == SY_ETA2) {  // g = -(S + Kinv / 2) - eta2 ; eta2 += lr g   (analyticVI.jl:172-180, 229-246)
        float e2 = e2v[i];
        const float g = -(val + 0.5f * kiv[i]) - e2;
        e2 += lr * g;
        val = e2;
        smem[r * LDT128 + c] = val;
        eta2[gr * ldm + gc] = val;
        out[gr * ldo + gc] = -2.0f * val;
      } else {
        out[gr * ldo + gc] = val;
      }
    }
  }
  __syncthreads();
  // mirrored part: element (mr, mc) of the mirrored tile = value (mc, mr) of the tile
#pragma unroll 4
  for (int e = tid; e < T128 * T128; e += NT128) {
    const int mr = e >> 7, mc = e & 127;
    if (diag && mc <= mr) continue;  // (diagonal tile: only its strict upper half is a mirror image)
    const float val = smem[mc * LDT128 + mr];
    const int64_t gr = b0 + mr, gc = a0 + mc;
    if (MODE == SY_ETA2) {
      eta2[gr * ldm + gc] = val;
      out[gr * ldo + gc] = -2.0f * val;
    } else {
      out[gr * ldo + gc] = val;
    }
  }
  __syncthreads();
}

// The riders of k_syrk_tn -- nt workgroups that take the eta1 step (t = A' r, analyticVI.jl:160-169), 96 that refill a dirty hand-over
// set of the task graph -- do not exist here: a stream-K launch is one equal range per CU, a rider that gets a CU to itself only when
// a main workgroup retires is a serial tail (measured: 32 eta1 riders of 256 dependent loads each cost the launch ~70 us).  Both jobs
// are cut into G equal shares and every main workgroup takes one BEFORE its range: 8 columns of t over all k (64 KB of A, 32-byte
// row pieces that the four workgroups next to it on the same XCD share line by line) and 1 / G of the refill stores.
__device__ __forceinline__ void syrk128_eta1_share(const float* __restrict__ A, int64_t lda, int64_t Kdim, int64_t n,
                                                   const float* __restrict__ rvec, float* __restrict__ eta1,
                                                   const float* __restrict__ kinv_mu0, float lr, int64_t g, int64_t G, float* smem) {
  // columns [c_lo, c_lo + cw) of t, cw = 4 ceil(n / (4 G)) <= 32.  Thread t: k = t, t + 512, ...; a float4 of four columns per load.
  const int64_t cw = 4 * ((n + 4 * G - 1) / (4 * G)), c_lo = g * cw;
  if (c_lo >= n) return;
  const int nv = (int)(((c_lo + cw <= n) ? cw : n - c_lo) / 4);  // float4 groups of this share (n is a multiple of 4)
  const int tid = threadIdx.x;
  for (int v = 0; v < nv; ++v) {
    f4 s = {0.0f, 0.0f, 0.0f, 0.0f};
    const float* col = A + c_lo + 4 * v;
#pragma unroll 4
    for (int64_t k = tid; k < Kdim; k += NT128) {
      const f4 x = *reinterpret_cast<const f4*>(col + k * lda);
      const float rk = rvec[k];
      s[0] += x[0] * rk, s[1] += x[1] * rk, s[2] += x[2] * rk, s[3] += x[3] * rk;
    }
    // fixed-order reduction: lanes by halving shuffles, then the eight waves in order
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float t = s[e];
      for (int o = 32; o > 0; o >>= 1) t += __shfl_down(t, o);
      if ((tid & 63) == 0) smem[(tid >> 6) * 4 + e] = t;
    }
    __syncthreads();
    if (tid < 4) {
      float t = 0.0f;
      for (int q = 0; q < NT128 / 64; ++q) t += smem[q * 4 + tid];
      const int64_t c = c_lo + 4 * v + tid;
      const float e = eta1[c];
      eta1[c] = e + lr * (t + (kinv_mu0 ? kinv_mu0[c] : 0.0f) - e);
    }
    __syncthreads();
  }
}

template <int MODE>
__global__ __launch_bounds__(NT128) void k_syrk128_tn(const float* __restrict__ A, int64_t lda, int64_t Kdim, const float* __restrict__ w,
                                                      float* __restrict__ out, int64_t ldo, float* __restrict__ eta2,
                                                      const float* __restrict__ Kinv, int64_t ldm, float lr, int64_t ntiles, int64_t n,
                                                      float* __restrict__ ws, const float* __restrict__ rvec, float* __restrict__ eta1,
                                                      const float* __restrict__ kinv_mu0, float* __restrict__ fillp, int64_t fill_used,
                                                      int64_t fill_stride, int fill_nb) {
  static_assert(MODE == SY_STORE || MODE == SY_ETA2, "k_syrk128_tn: store or fused eta2 step");
  __shared__ __attribute__((aligned(16))) float smem[SMEM128];
  const int64_t G = gridDim.x;
  if (fillp) {  // this workgroup's share of the hand-over refill (fire-and-forget stores; 16 bytes per lane)
    const u32x4 sv = {Sent<float>::bits, Sent<float>::bits, Sent<float>::bits, Sent<float>::bits};
    const int64_t nv = fill_used / 4;
    for (int q = 0; q < fill_nb; ++q) {
      u32x4* dst = reinterpret_cast<u32x4*>(fillp + q * fill_stride);
      for (int64_t i = (int64_t)blockIdx.x * NT128 + threadIdx.x; i < nv; i += G * NT128) dst[i] = sv;
      for (int64_t i = 4 * nv + (int64_t)blockIdx.x * NT128 + threadIdx.x; i < fill_used; i += G * NT128)
        fillp[q * fill_stride + i] = __builtin_bit_cast(float, Sent<float>::bits);
    }
  }
  // this workgroup's range of the iteration space (iteration = tile * nslab + slab); XCD x = blockIdx % 8 owns a contiguous stretch
  const int64_t nslab = Kdim / BK128, total = ntiles * nslab, per = sk_per(total, G), nslots = sk_slots(nslab, per);
  const int64_t g = xcd_contiguous((int64_t)blockIdx.x, G);
  if (rvec) syrk128_eta1_share(A, lda, Kdim, n, rvec, eta1, kinv_mu0, lr, g, G, smem);
  int64_t it = g * per;
  const int64_t it_end = (it + per < total) ? it + per : total;
  while (it < it_end) {
    const int64_t t = it / nslab, s0 = it - t * nslab;
    const int64_t s1 = (it_end - t * nslab < nslab) ? it_end - t * nslab : nslab;
    int64_t ta, tb;
    tri_index(t, ta, tb);
    Acc128 acc;
    acc.zero();
    gemm128_slabs<RC128, RC128>(A + ta * T128, lda, A + tb * T128, lda, s0, s1, w, acc, smem);
    if (s0 != 0) {
      // the tail (or the middle) of a tile that another workgroup owns: post it.  ordinal of this range within the tile, 1-based
      const int64_t ord = g - (t * nslab) / per;
      acc128_post(acc, ws + (t * nslots + (ord - 1)) * (int64_t)(T128 * T128));
    } else {
      // first slabs of the tile: this workgroup owns it; the later ranges arrive in slab order
      Touch128 tch;
      const bool split = s1 < nslab;
      if (split) tch = syrk128_epi_touch<MODE>(ta, tb, eta2, Kinv, ldm);
      int64_t done = s1, ord = 1;
      while (done < nslab) {
        acc128_take(acc, ws + (t * nslots + (ord - 1)) * (int64_t)(T128 * T128));
        // range g + ord covers [(g + ord) per, (g + ord + 1) per) of the iteration space
        const int64_t hi = (g + ord + 1) * per - t * nslab;
        done = hi < nslab ? hi : nslab;
        ++ord;
      }
      if (split) {
        wait_vmcnt0();
        touch_done(tch.a, tch.b);
      }
      syrk128_epilogue<MODE>(acc, ta, tb, out, ldo, eta2, Kinv, ldm, lr, smem);
    }
    it = t * nslab + s1;
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// C(M x N) = A(M x K) B(N x K)' on 128-tiles, both operands k-contiguous.  grid = (N / 128) (M / 128) workgroups, XCD-aware tile
// order.  EPI_STORE: C = acc.  EPI_KAPPA: C = acc, part1 = acc (same ldc: the factorisation workspace) and the row-dot partial sums
// part0[slice][row] = sum over the slice's 32 columns of acc * E[row][col], slice = column / 32 -- the slices k_gemm_nt<EPI_KAPPA>
// writes (two per 64-column tile), summed by the same consumers in the same order.
// ---------------------------------------------------------------------------------------------------------------------------------
template <int EPI>
__global__ __launch_bounds__(NT128) void k_gemm128_nt(const float* __restrict__ A, int64_t lda, const float* __restrict__ B, int64_t ldb,
                                                      int64_t K, float* __restrict__ C, int64_t ldc, const float* __restrict__ E,
                                                      int64_t lde, float* __restrict__ part0, float* __restrict__ part1, int64_t ldp,
                                                      int64_t gx, int64_t gy) {
  static_assert(EPI == EPI_STORE || EPI == EPI_KAPPA, "k_gemm128_nt: store or kappa epilogue");
  __shared__ __attribute__((aligned(16))) float smem[SMEM128];
  int64_t bm, bn;
  banded_tile(xcd_contiguous((int64_t)blockIdx.x, gx * gy), gx, gy, bm, bn);
  const int64_t r0 = bm * T128, c0 = bn * T128;
  Acc128 acc;
  acc.zero();
  gemm128_slabs<KC128, KC128>(A + r0 * lda, lda, B + c0 * ldb, ldb, 0, K / BK128, nullptr, acc, smem);
  acc128_foreach(acc, [&](int r, int c, float& val) {
    C[(r0 + r) * ldc + c0 + c] = val;
    if (EPI == EPI_KAPPA && part1) part1[(r0 + r) * ldc + c0 + c] = val;
  });
  if (EPI == EPI_KAPPA) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wm = wave >> 2, wn = wave & 3;
    float* p0 = part0 + ((c0 + wn * 32) / 32) * ldp;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wm * 64 + mi * 32 + 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3), col = wn * 32 + (lane & 31);
        float s = acc.a[mi][r] * E[(r0 + row) * lde + c0 + col];
        s += __shfl_xor(s, 1);
        s += __shfl_xor(s, 2);
        s += __shfl_xor(s, 4);
        s += __shfl_xor(s, 8);
        s += __shfl_xor(s, 16);
        if ((lane & 31) == 0) p0[r0 + row] = s;
      }
  }
}

}  // namespace agp
