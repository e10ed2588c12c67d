// micro-benchmark: Gauss-Jordan of a 16x16 SPD block [A | I] inside ONE wave, no LDS traffic and no barrier between the
// columns.  Lane c (of every 16-lane DPP row) holds COLUMN c, register r holds row r, so the row operation
//     row_r -= (A[r][j] / d_j) row_j
// is one v_fmac per register: the multiplier A[r][j] sits in lane j of the SAME register and comes through the DPP
// row_newbcast:j control of the instruction itself (gfx90a+: v_mov_b64 / v_fmac_f64 accept it), the pivot-row entry
// -A[j][c] / d_j is the lane's own value.  Dependent chain per column: pivot broadcast -> rcp (+ Newton) -> scale -> first fmac.
// Variants: 0 = compiler builtin (v_mov_b64_dpp + v_fma), 1 = inline-asm fused v_fmac_*_dpp
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int J>
__device__ __forceinline__ double bc(double v) { return __builtin_amdgcn_update_dpp(v, v, 0x150 + J, 0xf, 0xf, true); }
template <int J>
__device__ __forceinline__ float bc(float v) { return __builtin_amdgcn_update_dpp(v, v, 0x150 + J, 0xf, 0xf, true); }
// the 64-bit broadcast as two 32-bit DPP moves (v_mov_b32_dpp is a full-rate op, v_mov_b64_dpp is not)
template <int J>
__device__ __forceinline__ double bc32(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x150 + J, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x150 + J, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
template <int J>
__device__ __forceinline__ float bc32(float v) { return bc<J>(v); }
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

// acc += bcast_J(src) * own
template <int J>
__device__ __forceinline__ void fmac_bc(double& acc, double src, double own) {
  asm("s_nop 1\n\tv_fmac_f64 %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(own), "n"(J));
}
template <int J>
__device__ __forceinline__ void fmac_bc(float& acc, float src, float own) {
  asm("s_nop 1\n\tv_fmac_f32 %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(own), "n"(J));
}

__device__ __forceinline__ double rcp1(double p) {
  double r = __builtin_amdgcn_rcp(p);
  return fma(r, fma(-p, r, 1.0), r);
}
__device__ __forceinline__ float rcp1(float p) {
  float r = __builtin_amdgcn_rcpf(p);
  return fmaf(r, fmaf(-p, r, 1.0f), r);
}

template <typename T, int VAR, int J>
struct Col {
  static __device__ __forceinline__ void run(T (&a)[16], T (&m)[16], T (&dv)[16]) {
    const T d = (VAR >= 2) ? bc32<J>(a[J]) : bc<J>(a[J]);
    dv[J] = d;
    const T nr = -rcp1(d);
    const T t = a[J] * nr, tm = m[J] * nr;
#pragma unroll
    for (int r = J + 1; r < 16; ++r) {
      if (VAR == 0) {
        const T l = bc<J>(a[r]);
        a[r] = fma(l, t, a[r]);
        m[r] = fma(l, tm, m[r]);
      } else if (VAR == 2) {
        const T l = bc32<J>(a[r]);
        a[r] = fma(l, t, a[r]);
        m[r] = fma(l, tm, m[r]);
      } else if (VAR == 3) {  // A only (no running inverse): what the inverse costs
        const T l = bc32<J>(a[r]);
        a[r] = fma(l, t, a[r]);
      } else {
        fmac_bc<J>(m[r], a[r], tm);
        fmac_bc<J>(a[r], a[r], t);
      }
    }
    if constexpr (J + 1 < 16) Col<T, VAR, J + 1>::run(a, m, dv);
  }
};

// in: S[r*LD + c] SPD (lower valid).  out: L (Cholesky factor) in S, X = L^-1 in Xo.  One wave; all four DPP rows do the same.
template <typename T, int VAR>
__device__ __forceinline__ void elim16(T* S, T* Xo, int LD) {
  const int c = threadIdx.x & 15;
  T a[16], m[16], dv[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    a[r] = r >= c ? S[r * LD + c] : S[c * LD + r];
    m[r] = r == c ? T(1) : T(0);
  }
  Col<T, VAR, 0>::run(a, m, dv);
  // a[j] in lane c = d_j L1[c][j] (c >= j) ; m[r] in lane c = M[r][c], M = L1^-1 (unit lower) ; L = L1 D^1/2, X = D^-1/2 M
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const T rs = rsqrt1(dv[j]);
    if (threadIdx.x < 16) {
      S[c * LD + j] = c >= j ? a[j] * rs : T(0);
      Xo[j * LD + c] = j >= c ? m[j] * rs : T(0);
    }
  }
}

template <typename T, int VAR>
__global__ __launch_bounds__(64) void k_bench(const T* A, T* out, int reps) {
  __shared__ T S[16 * 18], Xo[16 * 18], S0[256];
  for (int e = threadIdx.x; e < 256; e += 64) S0[e] = A[e];
  __syncthreads();
  for (int it = 0; it < reps; ++it) {
    for (int e = threadIdx.x; e < 256; e += 64) S[(e >> 4) * 18 + (e & 15)] = S0[e];
    __syncthreads();
    if (VAR != 9) elim16<T, VAR>(S, Xo, 18);
  }
  for (int e = threadIdx.x; e < 256; e += 64) {
    out[e] = S[(e >> 4) * 18 + (e & 15)];
    out[256 + e] = Xo[(e >> 4) * 18 + (e & 15)];
  }
}

template <typename T, int VAR>
void run(const char* name) {
  std::vector<double> G(256), A(256, 0.0);
  srand(1);
  for (auto& g : G) g = rand() / (double)RAND_MAX - 0.5;
  for (int i = 0; i < 16; ++i)
    for (int j = 0; j < 16; ++j) {
      double s = i == j ? 1.0 : 0.0;
      for (int k = 0; k < 16; ++k) s += G[i * 16 + k] * G[j * 16 + k];
      A[i * 16 + j] = s;
    }
  std::vector<T> hA(256), ho(512);
  for (int i = 0; i < 256; ++i) hA[i] = (T)A[i];
  T *dA, *dO;
  hipMalloc(&dA, 256 * sizeof(T));
  hipMalloc(&dO, 512 * sizeof(T));
  hipMemcpy(dA, hA.data(), 256 * sizeof(T), hipMemcpyHostToDevice);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int reps = 2000;
  hipLaunchKernelGGL((k_bench<T, VAR>), dim3(1), dim3(64), 0, 0, dA, dO, 10);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k_bench<T, VAR>), dim3(1), dim3(64), 0, 0, dA, dO, reps);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  hipMemcpy(ho.data(), dO, 512 * sizeof(T), hipMemcpyDeviceToHost);
  double r1 = 0, r2 = 0;
  for (int i = 0; i < 16; ++i)
    for (int j = 0; j < 16; ++j) {
      double s = 0, x = 0;
      for (int k = 0; k < 16; ++k) {
        s += (double)ho[i * 16 + k] * (double)ho[j * 16 + k];
        x += (double)ho[256 + i * 16 + k] * (double)ho[k * 16 + j];
      }
      r1 = fmax(r1, fabs(s - A[i * 16 + j]));
      r2 = fmax(r2, fabs(x - (i == j ? 1.0 : 0.0)));
    }
  printf("%-28s %7.3f us per 16x16 [A|I] elimination (incl. LDS load/store)   |LL'-A| %.2e  |XL-I| %.2e\n", name,
         ms * 1e3 / reps, r1, r2);
}

int main() {
  run<double, 0>("f64 builtin mov_dpp + fma");
  run<double, 1>("f64 fused v_fmac_f64_dpp");
  run<double, 2>("f64 2 x v_mov_b32_dpp + fma");
  run<double, 3>("f64 2 x v_mov_b32_dpp, A only");
  run<float, 0>("f32 builtin mov_dpp + fma");
  run<float, 1>("f32 fused v_fmac_f32_dpp");
  run<float, 3>("f32 A only");
  run<double, 9>("harness only");
  return 0;
}
