// micro-benchmark: single-wave dependent-issue latencies on gfx950 (unrolled chains, loop overhead amortised over 64 ops)
//   f64 / f32 fma, v_rcp_f64, v_mov_b64 dpp row_newbcast, v_readlane -> v_fma with an SGPR operand, ds_bpermute, LDS write->read in
//   one wave, s_barrier with 4 / 8 waves
#include <hip/hip_runtime.h>
#include <cstdio>
template <typename F>
float timeit(F f) {
  hipEvent_t a, b;
  (void)hipEventCreate(&a);
  (void)hipEventCreate(&b);
  f();
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(a);
  f();
  (void)hipEventRecord(b);
  (void)hipEventSynchronize(b);
  float ms;
  (void)hipEventElapsedTime(&ms, a, b);
  return ms;
}
constexpr int U = 64;
template <typename T>
__global__ void k_fma_dep(T* out, int iters, T b, T c) {
  T x = T(threadIdx.x) * T(1e-3);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < U; ++u) x = fma(x, b, c);
  }
  out[threadIdx.x] = x;
}
template <typename T, int NCH>
__global__ void k_fma_ind(T* out, int iters, T b, T c) {
  T x[NCH];
  for (int i = 0; i < NCH; ++i) x[i] = T(threadIdx.x + i) * T(1e-3);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < U / NCH; ++u)
#pragma unroll
      for (int i = 0; i < NCH; ++i) x[i] = fma(x[i], b, c);
  }
  T s = 0;
  for (int i = 0; i < NCH; ++i) s += x[i];
  out[threadIdx.x] = s;
}
__global__ void k_rcp_dep(double* out, int iters) {
  double x = 1.0 + threadIdx.x * 1e-3;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < U; ++u) x = __builtin_amdgcn_rcp(x);
  }
  out[threadIdx.x] = x;
}
__global__ void k_rsq_dep(double* out, int iters) {
  double x = 1.0 + threadIdx.x * 1e-3;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < U; ++u) x = __builtin_amdgcn_rsq(x);
  }
  out[threadIdx.x] = x;
}
__global__ void k_dpp_dep(double* out, int iters) {
  double x = 1.0 + threadIdx.x * 1e-3;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < U; ++u) x = __builtin_amdgcn_update_dpp(x, x, 0x150 + 5, 0xf, 0xf, true);
  }
  out[threadIdx.x] = x;
}
__global__ void k_dppfma_dep(double* out, int iters, double b) {
  double x = 1.0 + threadIdx.x * 1e-3;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < U; ++u) x = fma(__builtin_amdgcn_update_dpp(x, x, 0x150 + 5, 0xf, 0xf, true), b, x);
  }
  out[threadIdx.x] = x;
}
__global__ void k_readlane_dep(double* out, int iters, double b) {
  double x = 1.0 + threadIdx.x * 1e-3;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int lo = __builtin_amdgcn_readlane(__double2loint(x), 5), hi = __builtin_amdgcn_readlane(__double2hiint(x), 5);
      x = fma(__hiloint2double(hi, lo), b, x);
    }
  }
  out[threadIdx.x] = x;
}
__global__ void k_bperm_dep(double* out, int iters, double b) {
  double x = 1.0 + threadIdx.x * 1e-3;
  const int addr = ((threadIdx.x + 17) & 63) * 4;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int lo = __builtin_amdgcn_ds_bpermute(addr, __double2loint(x)), hi = __builtin_amdgcn_ds_bpermute(addr, __double2hiint(x));
      x = fma(__hiloint2double(hi, lo), b, x);
    }
  }
  out[threadIdx.x] = x;
}
__global__ void k_lds_wave(double* out, int iters, double b) {
  __shared__ double buf[64];
  double x = 1.0 + threadIdx.x * 1e-3;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      buf[threadIdx.x] = x;
      __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
      x = fma(buf[(threadIdx.x + 17) & 63], b, x);
    }
  }
  out[threadIdx.x] = x;
}
__global__ void k_barrier(double* out, int iters) {
  double x = threadIdx.x;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      __builtin_amdgcn_s_barrier();
      x += 1.0;
    }
  }
  out[threadIdx.x] = x;
}
__global__ void k_ldshop(double* out, int iters) {  // write -> barrier -> read, all waves
  __shared__ double buf[2][512];
  double v = threadIdx.x;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      buf[u & 1][threadIdx.x] = v;
      __syncthreads();
      v = buf[u & 1][(threadIdx.x + 17) & (blockDim.x - 1)] + 1.0;
    }
  }
  out[threadIdx.x] = v;
}
int main() {
  double* d;
  (void)hipMalloc(&d, 1 << 20);
  const int iters = 2000;
  const double n = (double)iters * U;
#define RUN(name, ...)                                   \
  {                                                      \
    float ms = timeit([&] { hipLaunchKernelGGL(__VA_ARGS__); }); \
    printf("%-52s %7.2f ns per op\n", name, ms * 1e6 / n); \
  }
  RUN("f64 fma dependent chain (1 wave)", k_fma_dep<double>, dim3(1), dim3(64), 0, 0, d, iters, 1.0000001, 1e-9);
  RUN("f32 fma dependent chain (1 wave)", k_fma_dep<float>, dim3(1), dim3(64), 0, 0, (float*)d, iters, 1.0000001f, 1e-9f);
  RUN("f64 fma 2 independent chains (1 wave)", (k_fma_ind<double, 2>), dim3(1), dim3(64), 0, 0, d, iters, 1.0000001, 1e-9);
  RUN("f64 fma 4 independent chains (1 wave)", (k_fma_ind<double, 4>), dim3(1), dim3(64), 0, 0, d, iters, 1.0000001, 1e-9);
  RUN("f64 fma 8 independent chains (1 wave)", (k_fma_ind<double, 8>), dim3(1), dim3(64), 0, 0, d, iters, 1.0000001, 1e-9);
  RUN("f32 fma 8 independent chains (1 wave)", (k_fma_ind<float, 8>), dim3(1), dim3(64), 0, 0, (float*)d, iters, 1.0000001f, 1e-9f);
  RUN("f64 fma 8 independent chains (4 waves, 1 per SIMD)", (k_fma_ind<double, 8>), dim3(1), dim3(256), 0, 0, d, iters, 1.0000001, 1e-9);
  RUN("f64 fma 8 independent chains (8 waves, 2 per SIMD)", (k_fma_ind<double, 8>), dim3(1), dim3(512), 0, 0, d, iters, 1.0000001, 1e-9);
  RUN("f64 fma dependent chain (8 waves, 2 per SIMD)", k_fma_dep<double>, dim3(1), dim3(512), 0, 0, d, iters, 1.0000001, 1e-9);
  RUN("v_rcp_f64 dependent chain", k_rcp_dep, dim3(1), dim3(64), 0, 0, d, iters);
  RUN("v_rsq_f64 dependent chain", k_rsq_dep, dim3(1), dim3(64), 0, 0, d, iters);
  RUN("v_mov_b64 dpp row_newbcast dependent chain", k_dpp_dep, dim3(1), dim3(64), 0, 0, d, iters);
  RUN("v_mov_b64 dpp + v_fma_f64 dependent pair", k_dppfma_dep, dim3(1), dim3(64), 0, 0, d, iters, 1e-9);
  RUN("2 v_readlane + v_fma_f64 dependent", k_readlane_dep, dim3(1), dim3(64), 0, 0, d, iters, 1e-9);
  RUN("2 ds_bpermute + v_fma_f64 dependent", k_bperm_dep, dim3(1), dim3(64), 0, 0, d, iters, 1e-9);
  RUN("LDS write -> read -> fma inside one wave", k_lds_wave, dim3(1), dim3(64), 0, 0, d, iters, 1e-9);
  RUN("s_barrier, 4 waves", k_barrier, dim3(1), dim3(256), 0, 0, d, iters);
  RUN("s_barrier, 8 waves", k_barrier, dim3(1), dim3(512), 0, 0, d, iters);
  RUN("LDS write -> barrier -> read, 4 waves", k_ldshop, dim3(1), dim3(256), 0, 0, d, iters);
  RUN("LDS write -> barrier -> read, 8 waves", k_ldshop, dim3(1), dim3(512), 0, 0, d, iters);
  return 0;
}
