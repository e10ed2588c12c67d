// One-workgroup (512 threads) 64x64x64 f64 product from LDS: variants of the operand fetch schedule, timed with wall_clock64.
#include "../../augmentedgaussianprocesses.jl_amd/csrc/agp_chol.h"
#include <cstdio>
#include <vector>
using namespace agp;

template <typename T>
__device__ __forceinline__ void mma8_v1(const T* As, const T* Bs, Acc8<T>& acc) {  // operands of 8 k-steps in registers first
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wm = wave >> 2, wn = wave & 3;
  const T* pa0 = As + (wm * 32 + (lane & 15)) * LDP + (lane >> 4);
  const T* pa1 = pa0 + 16 * LDP;
  const T* pb0 = Bs + (wn * 16 + (lane & 15)) * LDP + (lane >> 4);
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    T a0[8], a1[8], b0[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      a0[q] = pa0[(h * 8 + q) * 4];
      a1[q] = pa1[(h * 8 + q) * 4];
      b0[q] = pb0[(h * 8 + q) * 4];
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      acc.a[0] = Mfma<T>::mma(a0[q], b0[q], acc.a[0]);
      acc.a[1] = Mfma<T>::mma(a1[q], b0[q], acc.a[1]);
    }
  }
}

template <typename T>
__device__ __forceinline__ void mma8_v2(const T* As, const T* Bs, Acc8<T>& acc) {  // all 16 k-steps in registers, 4 chains
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wm = wave >> 2, wn = wave & 3;
  const T* pa0 = As + (wm * 32 + (lane & 15)) * LDP + (lane >> 4);
  const T* pa1 = pa0 + 16 * LDP;
  const T* pb0 = Bs + (wn * 16 + (lane & 15)) * LDP + (lane >> 4);
  T a0[16], a1[16], b0[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    a0[q] = pa0[q * 4];
    a1[q] = pa1[q * 4];
    b0[q] = pb0[q * 4];
  }
  typename Mfma<T>::acc_t e0 = acc.a[0], e1 = acc.a[1], f0, f1;
#pragma unroll
  for (int r = 0; r < 4; ++r) f0[r] = f1[r] = T(0);
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    e0 = Mfma<T>::mma(a0[q], b0[q], e0);
    e1 = Mfma<T>::mma(a1[q], b0[q], e1);
    f0 = Mfma<T>::mma(a0[8 + q], b0[8 + q], f0);
    f1 = Mfma<T>::mma(a1[8 + q], b0[8 + q], f1);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    acc.a[0][r] = e0[r] + f0[r];
    acc.a[1][r] = e1[r] + f1[r];
  }
}

template <int VAR>
__global__ __launch_bounds__(512) void k_mma(const double* Ag, const double* Bg, double* out, long* ticks, int reps) {
  __shared__ __attribute__((aligned(16))) double sm[2 * TILE * LDP];
  double* bufA = sm;
  double* bufB = sm + TILE * LDP;
  load_tile_lds<double, 512>(Ag, 64, bufA);
  load_tile_lds<double, 512>(Bg, 64, bufB);
  __syncthreads();
  Acc8<double> acc;
  acc.zero();
  long t0 = wall_clock64();
  for (int i = 0; i < reps; ++i) {
    if (VAR == 0) mma8<double>(bufA, bufB, acc);
    if (VAR == 1) mma8_v1<double>(bufA, bufB, acc);
    if (VAR == 2) mma8_v2<double>(bufA, bufB, acc);
    __syncthreads();
  }
  long t1 = wall_clock64();
  if (threadIdx.x == 0) ticks[0] = t1 - t0;
  acc8_foreach<double>(acc, [&](int r, int c, double& v) { out[r * 64 + c] = v; });
}

int main() {
  std::vector<double> a(4096), b(4096), o(4096), ref(4096);
  for (int i = 0; i < 4096; ++i) { a[i] = sin(0.1 * i); b[i] = cos(0.07 * i); }
  for (int r = 0; r < 64; ++r) for (int c = 0; c < 64; ++c) { double s = 0; for (int k = 0; k < 64; ++k) s += a[r * 64 + k] * b[c * 64 + k]; ref[r * 64 + c] = s; }
  double *A, *B, *O; long* tk;
  hipMalloc(&A, 32768); hipMalloc(&B, 32768); hipMalloc(&O, 32768); hipMalloc(&tk, 8);
  hipMemcpy(A, a.data(), 32768, hipMemcpyHostToDevice); hipMemcpy(B, b.data(), 32768, hipMemcpyHostToDevice);
  const int reps = 200;
  for (int var = 0; var < 3; ++var) {
    for (int w = 0; w < 2; ++w) {
      if (var == 0) hipLaunchKernelGGL(k_mma<0>, dim3(1), dim3(512), 0, 0, A, B, O, tk, reps);
      if (var == 1) hipLaunchKernelGGL(k_mma<1>, dim3(1), dim3(512), 0, 0, A, B, O, tk, reps);
      if (var == 2) hipLaunchKernelGGL(k_mma<2>, dim3(1), dim3(512), 0, 0, A, B, O, tk, reps);
      hipDeviceSynchronize();
    }
    long t; hipMemcpy(&t, tk, 8, hipMemcpyDeviceToHost); hipMemcpy(o.data(), O, 32768, hipMemcpyDeviceToHost);
    double err = 0; for (int i = 0; i < 4096; ++i) err = fmax(err, fabs(o[i] / reps - ref[i]));
    printf("variant %d: %.3f us per 64^3 product (+barrier)   max err %.2e\n", var, t * 0.01 / reps, err);
  }
  return 0;
}
