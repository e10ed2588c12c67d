// micro-benchmarks: f64/f32 VALU FMA throughput + dependent latency, v_rcp_f64 accuracy, LDS round trip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
template <typename T, int NCH>
__global__ void k_fma(T* out, int iters, T b, T c) {
  T x[NCH];
  for (int i = 0; i < NCH; ++i) x[i] = T(threadIdx.x + i) * T(1e-3);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NCH; ++i) x[i] = fma(x[i], b, c);
  }
  T s = 0;
  for (int i = 0; i < NCH; ++i) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_rcp(const double* p, double* r0, double* r1, double* r2, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double x = p[i];
  double r = __builtin_amdgcn_rcp(x);
  r0[i] = r;
  double e = fma(-x, r, 1.0); r = fma(r, e, r); r1[i] = r;
  e = fma(-x, r, 1.0); r = fma(r, e, r); r2[i] = r;
}
// LDS write -> barrier -> read dependent chain, 256 threads
__global__ void k_ldschain(double* out, int iters) {
  __shared__ double buf[2][256];
  double v = threadIdx.x;
  for (int it = 0; it < iters; ++it) {
    buf[it & 1][threadIdx.x] = v;
    __syncthreads();
    v = buf[it & 1][(threadIdx.x + 17) & 255] + 1.0;
  }
  out[threadIdx.x] = v;
}
template <typename F> float timeit(F f) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  f(); hipDeviceSynchronize();
  hipEventRecord(a); f(); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); return ms;
}
int main() {
  double* d; hipMalloc(&d, 1 << 24);
  const int iters = 20000;
  // throughput: full chip
  { float ms = timeit([&] { hipLaunchKernelGGL((k_fma<double, 8>), dim3(256 * 8), dim3(256), 0, 0, d, iters, 1.0000001, 1e-9); });
    double fl = 256.0 * 8 * 256 * iters * 8 * 2; printf("f64 VALU fma throughput: %.1f TFLOP/s\n", fl / ms / 1e9); }
  { float ms = timeit([&] { hipLaunchKernelGGL((k_fma<float, 8>), dim3(256 * 8), dim3(256), 0, 0, (float*)d, iters, 1.0000001f, 1e-9f); });
    double fl = 256.0 * 8 * 256 * iters * 8 * 2; printf("f32 VALU fma throughput: %.1f TFLOP/s\n", fl / ms / 1e9); }
  // single wave: issue rate (8 independent chains) and latency (1 chain)
  { float ms = timeit([&] { hipLaunchKernelGGL((k_fma<double, 8>), dim3(1), dim3(64), 0, 0, d, iters, 1.0000001, 1e-9); });
    printf("f64 fma, 1 wave, 8 indep chains: %.1f ns per fma\n", ms * 1e6 / (iters * 8.0)); }
  { float ms = timeit([&] { hipLaunchKernelGGL((k_fma<double, 1>), dim3(1), dim3(64), 0, 0, d, iters, 1.0000001, 1e-9); });
    printf("f64 fma, 1 wave, dependent chain: %.1f ns per fma\n", ms * 1e6 / (iters * 1.0)); }
  { float ms = timeit([&] { hipLaunchKernelGGL((k_fma<float, 8>), dim3(1), dim3(64), 0, 0, (float*)d, iters, 1.0000001f, 1e-9f); });
    printf("f32 fma, 1 wave, 8 indep chains: %.1f ns per fma\n", ms * 1e6 / (iters * 8.0)); }
  { float ms = timeit([&] { hipLaunchKernelGGL((k_fma<float, 1>), dim3(1), dim3(64), 0, 0, (float*)d, iters, 1.0000001f, 1e-9f); });
    printf("f32 fma, 1 wave, dependent chain: %.1f ns per fma\n", ms * 1e6 / (iters * 1.0)); }
  { float ms = timeit([&] { hipLaunchKernelGGL(k_ldschain, dim3(1), dim3(256), 0, 0, d, iters); });
    printf("LDS write->barrier->read chain (256 thr): %.1f ns per hop\n", ms * 1e6 / iters); }
  // rcp accuracy
  { const int n = 4096; double hp[n]; for (int i = 0; i < n; ++i) hp[i] = 0.5 + 3.0 * i / n + 1e-3 * sin(i);
    double *p, *r0, *r1, *r2; hipMalloc(&p, n * 8); hipMalloc(&r0, n * 8); hipMalloc(&r1, n * 8); hipMalloc(&r2, n * 8);
    hipMemcpy(p, hp, n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_rcp, dim3(n / 256), dim3(256), 0, 0, p, r0, r1, r2, n);
    double h0[n], h1[n], h2[n]; hipMemcpy(h0, r0, n * 8, hipMemcpyDeviceToHost); hipMemcpy(h1, r1, n * 8, hipMemcpyDeviceToHost); hipMemcpy(h2, r2, n * 8, hipMemcpyDeviceToHost);
    double e0 = 0, e1 = 0, e2 = 0; for (int i = 0; i < n; ++i) { double t = 1.0 / hp[i]; e0 = fmax(e0, fabs(h0[i] - t) / t); e1 = fmax(e1, fabs(h1[i] - t) / t); e2 = fmax(e2, fabs(h2[i] - t) / t); }
    printf("v_rcp_f64 rel err: raw %.3e, 1 NR %.3e, 2 NR %.3e\n", e0, e1, e2); }
  return 0;
}
