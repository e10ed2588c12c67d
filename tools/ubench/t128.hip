// fp32 128 x 128 tile kernels (agp_tile128.h) against the 64 x 64 ones: correctness on sampled entries (double reference on the
// host) and time per launch.   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o t128 t128.hip ; ./t128 [n] [K] [G]
#include "agp_tile128.h"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
using namespace agp;

#define CK(x)                                                                     \
  do {                                                                            \
    hipError_t e_ = (x);                                                          \
    if (e_ != hipSuccess) {                                                       \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      exit(1);                                                                    \
    }                                                                             \
  } while (0)

template <typename F>
static double time_us(F f, int reps = 20) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  for (int i = 0; i < 3; ++i) f();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a, 0));
  for (int i = 0; i < reps; ++i) f();
  CK(hipEventRecord(b, 0));
  CK(hipEventSynchronize(b));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, a, b));
  return ms * 1e3 / reps;
}

int main(int argc, char** argv) {
  const int64_t n = argc > 1 ? atoll(argv[1]) : 2048, K = argc > 2 ? atoll(argv[2]) : 2048;
  int64_t G = argc > 3 ? atoll(argv[3]) : 256;
  std::mt19937 rng(1);
  std::uniform_real_distribution<float> U(-1.f, 1.f);
  std::vector<float> hA((size_t)K * n), hw(K), he2((size_t)n * n), hKi((size_t)n * n), hr(K), he1(n), hB((size_t)n * K);
  for (auto& x : hA) x = U(rng);
  for (auto& x : hB) x = U(rng);
  for (auto& x : hw) x = 0.5f + 0.5f * U(rng) * U(rng);
  for (auto& x : hr) x = U(rng);
  for (auto& x : he1) x = U(rng);
  for (int64_t i = 0; i < n; ++i)
    for (int64_t j = 0; j <= i; ++j) {
      he2[i * n + j] = he2[j * n + i] = U(rng);
      hKi[i * n + j] = hKi[j * n + i] = U(rng);
    }
  float *A, *w, *e2, *Ki, *out, *e2b, *outb, *ws, *r, *e1, *e1b, *B, *C, *Cb, *P0, *P0b, *P1, *P1b;
  const int64_t nt = n / T128, ntiles = nt * (nt + 1) / 2, nslab = K / BK128, per = sk_per(ntiles * nslab, G), nslots = sk_slots(nslab, per);
  const size_t wsn = (size_t)ntiles * nslots * T128 * T128;
  CK(hipMalloc(&A, sizeof(float) * K * n));
  CK(hipMalloc(&B, sizeof(float) * n * K));
  CK(hipMalloc(&w, sizeof(float) * K));
  CK(hipMalloc(&r, sizeof(float) * K));
  CK(hipMalloc(&e1, sizeof(float) * n));
  CK(hipMalloc(&e1b, sizeof(float) * n));
  CK(hipMalloc(&e2, sizeof(float) * n * n));
  CK(hipMalloc(&e2b, sizeof(float) * n * n));
  CK(hipMalloc(&Ki, sizeof(float) * n * n));
  CK(hipMalloc(&out, sizeof(float) * n * n));
  CK(hipMalloc(&outb, sizeof(float) * n * n));
  CK(hipMalloc(&C, sizeof(float) * n * n));
  CK(hipMalloc(&Cb, sizeof(float) * n * n));
  CK(hipMalloc(&P1, sizeof(float) * n * n));
  CK(hipMalloc(&P1b, sizeof(float) * n * n));
  const int64_t ldp = n, nsl = n / 32;
  CK(hipMalloc(&P0, sizeof(float) * nsl * ldp));
  CK(hipMalloc(&P0b, sizeof(float) * nsl * ldp));
  CK(hipMalloc(&ws, sizeof(float) * wsn));
  CK(hipMemcpy(A, hA.data(), sizeof(float) * K * n, hipMemcpyHostToDevice));
  CK(hipMemcpy(B, hB.data(), sizeof(float) * n * K, hipMemcpyHostToDevice));
  CK(hipMemcpy(w, hw.data(), sizeof(float) * K, hipMemcpyHostToDevice));
  CK(hipMemcpy(r, hr.data(), sizeof(float) * K, hipMemcpyHostToDevice));
  CK(hipMemcpy(Ki, hKi.data(), sizeof(float) * n * n, hipMemcpyHostToDevice));
  hipLaunchKernelGGL((k_fill_sent<float>), dim3(256, 1), dim3(256), 0, 0, ws, (int64_t)wsn, (int64_t)0);
  const float lr = 0.3f;
  const int64_t nrider = n / TILE;
  auto reset = [&]() {
    CK(hipMemcpy(e2, he2.data(), sizeof(float) * n * n, hipMemcpyHostToDevice));
    CK(hipMemcpy(e2b, he2.data(), sizeof(float) * n * n, hipMemcpyHostToDevice));
    CK(hipMemcpy(e1, he1.data(), sizeof(float) * n, hipMemcpyHostToDevice));
    CK(hipMemcpy(e1b, he1.data(), sizeof(float) * n, hipMemcpyHostToDevice));
  };
  auto run128 = [&]() {
    hipLaunchKernelGGL((k_syrk128_tn<SY_ETA2>), dim3((unsigned)G), dim3(NT128), 0, 0, A, n, K, w, out, n, e2, Ki, n, lr, ntiles, n, ws, r, e1,
                       (const float*)nullptr, (float*)nullptr, (int64_t)0, (int64_t)0, 0);
  };
  const int64_t nt64 = n / TILE, tiles64 = nt64 * (nt64 + 1) / 2;
  auto run64 = [&]() {
    hipLaunchKernelGGL((k_syrk_tn<float, SY_ETA2, 2>), dim3((unsigned)(tiles64 + nrider)), dim3(2 * NTHREADS), 0, 0, A, n, K, w, 0, outb, n,
                       e2b, Ki, n, lr, tiles64, r, e1b, (const float*)nullptr, nrider, (float*)nullptr, (int64_t)0, (int64_t)0, 0);
  };
  printf("syrk: n = %lld K = %lld, 128-tiles %lld x %lld slabs over G = %lld ranges of %lld (slots per tile %lld, ws %.1f MB)\n", (long long)n,
         (long long)K, (long long)ntiles, (long long)nslab, (long long)G, (long long)per, (long long)nslots, wsn * 4 / 1e6);
  reset();
  run128();
  run64();
  CK(hipDeviceSynchronize());
  std::vector<float> o1((size_t)n * n), o2((size_t)n * n), q1((size_t)n * n), q2((size_t)n * n), v1(n), v2(n);
  CK(hipMemcpy(o1.data(), out, sizeof(float) * n * n, hipMemcpyDeviceToHost));
  CK(hipMemcpy(o2.data(), outb, sizeof(float) * n * n, hipMemcpyDeviceToHost));
  CK(hipMemcpy(q1.data(), e2, sizeof(float) * n * n, hipMemcpyDeviceToHost));
  CK(hipMemcpy(q2.data(), e2b, sizeof(float) * n * n, hipMemcpyDeviceToHost));
  CK(hipMemcpy(v1.data(), e1, sizeof(float) * n, hipMemcpyDeviceToHost));
  CK(hipMemcpy(v2.data(), e1b, sizeof(float) * n, hipMemcpyDeviceToHost));
  double emax = 0, emax64 = 0, sym = 0, scale = 0, e1d = 0;
  std::uniform_int_distribution<int64_t> I(0, n - 1);
  for (int s = 0; s < 4000; ++s) {
    int64_t i = I(rng), j = I(rng);
    if (s < 200) j = i;  // diagonal entries too
    const int64_t a = std::max(i, j), b = std::min(i, j);
    double S = 0;
    for (int64_t k = 0; k < K; ++k) S += (double)hw[k] * hA[k * n + a] * hA[k * n + b];
    const double e0 = he2[a * n + b], ref = e0 + lr * (-(S + 0.5 * hKi[a * n + b]) - e0);
    emax = std::max(emax, std::abs(q1[i * n + j] - ref));
    emax = std::max(emax, std::abs(o1[i * n + j] + 2 * ref));
    emax64 = std::max(emax64, std::abs(q2[i * n + j] - ref));
    scale = std::max(scale, std::abs(ref));
  }
  for (int64_t i = 0; i < n; ++i)
    for (int64_t j = 0; j < i; ++j) sym = std::max(sym, (double)std::abs(q1[i * n + j] - q1[j * n + i]) + std::abs(o1[i * n + j] - o1[j * n + i]));
  for (int64_t i = 0; i < n; ++i) e1d = std::max(e1d, (double)std::abs(v1[i] - v2[i]));
  printf("  eta2 step: max err 128-tiles %.3e, 64-tiles %.3e (scale %.2f); asymmetry %.1e; eta1 riders differ by %.1e\n", emax, emax64, scale, sym, e1d);
  // sentinels back in place?
  {
    std::vector<unsigned int> hs(wsn);
    CK(hipMemcpy(hs.data(), ws, sizeof(float) * wsn, hipMemcpyDeviceToHost));
    size_t bad = 0;
    for (auto x : hs) bad += x != Sent<float>::bits;
    printf("  workspace elements not holding the sentinel after the launch: %zu\n", bad);
  }
  // reproducibility
  {
    reset();
    run128();
    CK(hipDeviceSynchronize());
    std::vector<float> q3((size_t)n * n);
    CK(hipMemcpy(q3.data(), e2, sizeof(float) * n * n, hipMemcpyDeviceToHost));
    size_t diff = 0;
    for (size_t i = 0; i < q3.size(); ++i) diff += __builtin_bit_cast(unsigned, q3[i]) != __builtin_bit_cast(unsigned, q1[i]);
    printf("  second launch differs in %zu elements\n", diff);
  }
  const double fl = (double)n * n * K;  // executed as credited: one triangle
  const double t128 = time_us(run128), t64 = time_us(run64);
  printf("  k_syrk128_tn<SY_ETA2>: %.1f us = %.1f TF (B m^2)   |   k_syrk_tn<float, SY_ETA2, 2>: %.1f us = %.1f TF\n", t128, fl / t128 / 1e6,
         t64, fl / t64 / 1e6);
  // ---- kappa GEMM: C = A2 B', row-dot with E = A2
  const int64_t gx = n / T128, gy = n / T128;
  auto g128 = [&]() {
    hipLaunchKernelGGL((k_gemm128_nt<EPI_KAPPA>), dim3((unsigned)(gx * gy)), dim3(NT128), 0, 0, (const float*)B, K, (const float*)A, K, K, C, n,
                       (const float*)B, K, P0, P1, ldp, gx, gy);
  };
  auto g64 = [&]() {
    hipLaunchKernelGGL((k_gemm_nt<float, EPI_KAPPA, 2>), dim3((unsigned)(n / TILE), (unsigned)(n / TILE)), dim3(2 * NTHREADS), 0, 0,
                       (const float*)B, K, (const float*)A, K, K, 0, Cb, n, (const float*)B, K, (const float*)nullptr, P0b, P1b, ldp);
  };
  if (n == K) {
    g128();
    g64();
    CK(hipDeviceSynchronize());
    std::vector<float> c1((size_t)n * n), c2((size_t)n * n), p1(nsl * ldp), p2(nsl * ldp), w1((size_t)n * n);
    CK(hipMemcpy(c1.data(), C, sizeof(float) * n * n, hipMemcpyDeviceToHost));
    CK(hipMemcpy(c2.data(), Cb, sizeof(float) * n * n, hipMemcpyDeviceToHost));
    CK(hipMemcpy(w1.data(), P1, sizeof(float) * n * n, hipMemcpyDeviceToHost));
    CK(hipMemcpy(p1.data(), P0, sizeof(float) * nsl * ldp, hipMemcpyDeviceToHost));
    CK(hipMemcpy(p2.data(), P0b, sizeof(float) * nsl * ldp, hipMemcpyDeviceToHost));
    double ge = 0, ge64 = 0, pe = 0, we = 0;
    for (int s = 0; s < 4000; ++s) {
      const int64_t i = I(rng), j = I(rng);
      double S = 0;
      for (int64_t k = 0; k < K; ++k) S += (double)hB[i * K + k] * hA[j * K + k];
      ge = std::max(ge, std::abs(c1[i * n + j] - S));
      ge64 = std::max(ge64, std::abs(c2[i * n + j] - S));
      we = std::max(we, (double)std::abs(c1[i * n + j] - w1[i * n + j]));
    }
    for (int64_t i = 0; i < n; i += 37) {  // row-dot: sum over slices against the 64-tile kernel's
      double s1 = 0, s2 = 0;
      for (int64_t q = 0; q < nsl; ++q) s1 += p1[q * ldp + i], s2 += p2[q * ldp + i];
      pe = std::max(pe, std::abs(s1 - s2) / std::max(1.0, std::abs(s2)));
    }
    const double tg128 = time_us(g128), tg64 = time_us(g64), gfl = 2.0 * n * n * K;
    printf("gemm (kappa epilogue): max err 128-tiles %.3e, 64-tiles %.3e; copy differs %.1e; row-dot rel diff %.2e\n", ge, ge64, we, pe);
    printf("  k_gemm128_nt<EPI_KAPPA>: %.1f us = %.1f TF   |   k_gemm_nt<float, EPI_KAPPA, 2>: %.1f us = %.1f TF\n", tg128, gfl / tg128 / 1e6, tg64,
           gfl / tg64 / 1e6);
  }
  return 0;
}
