// MFMA ceiling of the chip, varied (round 6): the register-only issue loop behind
// `mfma_sustained` of bench.py (k_mfma_peak, agp_linalg.h: 8 accumulators, one wave per SIMD, 16x16x4) with the number of independent
// accumulators, the waves per SIMD, the instruction shape and the operand values as parameters.  Built as a shared object and driven
// by tools/mfma_ceiling.py, which samples the engine clock and the socket power while a configuration runs.
//     hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -o tools/ubench/libmfma_ceiling.so tools/ubench/mfma_ceiling.hip
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef double d4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

// SHAPE 0: v_mfma_f64_16x16x4 (2048 flop / wave-instruction, 64 cycles)   1: v_mfma_f64_4x4x4_4b (4 blocks: 512 flop, 16 cycles)
//       2: v_mfma_f32_16x16x4 (2048 flop, 32 cycles)                       3: v_mfma_f32_32x32x2 (4096 flop, 64 cycles)
template <int SHAPE, int NACC>
__global__ __launch_bounds__(256) void k_loop(float* out, int iters, int zero) {
  const double ad = zero ? 0.0 : (double)threadIdx.x * 1e-3 + 0.5, bd = zero ? 0.0 : (double)blockIdx.x * 1e-3 + 1.0;
  const float af = (float)ad, bf = (float)bd;
  float s = 0.f;
  if constexpr (SHAPE == 0) {
    d4 c[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) c[i] = d4{(double)i, 1.0, 2.0, 3.0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < NACC; ++i) c[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(ad, bd, c[i], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += (float)(c[i][0] + c[i][1] + c[i][2] + c[i][3]);
  } else if constexpr (SHAPE == 1) {
    double c[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) c[i] = (double)i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < NACC; ++i) c[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(ad, bd, c[i], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += (float)c[i];
  } else if constexpr (SHAPE == 2) {
    f4 c[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) c[i] = f4{(float)i, 1.f, 2.f, 3.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < NACC; ++i) c[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(af, bf, c[i], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
  } else {
    f16v c[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) c[i][r] = (float)(i + r);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < NACC; ++i) c[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(af, bf, c[i], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) s += c[i][r];
  }
  out[blockIdx.x * (int64_t)blockDim.x + threadIdx.x] = s;
}

static float* g_out = nullptr;
static int g_cus = 0;

template <int SHAPE, int NACC>
static double run_one(int waves_per_simd, int iters, int zero, int launches) {
  const int grid = g_cus * waves_per_simd;  // 256 threads = 4 waves = one per SIMD; k workgroups per CU = k waves per SIMD
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((k_loop<SHAPE, NACC>), dim3(grid), dim3(256), 0, 0, g_out, 64, zero);  // warm-up
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  for (int l = 0; l < launches; ++l) hipLaunchKernelGGL((k_loop<SHAPE, NACC>), dim3(grid), dim3(256), 0, 0, g_out, iters, zero);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  hipEventDestroy(e0);
  hipEventDestroy(e1);
  return (double)ms * 1e-3;
}

extern "C" {
// returns seconds for `launches` launches; *flops_out = flops executed
double mfma_ceiling_run(int shape, int nacc, int waves_per_simd, int iters, int zero, int launches, double* flops_out) {
  if (!g_out) {
    hipDeviceProp_t pr;
    hipGetDeviceProperties(&pr, 0);
    g_cus = pr.multiProcessorCount;
    hipMalloc((void**)&g_out, sizeof(float) * (size_t)g_cus * 8 * 256);
  }
  const double fl_per = shape == 0 ? 2048.0 : shape == 1 ? 512.0 : shape == 2 ? 2048.0 : 4096.0;
  *flops_out = fl_per * (double)nacc * (double)iters * 4.0 * (double)g_cus * (double)waves_per_simd * (double)launches;
#define CASE(S, N) \
  if (shape == S && nacc == N) return run_one<S, N>(waves_per_simd, iters, zero, launches);
  CASE(0, 4) CASE(0, 8) CASE(0, 16) CASE(1, 4) CASE(1, 8) CASE(1, 16) CASE(2, 4) CASE(2, 8) CASE(2, 16) CASE(3, 2) CASE(3, 4) CASE(3, 8)
#undef CASE
  return -1.0;
}
int mfma_ceiling_cus() { return g_cus; }
}
