// v_mfma_f64_4x4x4_4b_f64 on gfx950: operand / result layout and throughput with the A-broadcast modifiers (round 6).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o mfma4 mfma4.hip ; ./mfma4
// Why: the register-only issue loops of tools/mfma_ceiling.py show v_mfma_f64_16x16x4 at 47-48 TF (0.61 of the 78.6 TF datasheet
// peak, with ZERO operands at 585 W as well: not a power limit) and the 4-block 4x4x4 form at 73-75 TF (0.95).  With CBSZ = 2 one A
// block (ABID) is broadcast to all four blocks, so four 4x4x4 instructions with ABID = 0..3 cover a 16 x 16 x 4 product from the same
// two operand registers.  This program (1) decodes which (A lane, B lane) pairs reach which result lane for CBSZ / ABID = 0/0 and
// 2/0..3, (2) times the broadcast form.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int CBSZ, int ABID>
__global__ void k_decode(double* out) {  // out[la][lb][lane]
  const int lane = threadIdx.x;
  for (int la = 0; la < 64; ++la)
    for (int lb = 0; lb < 64; ++lb) {
      const double a = lane == la ? 1.0 : 0.0, b = lane == lb ? 1.0 : 0.0;
      const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, CBSZ, ABID, 0);
      out[((size_t)la * 64 + lb) * 64 + lane] = d;
    }
}

template <int CBSZ>
__global__ __launch_bounds__(256) void k_rate(double* out, int iters) {
  double c[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) c[i] = (double)i;
  const double a = (double)threadIdx.x * 1e-3 + 0.5, b = (double)blockIdx.x * 1e-3 + 1.0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      c[4 * g + 0] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c[4 * g + 0], CBSZ, 0, 0);
      c[4 * g + 1] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c[4 * g + 1], CBSZ, CBSZ ? 1 : 0, 0);
      c[4 * g + 2] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c[4 * g + 2], CBSZ, CBSZ ? 2 : 0, 0);
      c[4 * g + 3] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c[4 * g + 3], CBSZ, CBSZ ? 3 : 0, 0);
    }
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += c[i];
  out[blockIdx.x * (size_t)blockDim.x + threadIdx.x] = s;
}
// the 16x16x4 form with 4 accumulators (= the same 16 x 16 x 4 x 4 flops per group)
__global__ __launch_bounds__(256) void k_rate16(double* out, int iters) {
  typedef double d4 __attribute__((ext_vector_type(4)));
  d4 c[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) c[i] = d4{(double)i, 1.0, 2.0, 3.0};
  const double a = (double)threadIdx.x * 1e-3 + 0.5, b = (double)blockIdx.x * 1e-3 + 1.0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int g = 0; g < 4; ++g) c[g] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c[g], 0, 0, 0);
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
  out[blockIdx.x * (size_t)blockDim.x + threadIdx.x] = s;
}

template <int CBSZ, int ABID>
static void decode(double* dev, const char* name) {
  hipLaunchKernelGGL((k_decode<CBSZ, ABID>), dim3(1), dim3(64), 0, 0, dev);
  std::vector<double> h((size_t)64 * 64 * 64);
  hipMemcpy(h.data(), dev, h.size() * sizeof(double), hipMemcpyDeviceToHost);
  // for every result lane: the list of (la, lb) pairs that feed it
  printf("== %s: result lane <- sum over k of A[la_k] * B[lb_k]\n", name);
  for (int lane = 0; lane < 64; ++lane) {
    printf("  D[%2d] <-", lane);
    for (int la = 0; la < 64; ++la)
      for (int lb = 0; lb < 64; ++lb)
        if (h[((size_t)la * 64 + lb) * 64 + lane] != 0.0) printf(" (%d,%d)", la, lb);
    printf("\n");
  }
}

int main() {
  double* dev;
  hipMalloc((void**)&dev, sizeof(double) * 64 * 64 * 64);
  decode<0, 0>(dev, "cbsz=0 abid=0");
  decode<2, 0>(dev, "cbsz=2 abid=0");
  decode<2, 1>(dev, "cbsz=2 abid=1");
  decode<2, 3>(dev, "cbsz=2 abid=3");
  hipDeviceProp_t pr;
  hipGetDeviceProperties(&pr, 0);
  const int cus = pr.multiProcessorCount;
  double* out;
  hipMalloc((void**)&out, sizeof(double) * cus * 4 * 256);
  for (int wps = 1; wps <= 4; wps *= 2) {
    for (int form = 0; form < 3; ++form) {
      const int iters = 20000 / wps;
      hipEvent_t e0, e1;
      hipEventCreate(&e0);
      hipEventCreate(&e1);
      auto launch = [&]() {
        if (form == 0) hipLaunchKernelGGL((k_rate<0>), dim3(cus * wps), dim3(256), 0, 0, out, iters);
        else if (form == 1) hipLaunchKernelGGL((k_rate<2>), dim3(cus * wps), dim3(256), 0, 0, out, iters);
        else hipLaunchKernelGGL(k_rate16, dim3(cus * wps), dim3(256), 0, 0, out, iters);
      };
      launch();
      hipDeviceSynchronize();
      hipEventRecord(e0, 0);
      for (int r = 0; r < 20; ++r) launch();
      hipEventRecord(e1, 0);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      const double flops = (form == 2 ? 4.0 * 2048.0 : 16.0 * 512.0) * iters * 4.0 * cus * wps * 20.0;
      printf("rate %-28s waves/SIMD %d : %.1f TF\n", form == 0 ? "4x4x4 4b cbsz=0" : form == 1 ? "4x4x4 4b cbsz=2 abid=0..3" : "16x16x4 (4 acc)", wps,
             flops / (ms * 1e-3) / 1e12);
    }
  }
  return 0;
}
