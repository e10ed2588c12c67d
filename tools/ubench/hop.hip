// Producer -> consumer hand-over of one 64x64 f64 tile (32 KB) between two workgroups on different XCDs.
// variant 0: plain stores + agent release fence ; plain loads after an agent acquire fence   (what k_chol_flow does)
// variant 1: relaxed agent-scope atomic (sc1) stores + s_waitcnt ; relaxed agent-scope atomic (sc1) loads, no fences
//            (NOT sufficient on its own: the flag can overtake a write-through store)
// variant 2: variant 1 + the tile slot is pre-filled with a sentinel and the consumer re-loads every element that still reads as
//            the sentinel (what k_chol_dag does)
// Reports (wall_clock64 ticks = 10 ns): publish = producer start -> flag stored ; seen = flag stored -> consumer saw it ;
// fetch = consumer saw it -> tile in registers.  A third "noise" set of workgroups streams memory to keep the L2s dirty.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ long wc() { return (long)wall_clock64(); }

template <int VAR>
__global__ __launch_bounds__(512) void k_hop(double* tile, int* flags, long* stamps, double* sink, double* noise, long nnoise,
                                             int rounds) {
  const int tid = threadIdx.x;
  if (blockIdx.x >= 2) {  // noise: read-modify-write a private slab so the L2 has dirty lines to write back
    double* p = noise + (long)(blockIdx.x - 2) * nnoise;
    for (int it = 0; it < 64; ++it)
      for (long i = tid; i < nnoise; i += 512) p[i] = p[i] * 1.0000001 + 1.0;
    return;
  }
  for (int r = 0; r < rounds; ++r) {
    double* t = tile + (long)r * 4096;
    int* flag = flags + r * 32;
    if (blockIdx.x == 0) {
      // wait until the consumer is parked on this round's flag (it sets flag+1)
      if (tid == 0) while (__hip_atomic_load(flag + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) __builtin_amdgcn_s_sleep(2);
      __syncthreads();
      long t0 = wc();
      if (VAR == 0) {
        for (int e = tid; e < 4096; e += 512) t[e] = (double)(r * 4096 + e);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      } else {
        for (int e = tid; e < 4096; e += 512)
          __hip_atomic_store(t + e, (double)(r * 4096 + e), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // s_waitcnt vmcnt(0): the write-through stores are acknowledged
      }
      __syncthreads();
      if (tid == 0) {
        __hip_atomic_store(flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        stamps[r * 4 + 0] = t0;
        stamps[r * 4 + 1] = wc();
      }
    } else {
      if (tid == 0) {
        __hip_atomic_store(flag + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) __builtin_amdgcn_s_sleep(1);
        stamps[r * 4 + 2] = wc();
      }
      __syncthreads();
      double v[8];
      if (VAR == 0) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = t[tid + q * 512];
      } else {
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = __hip_atomic_load(t + tid + q * 512, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (VAR == 2) {
#pragma unroll
          for (int q = 0; q < 8; ++q)
            while (__builtin_bit_cast(unsigned long long, v[q]) == 0x7FF4DEADBEEF1234ull)
              v[q] = __hip_atomic_load(t + tid + q * 512, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      double s = 0;
#pragma unroll
      for (int q = 0; q < 8; ++q) s += v[q] - (double)(r * 4096 + tid + q * 512);  // 0 iff every element arrived
      sink[r * 512 + tid] = s;
      __syncthreads();
      if (tid == 0) stamps[r * 4 + 3] = wc();
    }
  }
}

int main(int argc, char** argv) {
  const int rounds = 64, nnoiseWG = argc > 1 ? atoi(argv[1]) : 0;
  const long nnoise = 1 << 16;
  double *tile, *sink, *noise;
  int* flags;
  long* stamps;
  CHECK(hipMalloc(&tile, sizeof(double) * 4096 * rounds));
  CHECK(hipMalloc(&sink, sizeof(double) * 512 * rounds));
  CHECK(hipMalloc(&noise, sizeof(double) * nnoise * (nnoiseWG + 1)));
  CHECK(hipMalloc(&flags, sizeof(int) * 32 * rounds));
  CHECK(hipMalloc(&stamps, sizeof(long) * 4 * rounds));
  CHECK(hipMemset(noise, 0, sizeof(double) * nnoise * (nnoiseWG + 1)));
  for (int var = 0; var < 3; ++var) {
    for (int rep = 0; rep < 2; ++rep) {
      CHECK(hipMemset(flags, 0, sizeof(int) * 32 * rounds));
      CHECK(hipMemset(tile, 0, sizeof(double) * 4096 * rounds));
      if (var == 2) {
        std::vector<unsigned long long> sv(4096 * rounds, 0x7FF4DEADBEEF1234ull);
        CHECK(hipMemcpy(tile, sv.data(), sizeof(double) * 4096 * rounds, hipMemcpyHostToDevice));
      }
      if (var == 2) hipLaunchKernelGGL(k_hop<2>, dim3(2 + nnoiseWG), dim3(512), 0, 0, tile, flags, stamps, sink, noise, nnoise, rounds);
      else if (var == 0) hipLaunchKernelGGL(k_hop<0>, dim3(2 + nnoiseWG), dim3(512), 0, 0, tile, flags, stamps, sink, noise, nnoise, rounds);
      else hipLaunchKernelGGL(k_hop<1>, dim3(2 + nnoiseWG), dim3(512), 0, 0, tile, flags, stamps, sink, noise, nnoise, rounds);
      CHECK(hipDeviceSynchronize());
    }
    std::vector<long> st(4 * rounds);
    std::vector<double> sk(512 * rounds);
    CHECK(hipMemcpy(st.data(), stamps, sizeof(long) * 4 * rounds, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(sk.data(), sink, sizeof(double) * 512 * rounds, hipMemcpyDeviceToHost));
    double bad = 0;
    for (double v : sk) bad += v != 0.0;
    double pub = 0, seen = 0, fetch = 0;
    for (int r = 8; r < rounds; ++r) {
      pub += (st[r * 4 + 1] - st[r * 4 + 0]) * 0.01;
      seen += (st[r * 4 + 2] - st[r * 4 + 1]) * 0.01;
      fetch += (st[r * 4 + 3] - st[r * 4 + 2]) * 0.01;
    }
    const int n = rounds - 8;
    printf("variant %d noiseWG %d: publish %.2f us  flag-seen %.2f us  fetch %.2f us  total %.2f us  wrong elements %.0f\n", var,
           nnoiseWG, pub / n, seen / n, fetch / n, (pub + seen + fetch) / n, bad);
  }
  return 0;
}
