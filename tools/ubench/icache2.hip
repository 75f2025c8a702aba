// Micro-benchmark (GPU box): cost of COLD code in the regime of the fused kernel -- ~150 short "ops", each a few hundred
// instructions with skipped (exec-masked) regions and a workgroup barrier at the end, every op its own code (450 KB in
// total, never re-executed) -- against the same work as a loop over one op body (instruction cache warm).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define FMA16(a, b, c, d)                                                                                                   \
  asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n"   \
               "v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n"   \
               "v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n"   \
               "v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5"     \
               : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m), "v"(k))

// one "op": 4 x 16 FMAs, a region only lane 777 would run (skipped: s_cbranch_execz over 8 x 16 FMAs), 4 x 16 FMAs, barrier
#define OP_BODY                                                          \
  FMA16(a, b, c, d); FMA16(a, b, c, d); FMA16(a, b, c, d); FMA16(a, b, c, d);   \
  if (threadIdx.x == never) {                                            \
    FMA16(a, b, c, d); FMA16(a, b, c, d); FMA16(a, b, c, d); FMA16(a, b, c, d); \
    FMA16(a, b, c, d); FMA16(a, b, c, d); FMA16(a, b, c, d); FMA16(a, b, c, d); \
  }                                                                      \
  FMA16(a, b, c, d); FMA16(a, b, c, d); FMA16(a, b, c, d); FMA16(a, b, c, d);   \
  __syncthreads();

template <int UNROLL, int ITERS>
__global__ __launch_bounds__(512) void ops_kernel(float* out, long long* cyc, int never) {
  float a = threadIdx.x, b = a + 1, c = a + 2, d = a + 3;
  const float m = 0.999f, k = 0.001f;
  const long long t0 = clock64();
#pragma unroll 1
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) { OP_BODY }
  }
  const long long t1 = clock64();
  out[blockIdx.x * 512 + threadIdx.x] = a + b + c + d;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <typename F>
static void run(const char* name, F launch, int nops, long long* dcyc) {
  launch(); hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  for (int i = 0; i < 5; ++i) launch();
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  std::vector<long long> h(256);
  hipMemcpy(h.data(), dcyc, 256 * 8, hipMemcpyDeviceToHost);
  long long mx = 0; for (auto v : h) mx = v > mx ? v : mx;
  printf("%-34s %8.1f us/launch = %6.3f us per op   (%lld ticks per op)\n", name, ms * 1e3 / 5, ms * 1e3 / 5 / nops, mx / nops);
}

int main() {
  float* out; long long* cyc;
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8);
  // an op = 128 executed + 128 skipped FMAs (8 B each) + barrier  ~ 2.1 KB of code; 256 ops = 540 KB
  run("256 cold ops (540 KB, unrolled)", [&] { hipLaunchKernelGGL((ops_kernel<256, 1>), dim3(256), dim3(512), 0, 0, out, cyc, 777); }, 256, cyc);
  run("64 cold ops x 4", [&] { hipLaunchKernelGGL((ops_kernel<64, 4>), dim3(256), dim3(512), 0, 0, out, cyc, 777); }, 256, cyc);
  run("16 ops x 16 (34 KB: warm)", [&] { hipLaunchKernelGGL((ops_kernel<16, 16>), dim3(256), dim3(512), 0, 0, out, cyc, 777); }, 256, cyc);
  run("1 op x 256 (warm)", [&] { hipLaunchKernelGGL((ops_kernel<1, 256>), dim3(256), dim3(512), 0, 0, out, cyc, 777); }, 256, cyc);
  return 0;
}
