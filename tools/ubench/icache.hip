// Micro-benchmark (GPU box): what does straight-line code that does not fit the 64 KB instruction cache cost
// on gfx950?  Same dynamic instruction count three ways: fully unrolled (UNROLL x 8 B x 16 instr), a loop over
// a 16 KB body, and a loop over a 1 KB body.  Also: the cost of a workgroup barrier at 8 waves per workgroup.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define BLK16(a, b, c, d)                                   \
  asm volatile("v_fma_f32 %0, %0, %4, %5\n\t"               \
               "v_fma_f32 %1, %1, %4, %5\n\t"               \
               "v_fma_f32 %2, %2, %4, %5\n\t"               \
               "v_fma_f32 %3, %3, %4, %5\n\t"               \
               "v_fma_f32 %0, %0, %4, %5\n\t"               \
               "v_fma_f32 %1, %1, %4, %5\n\t"               \
               "v_fma_f32 %2, %2, %4, %5\n\t"               \
               "v_fma_f32 %3, %3, %4, %5\n\t"               \
               "v_fma_f32 %0, %0, %4, %5\n\t"               \
               "v_fma_f32 %1, %1, %4, %5\n\t"               \
               "v_fma_f32 %2, %2, %4, %5\n\t"               \
               "v_fma_f32 %3, %3, %4, %5\n\t"               \
               "v_fma_f32 %0, %0, %4, %5\n\t"               \
               "v_fma_f32 %1, %1, %4, %5\n\t"               \
               "v_fma_f32 %2, %2, %4, %5\n\t"               \
               "v_fma_f32 %3, %3, %4, %5"                   \
               : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m), "v"(k))

template <int UNROLL, int ITERS>
__global__ __launch_bounds__(512) void fma_kernel(float* out, long long* cyc) {
  float a = threadIdx.x, b = a + 1, c = a + 2, d = a + 3;
  const float m = 0.999f, k = 0.001f;
  const long long t0 = clock64();
#pragma unroll 1
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) BLK16(a, b, c, d);
  }
  const long long t1 = clock64();
  out[blockIdx.x * 512 + threadIdx.x] = a + b + c + d;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NB>
__global__ __launch_bounds__(512) void barrier_kernel(float* out, long long* cyc) {
  __shared__ float s[512];
  float a = threadIdx.x;
  const long long t0 = clock64();
#pragma unroll 1
  for (int it = 0; it < NB; ++it) {
    s[threadIdx.x] = a;
    __syncthreads();
    a += s[(threadIdx.x + 64) & 511];
    __syncthreads();
  }
  const long long t1 = clock64();
  out[blockIdx.x * 512 + threadIdx.x] = a;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <typename F>
static void run(const char* name, F launch, long long ninstr, long long* dcyc) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  launch();
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < 5; ++i) launch();
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<long long> h(256);
  hipMemcpy(h.data(), dcyc, 256 * sizeof(long long), hipMemcpyDeviceToHost);
  long long mx = 0, mn = 1LL << 60;
  for (auto v : h) { mx = v > mx ? v : mx; mn = v < mn ? v : mn; }
  printf("%-28s %8.1f us/launch  clock64 min %lld max %lld  -> %.2f ticks/instr (per wave stream of %lld)\n", name, ms * 1e3 / 5, mn,
         mx, (double)mx / ninstr, ninstr);
}

int main() {
  float* out; long long* cyc;
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8);
  const long long N = 2048LL * 16;   // 32768 instr = 256 KB straight-line
  run("straight 256KB x1", [&] { hipLaunchKernelGGL((fma_kernel<2048, 1>), dim3(256), dim3(512), 0, 0, out, cyc); }, N, cyc);
  run("straight 128KB x2", [&] { hipLaunchKernelGGL((fma_kernel<1024, 2>), dim3(256), dim3(512), 0, 0, out, cyc); }, N, cyc);
  run("straight 64KB x4", [&] { hipLaunchKernelGGL((fma_kernel<512, 4>), dim3(256), dim3(512), 0, 0, out, cyc); }, N, cyc);
  run("loop 32KB x8", [&] { hipLaunchKernelGGL((fma_kernel<256, 8>), dim3(256), dim3(512), 0, 0, out, cyc); }, N, cyc);
  run("loop 16KB x16", [&] { hipLaunchKernelGGL((fma_kernel<128, 16>), dim3(256), dim3(512), 0, 0, out, cyc); }, N, cyc);
  run("loop 1KB x256", [&] { hipLaunchKernelGGL((fma_kernel<8, 256>), dim3(256), dim3(512), 0, 0, out, cyc); }, N, cyc);
  run("straight 256KB x1 (64 thr)", [&] { hipLaunchKernelGGL((fma_kernel<2048, 1>), dim3(256), dim3(64), 0, 0, out, cyc); }, N, cyc);
  run("loop 1KB x256 (64 thr)", [&] { hipLaunchKernelGGL((fma_kernel<8, 256>), dim3(256), dim3(64), 0, 0, out, cyc); }, N, cyc);
  run("2x barrier x1000", [&] { hipLaunchKernelGGL((barrier_kernel<1000>), dim3(256), dim3(512), 0, 0, out, cyc); }, 2000, cyc);
  return 0;
}
