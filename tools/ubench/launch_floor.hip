// Start-to-start time of dependent launches on one HIP stream (the floor under the block mode's ~200 small-layer launches
// per chunk):   hipcc --offload-arch=gfx950 -O3 -o launch_floor launch_floor.hip && ./launch_floor
//   empty kernel, 1 wave / 256 x 1 wave / 1024 x 4 waves;  a kernel with a chain of `hops` dependent L2 round trips.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <string>
#pragma clang diagnostic ignored "-Wunused-value"

__global__ void empty_kernel(int* p) { if (p && threadIdx.x == 9999) p[0] = 1; }
__global__ void chase_kernel(const int* __restrict__ next, int* out, int hops) {
  int i = blockIdx.x & 1023;
  for (int h = 0; h < hops; ++h) i = next[i];
  if (i == -1) out[0] = i;
}

template <class F>
static void run(const char* what, int n, F launch) {
  for (int i = 0; i < 20; ++i) launch();
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto h0 = std::chrono::steady_clock::now();
  hipEventRecord(e0, 0);
  for (int i = 0; i < n; ++i) launch();
  hipEventRecord(e1, 0);
  auto h1 = std::chrono::steady_clock::now();
  hipDeviceSynchronize();
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  printf("%-58s %6.2f us per launch on the device, %5.2f us of host time per launch\n", what, 1e3 * ms / n,
         std::chrono::duration<double, std::micro>(h1 - h0).count() / n);
}

int main() {
  int *next, *out;
  hipMalloc(&next, 1024 * sizeof(int)); hipMalloc(&out, sizeof(int));
  int h[1024];
  for (int i = 0; i < 1024; ++i) h[i] = (i * 37 + 11) & 1023;
  hipMemcpy(next, h, sizeof(h), hipMemcpyHostToDevice);
  const int n = 2000;
  run("empty kernel, 1 workgroup of 64", n, [&] { hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, 0, nullptr); });
  run("empty kernel, 256 workgroups of 64", n, [&] { hipLaunchKernelGGL(empty_kernel, dim3(256), dim3(64), 0, 0, nullptr); });
  run("empty kernel, 1024 workgroups of 256", n, [&] { hipLaunchKernelGGL(empty_kernel, dim3(1024), dim3(256), 0, 0, nullptr); });
  run("empty kernel, 256 workgroups of 64, 48 KB of dynamic LDS", n, [&] { hipLaunchKernelGGL(empty_kernel, dim3(256), dim3(64), 48 * 1024, 0, nullptr); });
  for (int hops : {1, 2, 4, 8})
    run((std::string("256 workgroups of 64, ") + std::to_string(hops) + " dependent L2 round trips").c_str(), n,
        [&] { hipLaunchKernelGGL(chase_kernel, dim3(256), dim3(64), 0, 0, next, out, hops); });
  return 0;
}
