// Micro-benchmark (GPU box): how fast can every CU stream the SAME weight blob (all 256 workgroups read the same
// addresses in lock step, like the frame-step kernel does), as a function of blob size (fits the 4 MB L2 of an XCD
// or not), of the loads each wave keeps in flight, and of concurrent private streaming traffic (the state tensors).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int D, bool STATE, bool NT>
__global__ __launch_bounds__(512) void wstream(const f32x4* __restrict__ w, int n_frag, f32x4* __restrict__ state, int state_f4, float* out, long long* cyc) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f32x4 acc = {0, 0, 0, 0};
  f32x4* st = state + (size_t)blockIdx.x * state_f4;
  const long long t0 = wall_clock64();
  // fragment f (1 KB) is read by wave f % 8; D loads in flight per wave
  for (int f0 = wave; f0 < n_frag; f0 += 8 * D) {
    f32x4 r[D];
#pragma unroll
    for (int d = 0; d < D; ++d) {
      const int f = f0 + 8 * d;
      r[d] = f < n_frag ? w[(size_t)f * 64 + lane] : f32x4{0, 0, 0, 0};
    }
    if (STATE) {   // private streaming traffic: read + write 1 KB per wave per D weight fragments * ratio
      const int i = (f0 / 8) % (state_f4 / 512);
      f32x4 s;
      if (NT) s = __builtin_nontemporal_load(&st[(size_t)i * 512 + threadIdx.x]);
      else s = st[(size_t)i * 512 + threadIdx.x];
      s += 1.0f;
      if (NT) __builtin_nontemporal_store(s, &st[(size_t)i * 512 + threadIdx.x]);
      else st[(size_t)i * 512 + threadIdx.x] = s;
    }
#pragma unroll
    for (int d = 0; d < D; ++d) acc += r[d];
  }
  const long long t1 = wall_clock64();
  out[blockIdx.x * 512 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// dependent chain: latency of one 1 KB wave load
__global__ __launch_bounds__(512) void wlat(const f32x4* __restrict__ w, int n_frag, float* out, long long* cyc) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int f = wave;
  float acc = 0;
  const long long t0 = wall_clock64();
  int n = 0;
  while (f < n_frag) {
    const f32x4 r = w[(size_t)f * 64 + lane];
    acc += r[0];
    f += 8 + (acc > 1e30f ? 1 : 0);     // data-dependent next address
    ++n;
  }
  const long long t1 = wall_clock64();
  out[blockIdx.x * 512 + threadIdx.x] = acc;
  if (threadIdx.x == 0) { cyc[blockIdx.x] = t1 - t0; cyc[256 + blockIdx.x] = n; }
}

int main() {
  const size_t WMAX = 16u << 20;
  f32x4 *w, *state; float* out; long long* cyc;
  hipMalloc(&w, WMAX); hipMemset(w, 0, WMAX);
  const int state_f4 = (1600 * 1024) / 16;     // 1.6 MB per workgroup
  hipMalloc(&state, (size_t)256 * state_f4 * 16); hipMemset(state, 0, (size_t)256 * state_f4 * 16);
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 512 * 8);
  int khz = 100000; hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, 0);
  std::vector<long long> h(512);
  auto report = [&](const char* name, size_t bytes, int reps) {
    hipDeviceSynchronize();
    hipMemcpy(h.data(), cyc, 512 * 8, hipMemcpyDeviceToHost);
    double mx = 0; for (int i = 0; i < 256; ++i) mx = h[i] > mx ? h[i] : mx;
    const double us = mx * 1000.0 / khz;
    printf("%-44s blob %5.1f MB: %8.1f us/pass  -> %6.1f GB/s per CU\n", name, bytes / 1048576.0, us, bytes / us / 1e3);
  };
  for (size_t bytes : {size_t(2) << 20, size_t(3) << 20, size_t(6) << 20, (size_t(11) << 20) + (size_t(1) << 19)}) {
    const int nf = bytes / 1024;
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(wlat, dim3(256), dim3(512), 0, 0, w, nf, out, cyc);
    hipDeviceSynchronize();
    hipMemcpy(h.data(), cyc, 512 * 8, hipMemcpyDeviceToHost);
    double mx = 0; for (int i = 0; i < 256; ++i) mx = h[i] > mx ? h[i] : mx;
    printf("dependent 1 KB wave loads, blob %5.1f MB: %.0f ns per load (%lld loads per wave)\n", bytes / 1048576.0, mx * 1e6 / khz / h[256], h[256]);
#define RUN(D, S, NT, label) for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((wstream<D, S, NT>), dim3(256), dim3(512), 0, 0, w, nf, state, state_f4, out, cyc); report(label, bytes, 3);
    RUN(1, false, false, "stream D=1 in flight per wave")
    RUN(4, false, false, "stream D=4")
    RUN(8, false, false, "stream D=8")
    RUN(16, false, false, "stream D=16")
    RUN(8, true, false, "stream D=8 + private state traffic")
    RUN(8, true, true, "stream D=8 + private state traffic (nt)")
    RUN(16, true, true, "stream D=16 + private state traffic (nt)")
  }
  return 0;
}
