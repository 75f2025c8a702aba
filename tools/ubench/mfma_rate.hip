// Issue rate of the fp32 MFMA shapes on one SIMD of gfx950 (cycles per instruction, one wave and two waves per SIMD):
//   hipcc --offload-arch=gfx950 -O3 -o mfma_rate mfma_rate.hip && ./mfma_rate
// Motivation: the wave-local conv path (fused_step.hip conv_x4f) is built on v_mfma_f32_4x4x1_16b_f32.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE, int CH>
__global__ void k(float* out, long long* cyc, int iters) {
  f32x4 a4[8];
  f32x16 a16[2];
  for (int c = 0; c < 8; ++c) a4[c] = f32x4{0, 0, 0, 0};
  for (int c = 0; c < 2; ++c) for (int e = 0; e < 16; ++e) a16[c][e] = 0.f;
  float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
  int wi = threadIdx.x;
  __syncthreads();
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < (MODE >= 5 ? 0 : 16); ++u) {
      if (MODE == 0) a4[u % CH] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, a4[u % CH], 0, 0, 0);
      if (MODE == 1) a4[u % CH] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, a4[u % CH], 0, 0, 0);
      if (MODE == 2) a16[u % 2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, a16[u % 2], 0, 0, 0);
      if (MODE == 3) {          // 4x4x1 with one int8 -> f32 conversion per MFMA (as the conv paths do)
        const float av = static_cast<float>(static_cast<signed char>(wi >> (8 * (u & 3))));
        a4[u % CH] = __builtin_amdgcn_mfma_f32_4x4x1f32(av, b, a4[u % CH], 0, 0, 0);
        if ((u & 3) == 3) wi = wi * 1664525 + 1013904223;
      }
      if (MODE == 4) a16[u % 2] = __builtin_amdgcn_mfma_f32_16x16x1f32(a, b, a16[u % 2], 0, 0, 0);     // 4 blocks of 16x16x1
    }
    if (MODE == 7 || MODE == 8 || MODE == 9) {   // one conversion per MFMA, interleaved: 7 unsigned byte (v_cvt_f32_ubyteN), 8 the same + one v_add per MFMA, 9 signed byte (sdwa)
      unsigned w4[4] = {(unsigned)wi, (unsigned)wi * 3u, (unsigned)wi * 5u, (unsigned)wi * 7u};
      float sx = 0.f;
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        float av;
        if (MODE == 9) av = static_cast<float>(static_cast<signed char>(w4[u >> 2] >> (8 * (u & 3))));
        else av = static_cast<float>((w4[u >> 2] >> (8 * (u & 3))) & 0xffu);
        if (MODE == 8) sx += b * (u + 1);
        a4[u % CH] = __builtin_amdgcn_mfma_f32_4x4x1f32(av, b, a4[u % CH], 0, 0, 0);
      }
      if (MODE == 8) a4[0][0] += sx;
      wi = wi * 1664525 + 1013904223;
    }
    if (MODE == 5 || MODE == 6) {   // 16 conversions as one block, then 16 MFMAs as one block (5: fenced, 6: left to the scheduler)
      float av[16];
      int w4[4] = {wi, wi * 3, wi * 5, wi * 7};
#pragma unroll
      for (int u = 0; u < 16; ++u) av[u] = static_cast<float>(static_cast<signed char>(w4[u >> 2] >> (8 * (u & 3))));
      if (MODE == 5) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < 16; ++u) a4[u % CH] = __builtin_amdgcn_mfma_f32_4x4x1f32(av[u], b, a4[u % CH], 0, 0, 0);
      if (MODE == 5) __builtin_amdgcn_sched_barrier(0);
      wi = wi * 1664525 + 1013904223;
    }
  }
  long long t1 = clock64();
  float s = 0;
  for (int c = 0; c < 8; ++c) s += a4[c][0] + a4[c][1] + a4[c][2] + a4[c][3];
  for (int c = 0; c < 2; ++c) for (int e = 0; e < 16; ++e) s += a16[c][e];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int MODE, int CH>
void run(const char* name, int threads) {
  float* out; long long* cyc;
  hipMalloc(&out, 4096 * 4); hipMalloc(&cyc, 64 * 8);
  const int iters = 2000;
  hipLaunchKernelGGL((k<MODE, CH>), dim3(1), dim3(threads), 0, 0, out, cyc, iters);
  hipLaunchKernelGGL((k<MODE, CH>), dim3(1), dim3(threads), 0, 0, out, cyc, iters);
  hipDeviceSynchronize();
  long long h[8];
  hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  printf("%-34s %3d threads (%d wave/SIMD): %.2f clock64 ticks per MFMA per wave\n", name, threads, threads / 256 ? threads / 256 : 1, (double)h[0] / (iters * 16.0));
  hipFree(out); hipFree(cyc);
}

int main() {
  for (int th : {64, 256, 512}) {
    run<0, 4>("4x4x1 (16 blocks), 4 chains", th);
    run<0, 8>("4x4x1 (16 blocks), 8 chains", th);
    run<0, 1>("4x4x1 (16 blocks), 1 chain", th);
    run<3, 4>("4x4x1 + cvt per MFMA, 4 chains", th);
    run<5, 4>("4x4x1, 16 cvt then 16 MFMA (fenced)", th);
    run<6, 4>("4x4x1, 16 cvt then 16 MFMA (free)", th);
    run<9, 4>("4x4x1 + signed-byte cvt per MFMA", th);
    run<7, 4>("4x4x1 + unsigned-byte cvt per MFMA", th);
    run<8, 4>("4x4x1 + ubyte cvt + v_fma per MFMA", th);
    run<1, 4>("16x16x4, 4 chains", th);
    run<1, 1>("16x16x4, 1 chain", th);
    run<4, 2>("16x16x1 (4 blocks), 2 chains", th);
    run<2, 2>("32x32x2, 2 chains", th);
  }
  // clock64 tick rate vs wall clock
  return 0;
}
