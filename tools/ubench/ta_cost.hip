// Micro-benchmark (GPU box): what does a vector memory LOAD INSTRUCTION cost the CU's memory pipe, as a function of how many of its
// lanes fetch something useful?  The fused frame-step kernel issues every prefetch from every thread with clamped indices (no branch
// around a load: the compiler's outstanding-load bookkeeping stays exact), so most wave-loads of a small op re-fetch the last item
// in most lanes, or fetch weights an idle wave never uses.  Question: are such loads cheap (coalesced away) or do they cost the
// memory pipe as much as a useful one?  Every variant: 256 workgroups x 8 waves, each wave issues NL loads back to back (16 in
// flight), the blob is 2 MB and L2-resident, result = shader cycles per wave-load with all 8 waves issuing.
//   hipcc --offload-arch=gfx950 -O3 -o ta_cost ta_cost.hip && ./ta_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

enum { V_X4 = 0, V_X4_SAME, V_X4_Q, V_X4_ROW, V_X1, V_X1_SAME, V_BUF, V_BUF_OOR, V_BUF_Q_OOR, V_X2, NVAR };
static const char* kNames[NVAR] = {
    "dwordx4, 64 distinct lanes (1 KB)", "dwordx4, all lanes the same 16 B", "dwordx4, 16 lanes distinct + 48 clamped to the last",
    "dwordx4, lanes of a row share 16 B (4 x 16 B)", "dword, 64 distinct lanes (256 B)", "dword, all lanes the same 4 B",
    "buffer b128, 64 distinct lanes in range", "buffer b128, ALL lanes out of range", "buffer b128, 16 lanes in range + 48 out of range",
    "dwordx2, 64 distinct lanes (512 B)"};

template <int VAR>
__global__ __launch_bounds__(512) void ta(const char* __restrict__ blob, int nl, int waves_on, float* out, long long* cyc) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f32x4 acc = {0, 0, 0, 0};
  // buffer resource: base, stride 0, num_records (bytes), flags (raw, dword addressing)
  const int nrec = VAR == V_BUF_OOR ? 0 : (2 << 20);
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(blob), 0, nrec, 0x00020000);
  long long t0 = 0, t1 = 0;
  if (wave < waves_on) {
    t0 = __builtin_readcyclecounter();
    for (int i0 = 0; i0 < nl; i0 += 16) {
      f32x4 r[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        const unsigned frag = ((unsigned)(i0 + k) * 8u + wave) & 2047u;      // 1 KB fragment of the 2 MB blob
        unsigned off;
        if (VAR == V_X4 || VAR == V_BUF || VAR == V_BUF_OOR) off = frag * 1024u + lane * 16u;
        else if (VAR == V_X4_SAME) off = frag * 1024u;
        else if (VAR == V_X4_Q) off = frag * 1024u + (lane < 16 ? lane : 15) * 16u;
        else if (VAR == V_X4_ROW) off = frag * 1024u + (lane >> 4) * 16u;
        else if (VAR == V_X1) off = frag * 1024u + lane * 4u;
        else if (VAR == V_X1_SAME) off = frag * 1024u;
        else if (VAR == V_X2) off = frag * 1024u + lane * 8u;
        else off = lane < 16 ? frag * 1024u + lane * 16u : 0xfffffff0u;      // V_BUF_Q_OOR
        if (VAR == V_BUF || VAR == V_BUF_OOR || VAR == V_BUF_Q_OOR) {
          const i32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)off, 0, 0);
          r[k] = __builtin_bit_cast(f32x4, v);
        } else if (VAR == V_X1 || VAR == V_X1_SAME) {
          r[k] = f32x4{*(const float*)(blob + off), 0.f, 0.f, 0.f};
        } else if (VAR == V_X2) {
          const float2 v = *(const float2*)(blob + off);
          r[k] = f32x4{v.x, v.y, 0.f, 0.f};
        } else {
          r[k] = *(const f32x4*)(blob + off);
        }
      }
#pragma unroll
      for (int k = 0; k < 16; ++k) acc += r[k];
    }
    t1 = __builtin_readcyclecounter();
  }
  out[blockIdx.x * 512 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
  if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int VAR>
static void run(const char* blob, float* out, long long* cyc, std::vector<long long>& h) {
  for (int waves_on : {8, 1}) {
    const int nl = 256;
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(ta<VAR>, dim3(256), dim3(512), 0, 0, blob, nl, waves_on, out, cyc);
    hipDeviceSynchronize();
    hipMemcpy(h.data(), cyc, 256 * 8 * 8, hipMemcpyDeviceToHost);
    double mx = 0, sum = 0;
    for (int g = 0; g < 256; ++g)
      for (int w = 0; w < waves_on; ++w) { mx = h[g * 8 + w] > mx ? h[g * 8 + w] : mx; sum += h[g * 8 + w]; }
    const double mean = sum / (256.0 * waves_on);
    // with W waves issuing concurrently the pipe serves W * nl wave-loads in `mean` cycles
    printf("%-52s %d wave(s): %7.1f cycles per wave-load of a wave, %6.1f cycles of pipe per wave-load (max wave %.0f cycles)\n", kNames[VAR], waves_on,
           mean / nl, mean / (nl * waves_on), mx);
  }
}

int main() {
  char* blob; float* out; long long* cyc;
  hipMalloc(&blob, 4u << 20); hipMemset(blob, 0, 4u << 20);
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8 * 8);
  std::vector<long long> h(256 * 8);
  run<V_X4>(blob, out, cyc, h);
  run<V_X4_SAME>(blob, out, cyc, h);
  run<V_X4_Q>(blob, out, cyc, h);
  run<V_X4_ROW>(blob, out, cyc, h);
  run<V_X2>(blob, out, cyc, h);
  run<V_X1>(blob, out, cyc, h);
  run<V_X1_SAME>(blob, out, cyc, h);
  run<V_BUF>(blob, out, cyc, h);
  run<V_BUF_OOR>(blob, out, cyc, h);
  run<V_BUF_Q_OOR>(blob, out, cyc, h);
  return 0;
}
