// Offline / block mode: T consecutive frames of ONE utterance per call.
// The convolutions are causal in time with two taps (models/proposed.py:198-251, offline semantics
// :284-625), so with the activations of frame t stored in arena slot t+1 the "previous frame" tap of a
// layer is simply the same tensor one slot earlier -- every conv-like layer, CTFA gate and 1x1 layer of the
// block runs as ONE launch of the per-layer kernels (kernels.hip) with the frame index in the place of the
// stream index.  Only the 13 LSTMs are recurrent over frames; they are split here into
//   lstm_zx_kernel     (parallel over frames)  zx[t] = b + Wx . flatten(x_t)
//   lstm_scan_kernel   (one workgroup, sequential over frames)  gates from zx[t] + Wh . h_{t-1}; h_t, c_t
//   lstm_dense_kernel2 (parallel over frames)  y_t = Wd . h_t + bd
// (LSTM cell: models/proposed.py:70-119, converter_proposed.py:234-237; gate order i, f, g, o.)
#include <hip/hip_runtime.h>

#include "nutls_internal.hpp"

namespace nutls {

namespace {
constexpr int U = 21, G4 = 84;
// v_exp_f32 / v_rcp_f32 forms (as in the persistent kernel): the scan is a serial chain, IEEE expf / division are 10-30 instructions each
__device__ __forceinline__ float sigm(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float tanh_fast(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * x)); }
}  // namespace

// grid = frames; 128 threads.  zx [frames][84]
__global__ __launch_bounds__(128) void lstm_zx_kernel(const LstmParams p, float* __restrict__ zx) {
  __shared__ float v[256];
  const int b = blockIdx.x, tid = threadIdx.x;
  for (int k = tid; k < p.Din; k += 128) {
    const int f = k / p.x_cols, c = k - f * p.x_cols;
    v[k] = p.x[static_cast<size_t>(b) * p.sstride + static_cast<size_t>(f) * p.x_ld + c];
  }
  __syncthreads();
  if (tid < G4) {
    float a = p.bias[tid];
    for (int k = 0; k < p.Din; ++k) a = fmaf(p.wxT[k * G4 + tid], v[k], a);
    zx[static_cast<size_t>(b) * G4 + tid] = a;
  }
}

// one workgroup of 128 threads walks the frames; h/c of frame t-1 sit one arena slot before frame t's
__global__ __launch_bounds__(128) void lstm_scan_kernel(const LstmParams p, const float* __restrict__ zx, int frames) {
  __shared__ float hs[32];
  __shared__ float z[96];
  const int tid = threadIdx.x;
  float wh[U];
  if (tid < G4) {
#pragma unroll
    for (int u = 0; u < U; ++u) wh[u] = p.whT[u * G4 + tid];
  }
  float c_state = 0.f;
  if (tid < U) {
    hs[tid] = p.h_in[tid];
    c_state = p.c_in[tid];
  }
  __syncthreads();
  for (int t = 0; t < frames; ++t) {
    if (tid < G4) {
      float r = zx[static_cast<size_t>(t) * G4 + tid];
#pragma unroll
      for (int u = 0; u < U; ++u) r = fmaf(wh[u], hs[u], r);
      z[tid] = r;
    }
    __syncthreads();
    if (tid < U) {
      const float gi = sigm(z[tid]), gf = sigm(z[U + tid]), gg = tanh_fast(z[2 * U + tid]), go = sigm(z[3 * U + tid]);
      c_state = gf * c_state + gi * gg;
      const float h_new = go * tanh_fast(c_state);
      p.c_out[static_cast<size_t>(t) * p.sstride + tid] = c_state;
      p.h_out[static_cast<size_t>(t) * p.sstride + tid] = h_new;
      hs[tid] = h_new;
    }
    __syncthreads();
  }
}

// grid = frames; 128 threads
__global__ __launch_bounds__(128) void lstm_dense_kernel2(const LstmParams p) {
  __shared__ float hn[32];
  const int b = blockIdx.x, tid = threadIdx.x;
  if (tid < U) hn[tid] = p.h_out[static_cast<size_t>(b) * p.sstride + tid];
  __syncthreads();
  for (int m = tid; m < p.Dout; m += 128) {
    float a = p.bd[m];
#pragma unroll
    for (int u = 0; u < U; ++u) a = fmaf(p.wdT[u * p.Dout + m], hn[u], a);
    const int f = m / p.dst_cols, c = m - f * p.dst_cols;
    p.dst[static_cast<size_t>(b) * p.sstride + static_cast<size_t>(f) * p.dst_ld + c] = a;
  }
}

hipError_t launch_lstm_block(const LstmParams& p, float* zx, int frames, hipStream_t s) {
  if (p.Din > 256) return hipErrorInvalidValue;
  hipLaunchKernelGGL(lstm_zx_kernel, dim3(frames), dim3(128), 0, s, p, zx);
  hipLaunchKernelGGL(lstm_scan_kernel, dim3(1), dim3(128), 0, s, p, zx, frames);
  hipLaunchKernelGGL(lstm_dense_kernel2, dim3(frames), dim3(128), 0, s, p);
  return hipGetLastError();
}

}  // namespace nutls
