// Offline / block mode: T consecutive frames of ONE utterance per call.
// The convolutions are causal in time with two taps (models/proposed.py:198-251, offline semantics
// :284-625), so with the activations of frame t stored in arena slot t+1 the "previous frame" tap of a
// layer is simply the same tensor one slot earlier -- every conv-like layer, CTFA gate and 1x1 layer of the
// block runs as ONE launch of the per-layer kernels (kernels.hip) with the frame index in the place of the
// stream index.  Only the 13 LSTMs are recurrent over frames; they are split here into
//   lstm_zx_kernel     (parallel over frames)  zx[t] = b + Wx . flatten(x_t)
//   lstm_scan_kernel   (one wavefront, sequential over frames)  gates from zx[t] + Wh . h_{t-1}; h_t, c_t
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

// ONE wavefront walks the frames -- the recurrence is a serial chain of `frames` steps, so what counts is the latency
// of one step: no workgroup barrier, no LDS round trip, no memory wait on the chain.
//   lane 2u + p (u < 21): p = 0 owns gate columns i, f of unit u, p = 1 owns g, o (Keras column = gate * 21 + u);
//   h_{t-1} is broadcast with 21 v_readlane (lane 2u -> SGPR), the two lanes of a unit swap their gates with one
//   DPP quad permute, c_t / h_t live in the p = 0 lane;  zx is prefetched eight frames ahead in registers and
//   h_t / c_t (frame t's arena slot) are fire-and-forget stores.  tanh(x) = 2 sigmoid(2x) - 1 keeps both lanes
//   of a unit on one instruction stream.
template <int CTRL>
__device__ __forceinline__ float scan_dpp(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__global__ __launch_bounds__(64) void lstm_scan_kernel(const LstmParams p, const float* __restrict__ zx, int frames) {
  constexpr int PF = 8;                         // frames of zx in flight
  const int lane = threadIdx.x;
  const int u = lane >> 1, pr = lane & 1;
  const bool live = lane < 2 * U;
  const int na = live ? (pr ? 2 * U + u : u) : 0, nb = live ? (pr ? 3 * U + u : U + u) : 0;
  float wa[U], wb[U];
#pragma unroll
  for (int k = 0; k < U; ++k) {
    wa[k] = p.whT[k * G4 + na];
    wb[k] = p.whT[k * G4 + nb];
  }
  const float ka = pr ? 2.0f : 1.0f, da = pr ? -1.0f : 0.0f;      // column a: sigmoid (p = 0) or tanh (p = 1)
  float h = (live && !pr) ? p.h_in[u] : 0.f;
  float c = (live && !pr) ? p.c_in[u] : 0.f;
  float za[PF], zb[PF], ya[PF], yb[PF];
#pragma unroll
  for (int k = 0; k < PF; ++k) {
    const int t = k < frames ? k : frames - 1;
    za[k] = zx[static_cast<size_t>(t) * G4 + na];
    zb[k] = zx[static_cast<size_t>(t) * G4 + nb];
  }
  for (int t0 = 0; t0 < frames; t0 += PF) {
#pragma unroll
    for (int k = 0; k < PF; ++k) {              // the block after this one
      const int t = t0 + PF + k < frames ? t0 + PF + k : frames - 1;
      ya[k] = zx[static_cast<size_t>(t) * G4 + na];
      yb[k] = zx[static_cast<size_t>(t) * G4 + nb];
    }
#pragma unroll
    for (int k = 0; k < PF; ++k) {
      const int t = t0 + k;
      if (t < frames) {
        float ra0 = za[k], ra1 = 0.f, rb0 = zb[k], rb1 = 0.f;
        const int hbits = __builtin_bit_cast(int, h);
#pragma unroll
        for (int j = 0; j < U; ++j) {
          const float hj = __builtin_bit_cast(float, __builtin_amdgcn_readlane(hbits, 2 * j));
          if (j & 1) { ra1 = fmaf(wa[j], hj, ra1); rb1 = fmaf(wb[j], hj, rb1); }
          else       { ra0 = fmaf(wa[j], hj, ra0); rb0 = fmaf(wb[j], hj, rb0); }
        }
        const float ga = ka * sigm(ka * (ra0 + ra1)) + da;      // i (p = 0) / g (p = 1)
        const float gb = sigm(rb0 + rb1);                       // f (p = 0) / o (p = 1)
        const float gg = scan_dpp<0xB1>(ga), go = scan_dpp<0xB1>(gb);      // quad_perm [1,0,3,2]: the other lane of the unit
        c = gb * c + ga * gg;
        h = go * tanh_fast(c);
        if (live && !pr) {
          p.c_out[static_cast<size_t>(t) * p.sstride + u] = c;
          p.h_out[static_cast<size_t>(t) * p.sstride + u] = h;
        }
      }
    }
#pragma unroll
    for (int k = 0; k < PF; ++k) { za[k] = ya[k]; zb[k] = yb[k]; }
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
  hipLaunchKernelGGL(lstm_scan_kernel, dim3(1), dim3(64), 0, s, p, zx, frames);
  hipLaunchKernelGGL(lstm_dense_kernel2, dim3(frames), dim3(128), 0, s, p);
  return hipGetLastError();
}

}  // namespace nutls
