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
constexpr float kLog2e = 1.44269504088896341f;
// (sigmoid / tanh of the scan: v_exp_f32 / v_rcp_f32 forms as in the fused kernel -- the scan is a serial chain, IEEE expf / division are
// 10-30 instructions each)
}  // namespace

// grid = frames; 128 threads.  zx [frames][84]
__global__ __launch_bounds__(128) void lstm_zx_kernel(const LstmParams p, float* __restrict__ zx) {
  __shared__ float v[256];
  const int b = blockIdx.x, tid = threadIdx.x;
  for (int k = tid; k < p.Din; k += 128) {
    const int f = k / p.x_cols, c = k - f * p.x_cols;
    v[k] = p.x[slot_of(b, p.sm) * p.sstride + static_cast<size_t>(f) * p.x_ld + c];
  }
  __syncthreads();
  if (tid < G4) {
    float a = p.bias[tid];
    for (int k = 0; k < p.Din; ++k) a = fmaf(p.wxT[k * G4 + tid], v[k], a);
    // stored as the exponent the scan feeds to v_exp_f32: sigmoid(x) = 1 / (1 + 2^(-log2e x)), tanh(x) = 2 sigmoid(2x) - 1 (column g)
    zx[static_cast<size_t>(b) * G4 + tid] = a * (tid >= 2 * U && tid < 3 * U ? -2.0f * kLog2e : -kLog2e);
  }
}

// ONE wavefront walks the frames -- the recurrence is a serial chain of `frames` steps, so what counts is the issue
// count and latency of one step: no workgroup barrier, no LDS round trip, no memory wait on the chain.
//   lane 2u + p (u < 21): p = 0 owns gate columns i, f of unit u, p = 1 owns g, o (Keras column = gate * 21 + u);
//   lanes 42..63 repeat lanes 40 / 41 (same values to the same addresses: no exec masking anywhere in the loop).
//   h_{t-1} is broadcast with 21 v_readlane into 21 SGPRs up front (back to back: a readlane directly followed by
//   its consumer costs two wait states each), the two columns of a lane advance together on v_pk_fma_f32 (21 packed
//   FMAs instead of 42), the p = 0 lane of a unit reads the other lane's gates through DPP quad permutes and keeps
//   c_t / h_t; it stores c_t, the p = 1 lane stores h_t (one store instruction per frame).  zx is prefetched eight frames
//   ahead in registers.  The loop over full 8-frame groups has no branch, so the compiler counts the stores behind a
//   group's prefetch exactly (vmcnt is one in-order counter for loads and stores on gfx9: behind a branch it assumed
//   no store had been issued and waited for all of them at every group).  tanh(x) = 2 sigmoid(2x) - 1 keeps both
//   lanes of a unit on one instruction stream.
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int CTRL>
__device__ __forceinline__ float scan_dpp(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__global__ __launch_bounds__(64) void lstm_scan_kernel(const LstmParams p0, const float* __restrict__ zx0, int frames, long long utt_stride) {
  // one wavefront per utterance (blockIdx.x): its h / c slots start utt_stride floats after the previous utterance's, its rows of zx `frames` rows further
  LstmParams p = p0;
  p.h_in += blockIdx.x * utt_stride; p.c_in += blockIdx.x * utt_stride; p.h_out += blockIdx.x * utt_stride; p.c_out += blockIdx.x * utt_stride;
  const float* __restrict__ zx = zx0 + static_cast<size_t>(blockIdx.x) * frames * G4;
  constexpr int PF = 8;                         // frames of zx in flight
  const int lane = threadIdx.x;
  const int u = (lane >> 1) < U ? (lane >> 1) : U - 1, pr = lane & 1;
  const int na = pr ? 2 * U + u : u, nb = pr ? 3 * U + u : U + u;
  const float ka = pr ? 2.0f : 1.0f, da = pr ? -1.0f : 0.0f;      // column a: sigmoid (p = 0) or tanh (p = 1)
  const f32x2 sc = {-ka * kLog2e, -kLog2e};                          // sigmoid(k x) = 1 / (1 + 2^(-k log2e x))
  f32x2 w[U];
#pragma unroll
  for (int k = 0; k < U; ++k) w[k] = f32x2{p.whT[k * G4 + na], p.whT[k * G4 + nb]} * sc;      // exponent scale folded in (zx carries it too)
  float h = p.h_in[u], c = p.c_in[u];
  float* __restrict__ outp = pr ? p.h_out + u : p.c_out + u;
  const float* zp = zx + na;                    // column b = column a + 21 for both lanes of a unit
  f32x2 z[PF], y[PF];
  auto step = [&](const f32x2 zt, const int t) {
    const int hbits = __builtin_bit_cast(int, h);
    float hs[U];
#pragma unroll
    for (int j = 0; j < U; ++j) hs[j] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(hbits, 2 * j));
    __builtin_amdgcn_sched_barrier(0);          // all 21 broadcasts before the first FMA
    f32x2 r0 = zt, r1 = {0.f, 0.f};             // two chains: a packed FMA that reads the one before it costs a wait state
#pragma unroll
    for (int j = 0; j < U; ++j) {
      const f32x2 hj = {hs[j], hs[j]};
      if (j & 1) r1 = __builtin_elementwise_fma(w[j], hj, r1);
      else       r0 = __builtin_elementwise_fma(w[j], hj, r0);
    }
    const f32x2 r = r0 + r1;
    const f32x2 d = f32x2{__builtin_amdgcn_exp2f(r.x), __builtin_amdgcn_exp2f(r.y)} + 1.0f;
    const float ga = fmaf(ka, __builtin_amdgcn_rcpf(d.x), da);     // i (p = 0) / g (p = 1)
    const float gb = __builtin_amdgcn_rcpf(d.y);                   // f (p = 0) / o (p = 1)
    // c_t, h_t are kept by the p = 0 lane of a unit (own f, the p = 1 lane's g and o through DPP quad_perm [1,0,3,2]);
    // what the p = 1 lane computes beside it is never read.  Its store slot takes the h of its neighbour ([0,0,2,2]).
    c = fmaf(gb, c, ga * scan_dpp<0xB1>(ga));
    const float th = fmaf(__builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(c * (2.0f * kLog2e))), -2.0f, 1.0f);
    h = scan_dpp<0xB1>(gb) * th;
    const float hn = scan_dpp<0xA0>(h);
    outp[static_cast<size_t>(t) * p.sstride] = pr ? hn : c;
  };
  // the group of eight frames at q: 16 loads at immediate offsets from one pointer, no clamping -- the last rounds read
  // up to 16 rows past the block (rows of the next chunk or the slack rows of the buffer: kScanReadAhead, never used)
  auto fetch = [&](f32x2 (&d)[PF], const float* q) {
#pragma unroll
    for (int k = 0; k < PF; ++k) d[k] = f32x2{q[k * G4], q[k * G4 + U]};
  };
  // two groups per round, z and y swapping roles (no register copies: a copy deferred to the loop head waits with the
  // preheader's count and drains the stores of the group before it)
  // h, c (and the weights before them) resident before the loop: a load still counted as pending at the loop head
  // would be waited for with the preheader's count on every round
  asm volatile("; h, c resident" : "+v"(h), "+v"(c));
  fetch(z, zp);
  int t0 = 0;
  for (; t0 + 2 * PF <= frames; t0 += 2 * PF) {
    fetch(y, zp + PF * G4);
#pragma unroll
    for (int k = 0; k < PF; ++k) step(z[k], t0 + k);
    fetch(z, zp + 2 * PF * G4);
    zp += 2 * PF * G4;
#pragma unroll
    for (int k = 0; k < PF; ++k) step(y[k], t0 + PF + k);
  }
  fetch(y, zp + PF * G4);                       // the last frames % 16 frames
#pragma unroll
  for (int k = 0; k < PF; ++k)
    if (t0 + k < frames) step(z[k], t0 + k);
#pragma unroll
  for (int k = 0; k < PF; ++k)
    if (t0 + PF + k < frames) step(y[k], t0 + PF + k);
}

// grid = frames; 128 threads
__global__ __launch_bounds__(128) void lstm_dense_kernel2(const LstmParams p) {
  __shared__ float hn[32];
  const int b = blockIdx.x, tid = threadIdx.x;
  if (tid < U) hn[tid] = p.h_out[slot_of(b, p.sm) * p.sstride + tid];
  __syncthreads();
  for (int m = tid; m < p.Dout; m += 128) {
    float a = p.bd[m];
#pragma unroll
    for (int u = 0; u < U; ++u) a = fmaf(p.wdT[u * p.Dout + m], hn[u], a);
    const int f = m / p.dst_cols, c = m - f * p.dst_cols;
    p.dst[slot_of(b, p.sm) * p.sstride + static_cast<size_t>(f) * p.dst_ld + c] = a;
  }
}

hipError_t launch_lstm_block(const LstmParams& p, float* zx, int frames, hipStream_t s, int utts, long long utt_stride) {
  if (p.Din > 256) return hipErrorInvalidValue;
  hipLaunchKernelGGL(lstm_zx_kernel, dim3(utts * frames), dim3(128), 0, s, p, zx);
  hipLaunchKernelGGL(lstm_scan_kernel, dim3(utts), dim3(64), 0, s, p, zx, frames, utt_stride);
  hipLaunchKernelGGL(lstm_dense_kernel2, dim3(utts * frames), dim3(128), 0, s, p);
  return hipGetLastError();
}

}  // namespace nutls
