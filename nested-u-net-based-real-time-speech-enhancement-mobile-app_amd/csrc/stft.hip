// STFT front end and inverse-STFT / overlap-add back end of the streaming loop, on the device
// (SURVEY.md section 8(f).1).  Reference: /root/reference/dnn_model/interpreter_proposed.py
//   :17-26   frame 512 / hop 256, analysis window = periodic Hann with both end taps 1e-7,
//            inverse window = hann / (hann^2 + hann_shifted_by_hop^2)
//   :203-213 in_buffer shift, rfft(in_buffer * win), |X|, angle(X), bins 1..256 -> model
//   :352-365 DC re-created by edge padding (phone: zero, RTSE_NUTLS_LSTM.java:677), irfft(mag * e^{j phase}),
//            * inverse window, overlap-add, first half of the buffer leaves
// One workgroup (256 threads) per stream per hop: a 512-point radix-2 FFT in LDS (9 stages, one
// butterfly per thread and stage; twiddles from a 256-entry table computed on the host in double
// precision).  Per stream the library keeps the previous hop (analysis) and the overlap tail
// (synthesis); the phase travels as the unit phasor X/|X| (angle(0) = 0 -> (1, 0)).
#include <hip/hip_runtime.h>

#include "nutls_internal.hpp"

namespace nutls {

namespace {

constexpr int N = NUTLS_FRAME_LEN;      // 512
constexpr int H = NUTLS_FRAME_STEP;     // 256

__device__ __forceinline__ int bitrev9(int v) { return static_cast<int>(__brev(static_cast<unsigned>(v)) >> 23); }

// In-place radix-2 decimation-in-time FFT of 512 complex points in LDS; input in bit-reversed order.
// sign = -1: forward (e^{-j...}), +1: inverse (unscaled).  256 threads, every thread calls.
__device__ __forceinline__ void fft512(float* re, float* im, const float2* __restrict__ tw, int tid, float sign) {
#pragma unroll 1
  for (int s = 0; s < 9; ++s) {
    const int half = 1 << s;
    const int j = tid & (half - 1);
    const int i0 = ((tid >> s) << (s + 1)) + j, i1 = i0 + half;
    const float2 w = tw[j << (8 - s)];                 // e^{-2 pi i j / (2 half)}
    const float wr = w.x, wi = sign < 0.f ? w.y : -w.y;
    __syncthreads();
    const float xr = re[i1], xi = im[i1];
    const float tr = wr * xr - wi * xi, ti = wr * xi + wi * xr;
    const float ur = re[i0], ui = im[i0];
    re[i0] = ur + tr; im[i0] = ui + ti;
    re[i1] = ur - tr; im[i1] = ui - ti;
  }
  __syncthreads();
}

}  // namespace

// pcm [B,256] new hop; tail [B,256] previous hop (updated); mag [B,256] = |X| bins 1..256;
// ph [B,257] unit phasors of bins 0..256
__global__ __launch_bounds__(256) void stft_hop_kernel(const float* __restrict__ pcm, float* __restrict__ tail,
                                                      const float* __restrict__ win, const float2* __restrict__ tw,
                                                      float* __restrict__ mag, float2* __restrict__ ph) {
  __shared__ float re[N], im[N];
  const int b = blockIdx.x, tid = threadIdx.x;
  const float x_old = tail[static_cast<size_t>(b) * H + tid];
  const float x_new = pcm[static_cast<size_t>(b) * H + tid];
  tail[static_cast<size_t>(b) * H + tid] = x_new;
  re[bitrev9(tid)] = x_old * win[tid];
  re[bitrev9(tid + H)] = x_new * win[tid + H];
  im[tid] = 0.f;
  im[tid + H] = 0.f;
  fft512(re, im, tw, tid, -1.f);
  for (int k = tid; k <= H; k += 256) {
    const float xr = re[k], xi = im[k];
    const float m = sqrtf(xr * xr + xi * xi);
    if (k >= 1) mag[static_cast<size_t>(b) * H + (k - 1)] = m;
    ph[static_cast<size_t>(b) * (H + 1) + k] = m > 0.f ? make_float2(xr / m, xi / m) : make_float2(1.f, 0.f);
  }
}

// est [B,256] model output (bins 1..256); ph phasors of the same frame; ola [B,256] overlap tail (updated);
// pcm_out [B,256].  dc_edge: bin 0 = est[0] (PC loop) else 0 (phone).
__global__ __launch_bounds__(256) void istft_hop_kernel(const float* __restrict__ est, const float2* __restrict__ ph,
                                                       const float* __restrict__ inv_win, const float2* __restrict__ tw,
                                                       float* __restrict__ ola, float* __restrict__ pcm_out, int dc_edge) {
  __shared__ float re[N], im[N];
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* e = est + static_cast<size_t>(b) * H;
  const float2* p = ph + static_cast<size_t>(b) * (H + 1);
  // Hermitian spectrum: bins 0..256 given, 257..511 mirrored; irfft ignores the imaginary parts of bins 0 and 256
  {
    const int k = tid;                                   // bins 0..255
    const float m = k == 0 ? (dc_edge ? e[0] : 0.f) : e[k - 1];
    const float yr = m * p[k].x, yi = k == 0 ? 0.f : m * p[k].y;
    re[bitrev9(k)] = yr; im[bitrev9(k)] = yi;
    if (k >= 1) { re[bitrev9(N - k)] = yr; im[bitrev9(N - k)] = -yi; }
    if (tid == 0) { re[bitrev9(H)] = e[H - 1] * p[H].x; im[bitrev9(H)] = 0.f; }
  }
  fft512(re, im, tw, tid, 1.f);
  const float scale = 1.0f / static_cast<float>(N);
  const float y0 = re[tid] * scale * inv_win[tid];
  const float y1 = re[tid + H] * scale * inv_win[tid + H];
  const size_t o = static_cast<size_t>(b) * H + tid;
  pcm_out[o] = ola[o] + y0;
  ola[o] = y1;
}

hipError_t launch_stft_hop(const float* pcm, float* tail, const float* win, const float* tw, float* mag, float* ph, int B, hipStream_t s) {
  hipLaunchKernelGGL(stft_hop_kernel, dim3(B), dim3(256), 0, s, pcm, tail, win, reinterpret_cast<const float2*>(tw), mag,
                     reinterpret_cast<float2*>(ph));
  return hipGetLastError();
}

hipError_t launch_istft_hop(const float* est, const float* ph, const float* inv_win, const float* tw, float* ola, float* pcm_out,
                            int dc_edge, int B, hipStream_t s) {
  hipLaunchKernelGGL(istft_hop_kernel, dim3(B), dim3(256), 0, s, est, reinterpret_cast<const float2*>(ph), inv_win,
                     reinterpret_cast<const float2*>(tw), ola, pcm_out, dc_edge);
  return hipGetLastError();
}

}  // namespace nutls
