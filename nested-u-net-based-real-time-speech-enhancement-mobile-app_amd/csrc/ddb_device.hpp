// Dilated-dense bottleneck of the NUNet-TLS *baseline* variant, one stream per call.
// Reference: blocks /root/reference/dnn_model/models/nunet_tls.py:277-359 (dilated_dense_block_*_valid),
// streaming wiring converter_nunet_tls.py:374-411, state shift :1420-1427.
//
//   o_0 = PReLU(conv(2,3)([prev_in ; x]) : C -> G)                                  G = C/2
//   k = 1..6, d = 2^(k-1):  in_k = concat_C[o_{k-1}, ..., o_0]   (newest first, k*G channels)
//        y   = grouped(2,3) conv, groups = G (filter g sees channels g*k .. g*k+k-1), dilation d in
//              time (taps: frame t-d, frame t) AND frequency (bins f-d, f, f+d, zero padded)
//        o_k = PReLU(LN(W1 y + b1))
//   out = PReLU(conv(2,3)([prev_out ; o_6]) : G -> C)
//
// State: the reference keeps, per block k, the last d frames of in_k and shifts them by one every
// step.  Here the d frames live in a ring in HBM: slot (step mod d) holds frame t-d, is read, then
// overwritten with frame t -- no shifting traffic (the C ABI rotates on get/set so callers still see
// "oldest first").  Everything here is tiny (F <= 4 rows, <= 192 channels): plain VALU code.
#pragma once
#include <hip/hip_runtime.h>

#include "nutls_internal.hpp"

namespace nutls {

__device__ __forceinline__ float ddb_prelu(float v, float a) { return v >= 0.f ? v : a * v; }

// lds: >= 2*F*C + 8*F*G floats.  nthreads >= F*C, all threads of the workgroup must call.
__device__ __forceinline__ void ddb_block(const DdbParams& p, int stream, float* lds, int tid, int nthreads) {
  const int F = p.F, C = p.C, G = C >> 1;
  const int FG = F * G, FC = F * C;
  float* xs = lds;                 // [F][C]  current input
  float* pin = xs + FC;            // [F][C]  previous input (prev_in)
  float* o = pin + FC;             // [7][F][G]  o_0 .. o_6
  float* yv = o + 7 * FG;          // [F][G]
  const size_t soff = static_cast<size_t>(stream) * p.sstride;
  const int step = *p.step;
  const float* xg = p.x + soff;
  float* st_in = p.st_in + soff;
  // ---- stage x and prev_in; then prev_in <- x
  for (int q = tid; q < FC; q += nthreads) {
    const int f = q / C, c = q - f * C;
    xs[q] = xg[f * p.x_ld + c];
    pin[q] = st_in[q];
  }
  __syncthreads();
  for (int q = tid; q < FC; q += nthreads) st_in[q] = xs[q];
  // ---- o_0
  if (tid < FG) {
    const int f = tid / G, g = tid - f * G;
    float a = p.b_in[g];
    for (int t = 0; t < 2; ++t) {
      const float* X = t ? xs : pin;
      for (int kw = 0; kw < 3; ++kw) {
        const int fr = f + kw - 1;
        if (fr < 0 || fr >= F) continue;
        const float* wt = p.w_in + static_cast<size_t>((t * 3 + kw) * C) * G + g;     // [t][kw][c][g]
        for (int c = 0; c < C; ++c) a = fmaf(wt[c * G], X[fr * C + c], a);
      }
    }
    o[tid] = ddb_prelu(a, p.a_in);
  }
  __syncthreads();
  // ---- blocks 1..6
  for (int k = 1; k <= 6; ++k) {
    const int d = 1 << (k - 1);
    const int kG = k * G;
    float* ring = p.st_blk[k - 1] + soff + static_cast<size_t>(step & (d - 1)) * F * kG;   // frame t-d, then frame t
    if (tid < FG) {
      const int f = tid / G, g = tid - f * G;
      float a = p.bg[k - 1][g];
      for (int kw = 0; kw < 3; ++kw) {
        const int fr = f + (kw - 1) * d;
        if (fr < 0 || fr >= F) continue;
        for (int j = 0; j < k; ++j) {
          const int ch = g * k + j;             // channel of in_k seen by filter g
          const int m = ch / G;                 // in_k = [o_{k-1}, ..., o_0]: chunk m is o_{k-1-m}
          const float cur = o[(k - 1 - m) * FG + fr * G + (ch - m * G)];
          const float old = ring[fr * kG + ch];
          const float* wt = p.wg[k - 1] + static_cast<size_t>(kw * k + j) * G + g;     // [t][kw][j][g]
          a = fmaf(wt[0], old, a);
          a = fmaf(wt[static_cast<size_t>(3 * k) * G], cur, a);
        }
      }
      yv[tid] = a;
    }
    __syncthreads();              // every read of the ring slot is done
    for (int q = tid; q < F * kG; q += nthreads) {
      const int f = q / kG, ch = q - f * kG;
      const int m = ch / G;
      ring[q] = o[(k - 1 - m) * FG + f * G + (ch - m * G)];
    }
    if (tid < FG) {
      const int f = tid / G, g = tid - f * G;
      float z = p.b1[k - 1][g];
      for (int gi = 0; gi < G; ++gi) z = fmaf(p.w1[k - 1][gi * G + g], yv[f * G + gi], z);     // [gin][gout]
      // LayerNorm over the G channels of row f = G consecutive lanes
      float s = z;
      for (int msk = 1; msk < G; msk <<= 1) s += __shfl_xor(s, msk);
      const float mean = s / static_cast<float>(G);
      const float dv = z - mean;
      float q2 = dv * dv;
      for (int msk = 1; msk < G; msk <<= 1) q2 += __shfl_xor(q2, msk);
      const float rstd = 1.0f / sqrtf(q2 / static_cast<float>(G) + 1e-8f);
      o[k * FG + tid] = ddb_prelu(dv * rstd * p.gamma[k - 1][g] + p.beta[k - 1][g], p.alpha[k - 1]);
    }
    __syncthreads();
  }
  // ---- out conv over [prev_out ; o_6], then prev_out <- o_6
  float* st_out = p.st_out + soff;
  float res = 0.f;
  if (tid < FC) {
    const int f = tid / C, c = tid - f * C;
    float a = p.b_out[c];
    for (int t = 0; t < 2; ++t) {
      for (int kw = 0; kw < 3; ++kw) {
        const int fr = f + kw - 1;
        if (fr < 0 || fr >= F) continue;
        const float* wt = p.w_out + static_cast<size_t>((t * 3 + kw) * G) * C + c;   // [t][kw][g][c]
        for (int g = 0; g < G; ++g) {
          const float v = t ? o[6 * FG + fr * G + g] : st_out[fr * G + g];
          a = fmaf(wt[g * C], v, a);
        }
      }
    }
    res = ddb_prelu(a, p.a_out);
  }
  __syncthreads();                // prev_out fully read
  for (int q = tid; q < FG; q += nthreads) st_out[q] = o[6 * FG + q];
  if (tid < FC) {
    const int f = tid / C, c = tid - f * C;
    p.dst[soff + f * p.dst_ld + c] = res;
  }
  __syncthreads();
}

}  // namespace nutls
