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

// -------------------------------------------------------------------------------------------------
//  The same block for the persistent kernel (512 threads, ~70 KB of LDS scratch): everything that does
//  not depend on this frame's arithmetic -- the history rings, the previous input / output rows and
//  the small weights of the six grouped blocks -- is fetched into LDS by all threads up front, and the
//  two dense (2,3) convs (`in`: C -> G, `out`: G -> C) are split over K across the whole workgroup
//  (float4 weight loads, coalesced over the output channel) instead of running on F*G threads.
// -------------------------------------------------------------------------------------------------
namespace nutls {

typedef float ddb_f4 __attribute__((ext_vector_type(4)));

// sum over G = 16 or 32 consecutive lanes (aligned), result in all of them: DPP quad / row exchanges, one
// LDS-crossbar step only for the 32-lane case
template <int CTRL>
__device__ __forceinline__ float ddb_dpp(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float ddb_row_sum(float s, int G) {
  s += ddb_dpp<0xB1>(s);      // quad_perm [1,0,3,2]
  s += ddb_dpp<0x4E>(s);      // quad_perm [2,3,0,1]
  s += ddb_dpp<0x141>(s);     // row_half_mirror
  s += ddb_dpp<0x140>(s);     // row_mirror
  if (G == 32) s += __shfl_xor(s, 16);
  return s;
}

// K-split dense (2,3) conv:  out[f][co] = sum_{t,kw,ci} W[t][kw][ci][co] * X_t[f+kw-1][ci]
//   X0 / X1: LDS rows [F][CI] of the previous / current frame; W: global [2][3][CI][CO]
//   part: LDS [nks][F*CO/4] float4 partial sums.  All nthreads call; the result is left in part[0..] summed
//   by the first F*CO/4 threads (returned in `acc` for those threads, valid when tid < F*CO/4).
__device__ __forceinline__ ddb_f4 ddb_dense23(const float* X0, const float* X1, const float* W, int F, int CI, int CO, float* part,
                                              int tid, int nthreads) {
  const int nq = (F * CO) >> 2;                 // float4 outputs (a power of two)
  const int entries = 6 * CI;                   // (t, kw, ci); CI is a power of two
  // (F, CI, CO are powers of two: no integer divisions anywhere in this block -- a division by a run-time value
  //  is ~40 instructions)
  int lci = 0, lnq = 0, lcq = 0, lth = 0;
  while ((1 << lci) < CI) ++lci;
  while ((1 << lnq) < nq) ++lnq;
  while ((4 << lcq) < CO) ++lcq;
  while ((2 << lth) <= nthreads) ++lth;         // floor(log2(nthreads))
  // K slices: a power of two that divides 6*CI = 3 * 2^(lci+1), i.e. at most 2*CI, and fits the workgroup
  int lks = lth - lnq;
  if (lks > lci + 1) lks = lci + 1;
  const int nks = 1 << lks;
  const int epk = entries >> lks;               // <= 24
  const int q = tid & (nq - 1), ks = tid >> lnq;
  const int f = q >> lcq, cq = q & ((CO >> 2) - 1);
  ddb_f4 a = {0.f, 0.f, 0.f, 0.f};
  if (ks < nks) {
    // all weight loads of the slice first (independent, one L2 latency), then the products
    ddb_f4 w[24];
    float xv[24];
#pragma unroll
    for (int j = 0; j < 24; ++j) {
      if (j < epk) {
        const int e = ks * epk + j;
        const int t = e >= 3 * CI ? 1 : 0, r = e - t * 3 * CI;
        const int kw = r >> lci, ci = r & (CI - 1);
        const int fr = f + kw - 1;
        const bool ok = fr >= 0 && fr < F;
        w[j] = *reinterpret_cast<const ddb_f4 __attribute__((address_space(1)))*>((unsigned long long)(W + static_cast<size_t>(e) * CO + 4 * cq));
        xv[j] = ok ? (t ? X1 : X0)[fr * CI + ci] : 0.f;
      }
    }
#pragma unroll
    for (int j = 0; j < 24; ++j)
      if (j < epk) a += w[j] * xv[j];
    *reinterpret_cast<ddb_f4*>(part + (static_cast<size_t>(ks) * nq + q) * 4) = a;
  }
  __syncthreads();
  ddb_f4 s = {0.f, 0.f, 0.f, 0.f};
  if (tid < nq)
    for (int k2 = 0; k2 < nks; ++k2) s += *reinterpret_cast<const ddb_f4*>(part + (static_cast<size_t>(k2) * nq + tid) * 4);
  return s;
}

// lds: >= 17.5K floats.  All nthreads (a multiple of 64, >= 256) call.
// lds_y (optional): the output rows are also written to LDS at lds_y + f * lds_y_pitch + c (the fused kernel's next image).
__device__ __forceinline__ void ddb_block_wg(const DdbParams& p, int stream, float* lds, int tid, int nthreads,
                                             unsigned long long* dbg_lds = nullptr, float* lds_y = nullptr, int lds_y_pitch = 0) {
#define DDB_T(k) do { if (dbg_lds && (tid & 63) == 0) dbg_lds[(tid >> 6) * 16 + (k)] = clock64(); } while (0)
  DDB_T(0);
  const int F = p.F, C = p.C, G = C >> 1;
  const int FG = F * G, FC = F * C;
  int lg = 0;
  while ((1 << lg) < G) ++lg;                // G = 16 or 32, C = 2G
  float* xs = lds;                    // [F][C]   current input
  float* pin = xs + FC;               // [F][C]   previous input
  float* o = pin + FC;                // [7][F][G] o_0 .. o_6
  float* yv = o + 7 * FG;             // [F][G]
  float* pout = yv + FG;              // [F][G]   previous o_6
  float* rings = pout + FG;           // block k at rings + F*G*k(k-1)/2 : [F][k*G] (frame t-d)
  float* wgs = rings + 21 * FG;       // block k at wgs + 6*G*k(k-1)/2 : [2][3][k][G]
  float* w1s = wgs + 126 * G;         // [6][G][G]
  float* sm = w1s + 6 * G * G;        // [6][4][G] bg, b1, gamma, beta
  float* part = sm + 24 * G;          // K-split partial sums (<= 2048 floats)
  const size_t soff = static_cast<size_t>(stream) * p.sstride;
  const int step = *p.step;
  // (`p` lives in global memory and the block stores to global memory: every p.field the loops below touched
  //  would be re-read after each store -- take what the loops need into registers once)
  const float* const px = p.x + soff;
  float* const pst_in = p.st_in + soff;
  float* const pst_out = p.st_out + soff;
  const int x_ld = p.x_ld;
  float* rp[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) rp[k] = p.st_blk[k] + soff + static_cast<size_t>(step & ((1 << k) - 1)) * F * (k + 1) * G;
  // ---- phase A: everything that is already known
  for (int q = tid; q < FC; q += nthreads) {
    const int f = q >> (lg + 1), c = q & (C - 1);
    xs[q] = px[f * x_ld + c];
    pin[q] = pst_in[q];
  }
  for (int q = tid; q < FG; q += nthreads) pout[q] = pst_out[q];
  {
    // small weights: one packed blob in exactly this LDS order (wgs | w1s | sm), float4 copies, all independent
    const int n4 = (126 * G + 6 * G * G + 24 * G) / 4;
    const ddb_f4 __attribute__((address_space(1)))* src = (const ddb_f4 __attribute__((address_space(1)))*)(unsigned long long)p.wsmall;
    for (int q = tid; q < n4; q += nthreads) reinterpret_cast<ddb_f4*>(wgs)[q] = src[q];
    // history rings of the six blocks as one flat index space (block k starts at FG * k(k-1)/2)
    for (int q = tid; q < 21 * FG; q += nthreads) {
      // block k starts at FG * k(k-1)/2: q < FG -> 1, < 3FG -> 2, < 6FG -> 3, < 10FG -> 4, < 15FG -> 5, else 6
      const int k = q < FG ? 1 : (q < 3 * FG ? 2 : (q < 6 * FG ? 3 : (q < 10 * FG ? 4 : (q < 15 * FG ? 5 : 6))));
      const int base = FG * (k * (k - 1) / 2);
      const float* r = k == 1 ? rp[0] : (k == 2 ? rp[1] : (k == 3 ? rp[2] : (k == 4 ? rp[3] : (k == 5 ? rp[4] : rp[5]))));
      rings[q] = r[q - base];
    }
  }
  DDB_T(1);
  __syncthreads();
  DDB_T(2);
  for (int q = tid; q < FC; q += nthreads) pst_in[q] = xs[q];      // prev_in <- x
  // ---- o_0 = PReLU(conv(2,3)([prev_in ; x]))
  {
    const ddb_f4 s = ddb_dense23(pin, xs, p.w_in, F, C, G, part, tid, nthreads);
    if (tid < FG / 4) {
      const int cq = tid & ((G >> 2) - 1);
#pragma unroll
      for (int j = 0; j < 4; ++j) o[4 * tid + j] = ddb_prelu(s[j] + p.b_in[4 * cq + j], p.a_in);
    }
    __syncthreads();
  }
  DDB_T(3);
  // ---- blocks 1..6 (sequential: dense connectivity)
  int roff = 0, woff = 0;
  for (int k = 1; k <= 6; ++k) {
    const int d = 1 << (k - 1), kG = k * G;
    const float* ringl = rings + roff;
    const float* wgl = wgs + woff;
    const float* sml = sm + (k - 1) * 4 * G;
    if (tid < FG) {
      const int f = tid >> lg, g = tid & (G - 1);
      float a = sml[g];
      for (int kw = 0; kw < 3; ++kw) {
        const int fr = f + (kw - 1) * d;
        if (fr < 0 || fr >= F) continue;
        for (int j = 0; j < k; ++j) {
          const int ch = g * k + j;             // channel of in_k seen by filter g
          const int m = ch >> lg;               // in_k = [o_{k-1}, ..., o_0]: chunk m is o_{k-1-m}
          const float cur = o[(k - 1 - m) * FG + fr * G + (ch & (G - 1))];
          const float old = ringl[fr * kG + ch];
          a = fmaf(wgl[(kw * k + j) * G + g], old, a);
          a = fmaf(wgl[(3 * k + kw * k + j) * G + g], cur, a);
        }
      }
      yv[tid] = a;
    }
    // ring slot <- in_k of this frame (every read of the old slot went through LDS)
    {
      float* ring = k == 1 ? rp[0] : (k == 2 ? rp[1] : (k == 3 ? rp[2] : (k == 4 ? rp[3] : (k == 5 ? rp[4] : rp[5]))));
      for (int q = tid; q < F * kG; q += nthreads) {
        const int f = (q >= kG) + (q >= 2 * kG) + (q >= 3 * kG), ch = q - f * kG;      // F <= 4
        const int m = ch >> lg;
        ring[q] = o[(k - 1 - m) * FG + f * G + (ch & (G - 1))];
      }
    }
    __syncthreads();
    if (tid < FG) {
      const int f = tid >> lg, g = tid & (G - 1);
      const float* w1l = w1s + (k - 1) * G * G;
      float z = sml[G + g];
      for (int gi = 0; gi < G; ++gi) z = fmaf(w1l[gi * G + g], yv[f * G + gi], z);     // [gin][gout]
      // LayerNorm over the G channels of row f = G consecutive lanes
      const float inv_g = 1.0f / static_cast<float>(G);
      const float mean = ddb_row_sum(z, G) * inv_g;
      const float dv = z - mean;
      const float rstd = __builtin_amdgcn_rsqf(ddb_row_sum(dv * dv, G) * inv_g + 1e-8f);
      o[k * FG + tid] = ddb_prelu(dv * rstd * sml[2 * G + g] + sml[3 * G + g], p.alpha[k - 1]);
    }
    __syncthreads();
    roff += F * kG;
    woff += 6 * kG;
  }
  DDB_T(4);
  // ---- out conv over [prev_out ; o_6], then prev_out <- o_6
  {
    const ddb_f4 s = ddb_dense23(pout, o + 6 * FG, p.w_out, F, G, C, part, tid, nthreads);
    if (tid < FC / 4) {
      const int f = tid >> (lg - 1), cq = tid & ((C >> 2) - 1);
      ddb_f4 r;
#pragma unroll
      for (int j = 0; j < 4; ++j) r[j] = ddb_prelu(s[j] + p.b_out[4 * cq + j], p.a_out);
      *reinterpret_cast<ddb_f4*>(p.dst + soff + f * p.dst_ld + 4 * cq) = r;
      if (lds_y) *reinterpret_cast<ddb_f4*>(lds_y + f * lds_y_pitch + 4 * cq) = r;
    }
    for (int q = tid; q < FG; q += nthreads) pst_out[q] = o[6 * FG + q];
  }
  DDB_T(5);
  __syncthreads();
  DDB_T(6);
#undef DDB_T
}

}  // namespace nutls
