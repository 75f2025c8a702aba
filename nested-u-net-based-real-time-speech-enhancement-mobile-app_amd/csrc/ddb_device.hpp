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
//  The same block for the one-launch kernels (512 threads, ~70 KB of LDS scratch), built around the latency of
//  the chain in -> six blocks -> out (the arithmetic is tiny):
//    * every load that does not depend on this frame's arithmetic -- history rings, previous input / output rows,
//      the small weights of the six blocks, the K slice of the `in` conv's kernel -- is issued up front by all
//      threads (one memory latency for all of it); the `out` conv's slice is requested before the chain starts;
//    * the two dense (2,3) convs (`in`: C -> G, `out`: G -> C) are split over K across the whole workgroup;
//    * the six blocks run on the F*G threads that own an output (one wavefront when F*G = 64: no workgroup
//      barrier inside the chain, LDS is in order within a wave); a thread's grouped-conv kernel (6k floats) and
//      its row of the 1x1 kernel are contiguous in LDS; the current frame's o_6 .. o_0 sit in one row per
//      frequency bin, newest first, so in_k is simply its last k*G channels;
//    * the ring slots are rewritten once, after the chain.
// -------------------------------------------------------------------------------------------------
namespace nutls {

typedef float ddb_f4 __attribute__((ext_vector_type(4)));
typedef const ddb_f4 __attribute__((address_space(1)))* ddb_gf4;

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

// K-split dense (2,3) conv:  out[f][co] = sum_{t,kw,ci} W[t][kw][ci][co] * X_t[f+kw-1][ci],  W global [2][3][CI][CO].
// Thread (K slice ks, output float4 q): its <= 24 weight float4s are loaded ahead of use (ddb_dense_load), the products
// follow when the inputs are in LDS (ddb_dense_run: partial sums through `part`, one barrier, summed by the first F*CO/4
// threads).  F, CI, CO are powers of two: no integer divisions.
struct DdbDense { int lci, lnq, nks, epk, nq, q, ks, f, cq; };
__device__ __forceinline__ DdbDense ddb_dense_geom(int F, int CI, int CO, int tid, int nthreads) {
  DdbDense g;
  g.nq = (F * CO) >> 2;                         // float4 outputs
  int lcq = 0, lth = 0;
  g.lci = 0; g.lnq = 0;
  while ((1 << g.lci) < CI) ++g.lci;
  while ((1 << g.lnq) < g.nq) ++g.lnq;
  while ((4 << lcq) < CO) ++lcq;
  while ((2 << lth) <= nthreads) ++lth;         // floor(log2(nthreads))
  int lks = lth - g.lnq;                        // K slices: a power of two dividing 6*CI = 3 * 2^(lci+1), that fits the workgroup
  if (lks > g.lci + 1) lks = g.lci + 1;
  g.nks = 1 << lks;
  g.epk = (6 * CI) >> lks;                      // <= 24
  g.q = tid & (g.nq - 1);
  g.ks = tid >> g.lnq;
  g.f = g.q >> lcq;
  g.cq = g.q & ((CO >> 2) - 1);
  return g;
}
__device__ __forceinline__ void ddb_dense_load(const DdbDense& g, const float* W, int CO, ddb_f4 (&w)[24]) {
  if (g.ks < g.nks) {
#pragma unroll
    for (int j = 0; j < 24; ++j)
      if (j < g.epk) w[j] = *(ddb_gf4)(unsigned long long)(W + static_cast<size_t>(g.ks * g.epk + j) * CO + 4 * g.cq);
  }
}
// X0 / X1: LDS rows of the previous / current frame with pitches p0 / p1.  All threads call (one barrier inside).
__device__ __forceinline__ ddb_f4 ddb_dense_run(const DdbDense& g, const float* X0, int p0, const float* X1, int p1, int F, int CI,
                                                const ddb_f4 (&w)[24], float* part, int tid) {
  if (g.ks < g.nks) {
    ddb_f4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 24; ++j) {
      if (j < g.epk) {
        const int e = g.ks * g.epk + j;
        const int t = e >= 3 * CI ? 1 : 0, r = e - t * 3 * CI;
        const int kw = r >> g.lci, ci = r & (CI - 1);
        const int fr = g.f + kw - 1;
        const float xv = (fr >= 0 && fr < F) ? (t ? X1[fr * p1 + ci] : X0[fr * p0 + ci]) : 0.f;
        a += w[j] * xv;
      }
    }
    *reinterpret_cast<ddb_f4*>(part + (static_cast<size_t>(g.ks) * g.nq + g.q) * 4) = a;
  }
  __syncthreads();
  ddb_f4 s = {0.f, 0.f, 0.f, 0.f};
  if (tid < g.nq)
    for (int k2 = 0; k2 < g.nks; ++k2) s += *reinterpret_cast<const ddb_f4*>(part + (static_cast<size_t>(k2) * g.nq + tid) * 4);
  return s;
}

// first float4 of block k's ring / in_k rows in the flat [block][F][k*G] order: FG/4 * k(k-1)/2
__device__ __forceinline__ int ddb_blk_of(int q, int fg4) {
  return q < fg4 ? 1 : (q < 3 * fg4 ? 2 : (q < 6 * fg4 ? 3 : (q < 10 * fg4 ? 4 : (q < 15 * fg4 ? 5 : 6))));
}

// floats of the packed small-weight blob (engine.cpp prep_ddb_weights), in LDS order:
//   wg: block k [G][2][3][k] | w1: block k [G out][G in] | per block bg, b1, gamma, beta [4][G] | b_in [G] | b_out [2G]
__host__ __device__ constexpr int ddb_wsmall_floats(int G) { return 126 * G + 6 * G * G + 27 * G; }

// lds: >= 17.5K floats.  All NT threads (a multiple of 64, >= 512) call.
// lds_y (optional): the output rows are also written to LDS at lds_y + f * lds_y_pitch + c (the fused kernel's next image).
template <int NT>
__device__ __forceinline__ void ddb_block_wg(const DdbParams& p, int stream, float* lds, int tid,
                                             unsigned long long* dbg_lds = nullptr, float* lds_y = nullptr, int lds_y_pitch = 0) {
#define DDB_T(k) do { if (dbg_lds && (tid & 63) == 0) dbg_lds[(tid >> 6) * 16 + (k)] = wall_clock64(); } while (0)
  static_assert(NT >= 512 && NT % 64 == 0, "staging below is sized for >= 512 threads");
  DDB_T(0);
  const int F = p.F, C = p.C, G = C >> 1;
  const int FG = F * G, FC = F * C, G7 = 7 * G;
  int lg = 0;
  while ((1 << lg) < G) ++lg;                // G = 16 or 32, C = 2G, F = 1, 2 or 4 (run-time here; compile time in ddb_fused.hpp)
  float* xs = lds;                    // [F][C]    current input
  float* pin = xs + FC;               // [F][C]    previous input
  float* pout = pin + FC;             // [F][G]    previous o_6
  float* R = pout + FG;               // [F][7G]   this frame's o_6 | o_5 | ... | o_0  (in_k = the last k*G channels of a row)
  float* yv = R + 7 * FG;             // [F][G]
  float* rings = yv + FG;             // block k at rings + F*G*k(k-1)/2 : [F][k*G] (frame t-d)
  float* wgs = rings + 21 * FG;       // block k at wgs + 6*G*k(k-1)/2 : [G][2][3][k]
  float* w1s = wgs + 126 * G;         // [6][G out][G in]
  float* sm = w1s + 6 * G * G;        // [6][4][G] bg, b1, gamma, beta | b_in [G] | b_out [C]
  float* part = sm + 27 * G;          // K-split partial sums (<= 2048 floats)
  const size_t soff = static_cast<size_t>(stream) * p.sstride;
  const int step = *p.step;
  // (`p` lives in global memory and the block stores to global memory: every p.field the loops below touched
  //  would be re-read after each store -- take what the loops need into registers once)
  const float* const px = p.x + soff;
  float* const pst_in = p.st_in + soff;
  float* const pst_out = p.st_out + soff;
  float* const pdst = p.dst + soff;
  const int x_ld = p.x_ld, dst_ld = p.dst_ld;
  const float a_in = p.a_in, a_out = p.a_out;
  float alpha[6];
  float* rp[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    rp[k] = p.st_blk[k] + soff + static_cast<size_t>(step & ((1 << k) - 1)) * F * (k + 1) * G;
    alpha[k] = p.alpha[k];
  }
  const int fg4 = FG >> 2;
  // ---- phase A: every load that is already known, all in flight together
  {
    const DdbDense gi = ddb_dense_geom(F, C, G, tid, NT);
    ddb_f4 w[24];
    ddb_dense_load(gi, p.w_in, G, w);
    float vx = 0.f, vp = 0.f, vo = 0.f;
    if (tid < FC) {
      const int f = tid >> (lg + 1), c = tid & (C - 1);
      vx = px[f * x_ld + c];
      vp = pst_in[tid];
    }
    if (tid < FG) vo = pst_out[tid];
    constexpr int MAXW = (ddb_wsmall_floats(32) / 4 + NT - 1) / NT;      // 6 at 512 threads
    constexpr int MAXR = (21 * 128 / 4 + NT - 1) / NT;                   // 2
    const int n4 = ddb_wsmall_floats(G) >> 2, r4 = 21 * fg4;
    ddb_f4 ws[MAXW], rv[MAXR];
    const ddb_gf4 src = (ddb_gf4)(unsigned long long)p.wsmall;
#pragma unroll
    for (int i = 0; i < MAXW; ++i)
      if (tid + i * NT < n4) ws[i] = src[tid + i * NT];
#pragma unroll
    for (int i = 0; i < MAXR; ++i) {
      const int q = tid + i * NT;
      if (q < r4) {
        const int k = ddb_blk_of(q, fg4);
        const float* r = k == 1 ? rp[0] : (k == 2 ? rp[1] : (k == 3 ? rp[2] : (k == 4 ? rp[3] : (k == 5 ? rp[4] : rp[5]))));
        rv[i] = *(ddb_gf4)(unsigned long long)(r + 4 * (q - fg4 * (k * (k - 1) / 2)));
      }
    }
    if (tid < FC) { xs[tid] = vx; pin[tid] = vp; }
    if (tid < FG) pout[tid] = vo;
#pragma unroll
    for (int i = 0; i < MAXW; ++i)
      if (tid + i * NT < n4) reinterpret_cast<ddb_f4*>(wgs)[tid + i * NT] = ws[i];
#pragma unroll
    for (int i = 0; i < MAXR; ++i)
      if (tid + i * NT < r4) reinterpret_cast<ddb_f4*>(rings)[tid + i * NT] = rv[i];
    DDB_T(1);
    __syncthreads();
    DDB_T(2);
    if (tid < FC) pst_in[tid] = vx;        // prev_in <- x
    // ---- o_0 = PReLU(conv(2,3)([prev_in ; x]))  ->  channels [6G, 7G) of the rows
    const ddb_f4 s = ddb_dense_run(gi, pin, C, xs, C, F, C, w, part, tid);
    if (tid < fg4) {
      const int f = tid >> (lg - 2), cq = tid & ((G >> 2) - 1);
      ddb_f4 r;
#pragma unroll
      for (int j = 0; j < 4; ++j) r[j] = ddb_prelu(s[j] + sm[24 * G + 4 * cq + j], a_in);
      *reinterpret_cast<ddb_f4*>(R + f * G7 + 6 * G + 4 * cq) = r;
    }
  }
  // the `out` conv's K slice: requested now, used after the chain
  const DdbDense go = ddb_dense_geom(F, G, C, tid, NT);
  ddb_f4 wo[24];
  ddb_dense_load(go, p.w_out, C, wo);
  __syncthreads();
  DDB_T(3);
  // ---- blocks 1..6 (sequential: dense connectivity) on the threads that own an output
  const bool one_wave = FG <= 64;
  const int f = tid >> lg, g = tid & (G - 1);
  int roff = 0, woff = 0;
#pragma unroll
  for (int k = 1; k <= 6; ++k) {
    const int d = 1 << (k - 1), kG = k * G;
    const float* sml = sm + (k - 1) * 4 * G;
    if (tid < FG) {
      const float* wl = wgs + woff + g * 6 * k;            // [t][kw][j]
      float a = sml[g];
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int fr = f + (kw - 1) * d;
        if (fr >= 0 && fr < F) {
          const float* cur = R + fr * G7 + (7 - k) * G + g * k;      // in_k of this frame, channels g*k ..
          const float* old = rings + roff + fr * kG + g * k;         // in_k of frame t-d
#pragma unroll
          for (int j = 0; j < k; ++j) {
            a = fmaf(wl[kw * k + j], old[j], a);
            a = fmaf(wl[(3 + kw) * k + j], cur[j], a);
          }
        }
      }
      yv[tid] = a;
    }
    if (one_wave) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); else __syncthreads();
    if (tid < FG) {
      const ddb_f4* w1l = reinterpret_cast<const ddb_f4*>(w1s + (k - 1) * G * G + g * G);
      const ddb_f4* yr = reinterpret_cast<const ddb_f4*>(yv + f * G);
      float z0 = sml[G + g], z1 = 0.f;
      for (int i = 0; i < (G >> 2); i += 2) {
        const ddb_f4 wa = w1l[i], ya = yr[i], wb = w1l[i + 1], yb = yr[i + 1];
        z0 += wa[0] * ya[0] + wa[1] * ya[1] + wa[2] * ya[2] + wa[3] * ya[3];
        z1 += wb[0] * yb[0] + wb[1] * yb[1] + wb[2] * yb[2] + wb[3] * yb[3];
      }
      const float z = z0 + z1;
      // LayerNorm over the G channels of row f = G consecutive lanes
      const float inv_g = 1.0f / static_cast<float>(G);
      const float mean = ddb_row_sum(z, G) * inv_g;
      const float dv = z - mean;
      const float rstd = __builtin_amdgcn_rsqf(ddb_row_sum(dv * dv, G) * inv_g + 1e-8f);
      R[f * G7 + (6 - k) * G + g] = ddb_prelu(dv * rstd * sml[2 * G + g] + sml[3 * G + g], alpha[k - 1]);
    }
    if (one_wave) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); else __syncthreads();
    roff += F * kG;
    woff += 6 * kG;
  }
  __syncthreads();
  DDB_T(4);
  // ---- ring slots <- in_k of this frame (every read of the old slots went through LDS); prev_out <- o_6
  for (int q = tid; q < 21 * fg4; q += NT) {
    const int k = ddb_blk_of(q, fg4);
    const int kq = (k * G) >> 2;                                                      // float4s per row of block k
    const int local = q - fg4 * (k * (k - 1) / 2);
    const int fr = (local >= kq) + (local >= 2 * kq) + (local >= 3 * kq), c4 = local - fr * kq;      // F <= 4
    float* ring = k == 1 ? rp[0] : (k == 2 ? rp[1] : (k == 3 ? rp[2] : (k == 4 ? rp[3] : (k == 5 ? rp[4] : rp[5]))));
    *reinterpret_cast<ddb_f4*>(ring + 4 * local) = *reinterpret_cast<const ddb_f4*>(R + fr * G7 + (7 - k) * G + 4 * c4);
  }
  if (tid < FG) pst_out[tid] = R[f * G7 + g];
  // ---- out conv over [prev_out ; o_6]
  {
    const ddb_f4 s = ddb_dense_run(go, pout, G, R, G7, F, G, wo, part, tid);
    if (tid < FC / 4) {
      const int fo = tid >> (lg - 1), cq = tid & ((C >> 2) - 1);
      ddb_f4 r;
#pragma unroll
      for (int j = 0; j < 4; ++j) r[j] = ddb_prelu(s[j] + sm[25 * G + 4 * cq + j], a_out);
      *reinterpret_cast<ddb_f4*>(pdst + fo * dst_ld + 4 * cq) = r;
      if (lds_y) *reinterpret_cast<ddb_f4*>(lds_y + fo * lds_y_pitch + 4 * cq) = r;
    }
  }
  DDB_T(5);
  __syncthreads();
  DDB_T(6);
#undef DDB_T
}

}  // namespace nutls
