// Dilated-dense bottleneck (models/nunet_tls.py:277-359, streaming form converter_nunet_tls.py:374-411) as an op of
// the fused frame-step kernel: ddb_device.hpp's workgroup block with every size a compile-time constant
// (G = 16 or 32 channels per block, F = 1, 2 or 4 frequency bins) and the chain in -> six blocks -> out arranged
// around its latency -- the arithmetic is a few hundred FMAs per thread:
//   * all loads that do not depend on this frame (rings, previous input / output rows, the blocks' small weights, the
//     K slice of the `in` kernel) are requested two ops ahead and travel in the fused kernel's carry registers
//     (ddbz_prefetch); the `out` kernel's slice is requested before the chain;
//   * barriers wait for LDS only (the stores to the state tensors and the weight prefetch stay in flight);
//   * the input rows are read where the previous conv op left them (the next image in LDS);
//   * the six blocks run on the F*G threads that own an output.  Per block ONE LDS round trip: o_{k-1} is written,
//     then the taps of in_k (this frame's rows, the ring's rows) and the thread's weights are read back together
//     (taps outside the F bins are compiled out or masked, not branched over); the grouped conv's result stays in
//     a register, the 1x1 conv gathers the row's G values with DPP row rotations (the thread's kernel row is stored
//     pre-rotated: w1rot[g][n] = w1[g][(g - n) mod 16 ...]), LayerNorm sums are DPP row sums.
#pragma once
#include <cstddef>

#include "ddb_device.hpp"

namespace nutls {

__device__ __forceinline__ void ddbz_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ void ddbz_wave_sync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

constexpr int ddbz_log2(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }
constexpr int ddbz_floor_log2(int v) { int l = 0; while ((2 << l) <= v) ++l; return l; }
constexpr int ddbz_min(int a, int b) { return a < b ? a : b; }

// K-split dense (2,3) conv geometry (see ddb_dense_geom), compile time
template <int NT, int F, int CI, int CO>
struct DdbzDense {
  static constexpr int nq = F * CO / 4, lnq = ddbz_log2(nq), lci = ddbz_log2(CI), lcq = ddbz_log2(CO / 4);
  static constexpr int lks = ddbz_min(ddbz_floor_log2(NT) - lnq, lci + 1), nks = 1 << lks, epk = (6 * CI) >> lks;
  static_assert(epk >= 1 && epk <= 24 && nks * nq <= NT, "K split does not fit the workgroup");
};

template <int NT, int F, int CI, int CO>
__device__ __forceinline__ void ddbz_dense_load(const float* W, int tid, ddb_f4 (&w)[DdbzDense<NT, F, CI, CO>::epk]) {
  using D = DdbzDense<NT, F, CI, CO>;
  const int q = tid & (D::nq - 1), ks = tid >> D::lnq, cq = q & (CO / 4 - 1);
  if (ks < D::nks) {
#pragma unroll
    for (int j = 0; j < D::epk; ++j) w[j] = *(ddb_gf4)(unsigned long long)(W + static_cast<size_t>(ks * D::epk + j) * CO + 4 * cq);
  }
}
// X0 / X1: LDS rows of the previous / current frame (pitches P0 / P1).  One barrier inside; the sum is valid for tid < nq.
template <int NT, int F, int CI, int CO, int P0>
__device__ __forceinline__ ddb_f4 ddbz_dense_run(const float* X0, const float* X1, int p1, const ddb_f4 (&w)[DdbzDense<NT, F, CI, CO>::epk],
                                                 float* part, int tid) {
  using D = DdbzDense<NT, F, CI, CO>;
  const int q = tid & (D::nq - 1), ks = tid >> D::lnq, f = q >> D::lcq;
  if (ks < D::nks) {
    ddb_f4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < D::epk; ++j) {
      const int e = ks * D::epk + j;
      const int t = e >= 3 * CI ? 1 : 0, r = e - t * 3 * CI;
      const int kw = r >> D::lci, ci = r & (CI - 1);
      const int fr = f + kw - 1;
      const bool ok = fr >= 0 && fr < F;
      const int frc = ok ? fr : f;
      const float xv = t ? X1[frc * p1 + ci] : X0[frc * P0 + ci];
      a += w[j] * (ok ? xv : 0.f);
    }
    *reinterpret_cast<ddb_f4*>(part + (ks * D::nq + q) * 4) = a;
  }
  ddbz_barrier();
  ddb_f4 s = {0.f, 0.f, 0.f, 0.f};
  if (tid < D::nq) {
#pragma unroll 8
    for (int k2 = 0; k2 < D::nks; ++k2) s += *reinterpret_cast<const ddb_f4*>(part + (k2 * D::nq + tid) * 4);
  }
  return s;
}

// value of lane (i - N) mod 16 of the lane's 16-lane row
template <int N>
__device__ __forceinline__ float ddbz_ror(float v) {
  if constexpr (N == 0) return v;
  else return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x120 + N, 0xF, 0xF, false));
}
template <int N>
__device__ __forceinline__ void ddbz_mv16(float y, const float (&w)[16], float& z0, float& z1) {
  if constexpr (N < 16) {
    if constexpr (N & 1) z1 = fmaf(w[N], ddbz_ror<N>(y), z1);
    else z0 = fmaf(w[N], ddbz_ror<N>(y), z0);
    ddbz_mv16<N + 1>(y, w, z0, z1);
  }
}

template <int G>
constexpr int ddbz_lds_floats(int F) { return 2 * F * G + F * G + 7 * F * G + 21 * F * G + 126 * G + 6 * G * G + 27 * G + 2048; }

// What the op's loads that do not depend on this frame occupy in the fused kernel's carry (float4 slots per thread):
// [K slice of the `in` kernel, when it is small] [small weights] [ring rows] [previous input, previous o_6]
template <int NT, int G, int F>
struct DdbzCarry {
  static constexpr int epk = DdbzDense<NT, F, 2 * G, G>::epk;
  static constexpr int WI = epk <= 8 ? epk : 0;                 // (the central block's 24-slot slice is loaded in the op)
  static constexpr int n_wg = 126 * G / 4, n_w1 = 6 * G * G / 4, n_sm = 27 * G / 4, n4 = n_wg + n_w1 + n_sm;
  static constexpr int MAXW = (n4 + NT - 1) / NT, MAXR = (21 * F * G / 4 + NT - 1) / NT;
  static constexpr int N = WI + MAXW + MAXR + 1;
};

// The op's parameter record (DdbParams, global memory) read with SCALAR loads, one round trip for all of it.  Left to
// the compiler these are vector loads -- the kernel stores to global memory, so nothing is provably unclobbered -- each
// followed by `s_waitcnt vmcnt(0)` because the next address depends on it: five dependent memory round trips (~3 us)
// at the top of the op.  (The records are never written by the kernel: the scalar cache is coherent for them.)
struct DdbzRec {
  const float *w_in, *w_out, *wsmall;
  float *st_in, *st_out, *dst, *st_blk[6];
  float a_in, a_out, alpha[6];
  int dst_ld;
  long long sstride;
};
#define DDBZ_OFF(f) static_cast<int>(offsetof(DdbParams, f))
__device__ __forceinline__ DdbzRec ddbz_load_rec(const DdbParams* gp) {
  unsigned long long q[12];
  asm volatile(
      "s_load_dwordx2 %0, %12, %13\n\ts_load_dwordx2 %1, %12, %14\n\ts_load_dwordx2 %2, %12, %15\n\t"
      "s_load_dwordx2 %3, %12, %16\n\ts_load_dwordx2 %4, %12, %17\n\ts_load_dwordx2 %5, %12, %18\n\t"
      "s_load_dwordx2 %6, %12, %19\n\ts_load_dwordx2 %7, %12, %20\n\ts_load_dwordx2 %8, %12, %21\n\t"
      "s_load_dwordx2 %9, %12, %22\n\ts_load_dwordx2 %10, %12, %23\n\ts_load_dwordx2 %11, %12, %24\n\t"
      "s_waitcnt lgkmcnt(0)"
      : "=&s"(q[0]), "=&s"(q[1]), "=&s"(q[2]), "=&s"(q[3]), "=&s"(q[4]), "=&s"(q[5]), "=&s"(q[6]), "=&s"(q[7]), "=&s"(q[8]),
        "=&s"(q[9]), "=&s"(q[10]), "=&s"(q[11])
      : "s"(gp), "n"(DDBZ_OFF(w_in)), "n"(DDBZ_OFF(w_out)), "n"(DDBZ_OFF(wsmall)), "n"(DDBZ_OFF(st_in)), "n"(DDBZ_OFF(st_out)),
        "n"(DDBZ_OFF(dst)), "n"(DDBZ_OFF(st_blk[0])), "n"(DDBZ_OFF(st_blk[1])), "n"(DDBZ_OFF(st_blk[2])), "n"(DDBZ_OFF(st_blk[3])),
        "n"(DDBZ_OFF(st_blk[4])), "n"(DDBZ_OFF(st_blk[5])));
  unsigned long long ss;
  unsigned f[9];
  asm volatile(
      "s_load_dword %0, %10, %11\n\ts_load_dword %1, %10, %12\n\ts_load_dword %2, %10, %13\n\ts_load_dword %3, %10, %14\n\t"
      "s_load_dword %4, %10, %15\n\ts_load_dword %5, %10, %16\n\ts_load_dword %6, %10, %17\n\ts_load_dword %7, %10, %18\n\t"
      "s_load_dword %8, %10, %19\n\ts_load_dwordx2 %9, %10, %20\n\t"
      "s_waitcnt lgkmcnt(0)"
      : "=&s"(f[0]), "=&s"(f[1]), "=&s"(f[2]), "=&s"(f[3]), "=&s"(f[4]), "=&s"(f[5]), "=&s"(f[6]), "=&s"(f[7]), "=&s"(f[8]), "=&s"(ss)
      : "s"(gp), "n"(DDBZ_OFF(a_in)), "n"(DDBZ_OFF(a_out)), "n"(DDBZ_OFF(alpha[0])), "n"(DDBZ_OFF(alpha[1])), "n"(DDBZ_OFF(alpha[2])),
        "n"(DDBZ_OFF(alpha[3])), "n"(DDBZ_OFF(alpha[4])), "n"(DDBZ_OFF(alpha[5])), "n"(DDBZ_OFF(dst_ld)), "n"(DDBZ_OFF(sstride)));
  DdbzRec r;
  r.w_in = reinterpret_cast<const float*>(q[0]); r.w_out = reinterpret_cast<const float*>(q[1]); r.wsmall = reinterpret_cast<const float*>(q[2]);
  r.st_in = reinterpret_cast<float*>(q[3]); r.st_out = reinterpret_cast<float*>(q[4]); r.dst = reinterpret_cast<float*>(q[5]);
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    r.st_blk[k] = reinterpret_cast<float*>(q[6 + k]);
    r.alpha[k] = __builtin_bit_cast(float, f[2 + k]);
  }
  r.a_in = __builtin_bit_cast(float, f[0]);
  r.a_out = __builtin_bit_cast(float, f[1]);
  r.dst_ld = static_cast<int>(f[8]);
  r.sstride = static_cast<long long>(ss);
  return r;
}
#undef DDBZ_OFF

__device__ __forceinline__ float* ddbz_ring(const DdbzRec& p, size_t soff, int step, int k /* 0..5 */, int F, int G) {
  return p.st_blk[k] + soff + static_cast<size_t>(step & ((1 << k) - 1)) * F * (k + 1) * G;
}

// Issues those loads (any time after the previous frame's launch: the rings are only rewritten by this op itself).
template <int NT, int G, int F>
__device__ __forceinline__ void ddbz_prefetch(const DdbzRec& p, int stream, int step, int tid, ddb_f4 (&pre)[DdbzCarry<NT, G, F>::N]) {
  using K = DdbzCarry<NT, G, F>;
  constexpr int C = 2 * G, FG = F * G, FC = F * C, fg4 = FG / 4;
  const size_t soff = static_cast<size_t>(stream) * p.sstride;
  if constexpr (K::WI > 0) {
    ddb_f4 wi[K::epk];
    ddbz_dense_load<NT, F, C, G>(p.w_in, tid, wi);
#pragma unroll
    for (int j = 0; j < K::epk; ++j) pre[j] = wi[j];
  }
  const ddb_gf4 src = (ddb_gf4)(unsigned long long)p.wsmall;      // global blob = wg | w1 (plain) | sm | w1 (rotated); LDS = wg | w1 rotated | sm
#pragma unroll
  for (int i = 0; i < K::MAXW; ++i) {
    const int q = tid + i * NT;
    if (q < K::n4) pre[K::WI + i] = src[(q >= K::n_wg && q < K::n_wg + K::n_w1) ? q + K::n_w1 + K::n_sm : q];      // (sm sits at the same offset in both)
  }
  const float* rp[6];          // (scalar registers; indexing the record with a per-lane k would be a dependent vector load)
#pragma unroll
  for (int k = 0; k < 6; ++k) rp[k] = ddbz_ring(p, soff, step, k, F, G);
#pragma unroll
  for (int i = 0; i < K::MAXR; ++i) {
    const int q = tid + i * NT;
    if (q < 21 * fg4) {
      const int k = ddb_blk_of(q, fg4);
      const float* r = k == 1 ? rp[0] : (k == 2 ? rp[1] : (k == 3 ? rp[2] : (k == 4 ? rp[3] : (k == 5 ? rp[4] : rp[5]))));
      pre[K::WI + K::MAXW + i] = *(ddb_gf4)(unsigned long long)(r + 4 * (q - fg4 * (k * (k - 1) / 2)));
    }
  }
  if (tid < FC) pre[K::N - 1][0] = (p.st_in + soff)[tid];
  if (tid < FG) pre[K::N - 1][1] = (p.st_out + soff)[tid];
}

// lds: ddbz_lds_floats<G>(F) floats of scratch.  xs: the block's input rows as fp32 in LDS (pitch lds_pitch floats; written by
// the conv op before, visible after a barrier); put_y(row, float4 index, value): the block's output into the next conv's image
// (whatever its format).  `step`: the frame counter (ring position).
template <int NT, int G, int F, class PutY>
__device__ __forceinline__ void ddb_block_fz(const DdbzRec& p, int stream, int step, float* lds, int tid, unsigned long long* dbg_lds,
                                             const float* xs, int lds_pitch, PutY&& put_y, const ddb_f4 (&pre)[DdbzCarry<NT, G, F>::N]) {
#define DDBZ_T(k) do { if (dbg_lds && (tid & 63) == 0) dbg_lds[(tid >> 6) * 16 + (k)] = wall_clock64(); } while (0)
  static_assert((G == 16 || G == 32) && (F == 1 || F == 2 || F == 4) && NT >= 512 && NT % 64 == 0, "unsupported block shape");
  constexpr int C = 2 * G, FG = F * G, FC = F * C, G7 = 7 * G, lg = ddbz_log2(G), fg4 = FG / 4;
  DDBZ_T(0);
  float* pin = lds;                   // [F][C]    previous input
  float* pout = pin + FC;             // [F][G]    previous o_6
  float* R = pout + FG;               // [F][7G]   this frame's o_6 | o_5 | ... | o_0  (in_k = the last k*G channels of a row)
  float* rings = R + 7 * FG;          // block k at rings + F*G*k(k-1)/2 : [F][k*G] (frame t-d)
  float* wgs = rings + 21 * FG;       // block k at wgs + 6*G*k(k-1)/2 : [G][2][3][k]
  float* w1r = wgs + 126 * G;         // [6][G out][G] pre-rotated rows of the 1x1 kernels
  float* sm = w1r + 6 * G * G;        // [6][4][G] bg, b1, gamma, beta | b_in [G] | b_out [C]
  float* part = sm + 27 * G;          // K-split partial sums (<= 2048 floats)
  const size_t soff = static_cast<size_t>(stream) * p.sstride;
  float* const pst_in = p.st_in + soff;
  float* const pst_out = p.st_out + soff;
  float* const pdst = p.dst + soff;
  const int dst_ld = p.dst_ld;
  const float a_in = p.a_in, a_out = p.a_out;
  float alpha[6];
  float* rp[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    rp[k] = ddbz_ring(p, soff, step, k, F, G);
    alpha[k] = p.alpha[k];
  }
  // ---- phase A: what was requested ahead (ddbz_prefetch) goes to LDS
  using DI = DdbzDense<NT, F, C, G>;
  using DO = DdbzDense<NT, F, G, C>;
  using K = DdbzCarry<NT, G, F>;
  {
    ddb_f4 wi[DI::epk];
    if constexpr (K::WI > 0) {
#pragma unroll
      for (int j = 0; j < DI::epk; ++j) wi[j] = pre[j];
    } else {
      ddbz_dense_load<NT, F, C, G>(p.w_in, tid, wi);
    }
    if (tid < FC) pin[tid] = pre[K::N - 1][0];
    if (tid < FG) pout[tid] = pre[K::N - 1][1];
#pragma unroll
    for (int i = 0; i < K::MAXW; ++i)
      if (tid + i * NT < K::n4) reinterpret_cast<ddb_f4*>(wgs)[tid + i * NT] = pre[K::WI + i];
#pragma unroll
    for (int i = 0; i < K::MAXR; ++i)
      if (tid + i * NT < 21 * fg4) reinterpret_cast<ddb_f4*>(rings)[tid + i * NT] = pre[K::WI + K::MAXW + i];
    DDBZ_T(1);
    ddbz_barrier();
    DDBZ_T(2);
    if (tid < FC) pst_in[tid] = xs[(tid >> (lg + 1)) * lds_pitch + (tid & (C - 1))];        // prev_in <- x
    // ---- o_0 = PReLU(conv(2,3)([prev_in ; x]))  ->  channels [6G, 7G) of the rows
    const ddb_f4 s = ddbz_dense_run<NT, F, C, G, C>(pin, xs, lds_pitch, wi, part, tid);
    if (tid < fg4) {
      const int f = tid >> (lg - 2), cq = tid & (G / 4 - 1);
      ddb_f4 r;
#pragma unroll
      for (int j = 0; j < 4; ++j) r[j] = ddb_prelu(s[j] + sm[24 * G + 4 * cq + j], a_in);
      *reinterpret_cast<ddb_f4*>(R + f * G7 + 6 * G + 4 * cq) = r;
    }
  }
  // the `out` conv's K slice: requested now, used after the chain
  ddb_f4 wo[DO::epk];
  ddbz_dense_load<NT, F, G, C>(p.w_out, tid, wo);
  ddbz_barrier();
  DDBZ_T(3);
  // ---- blocks 1..6 (sequential: dense connectivity) on the threads that own an output
  constexpr bool one_wave = FG <= 64;
  const int f = tid >> lg, g = tid & (G - 1);
  if (one_wave ? tid < FG : true) {
    const bool own = tid < FG;
#pragma unroll
    for (int k = 1; k <= 6; ++k) {
      const int d = 1 << (k - 1), kG = k * G, roff = FG * (k * (k - 1) / 2), woff = 6 * G * (k * (k - 1) / 2);
      const float* sml = sm + (k - 1) * 4 * G;
      float z = 0.f;
      if (own) {
        const float* wl = wgs + woff + g * 6 * k;            // [t][kw][j]
        float a0 = sml[g], a1 = 0.f;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const int off = (kw - 1) * d;
          if (off > -F && off < F) {                           // (compile time after unrolling: dilations >= F keep the centre tap only)
            const int fr = f + off;
            const bool ok = fr >= 0 && fr < F;
            const int frc = ok ? fr : f;
            const float m = ok ? 1.0f : 0.0f;
            const float* cur = R + frc * G7 + (7 - k) * G + g * k;        // in_k of this frame, channels g*k ..
            const float* old = rings + roff + frc * kG + g * k;           // in_k of frame t-d
#pragma unroll
            for (int j = 0; j < k; ++j) {
              a0 = fmaf(m * wl[kw * k + j], old[j], a0);
              a1 = fmaf(m * wl[(3 + kw) * k + j], cur[j], a1);
            }
          }
        }
        const float y = a0 + a1;
        // 1x1 conv over the row's G channels: DPP rotations of y against the pre-rotated kernel row
        const ddb_f4* w4 = reinterpret_cast<const ddb_f4*>(w1r + (k - 1) * G * G + g * G);
        float wv[16];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const ddb_f4 t4 = w4[i];
          wv[4 * i] = t4[0]; wv[4 * i + 1] = t4[1]; wv[4 * i + 2] = t4[2]; wv[4 * i + 3] = t4[3];
        }
        float z0 = sml[G + g], z1 = 0.f;
        ddbz_mv16<0>(y, wv, z0, z1);
        if constexpr (G == 32) {
          const float yo = __shfl_xor(y, 16);                  // the other half of the row
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const ddb_f4 t4 = w4[4 + i];
            wv[4 * i] = t4[0]; wv[4 * i + 1] = t4[1]; wv[4 * i + 2] = t4[2]; wv[4 * i + 3] = t4[3];
          }
          ddbz_mv16<0>(yo, wv, z0, z1);
        }
        z = z0 + z1;
      }
      // LayerNorm over the G channels of row f = G consecutive lanes (all lanes of the wave run the DPP sums)
      constexpr float inv_g = 1.0f / static_cast<float>(G);
      const float mean = ddb_row_sum(z, G) * inv_g;
      const float dv = z - mean;
      const float rstd = __builtin_amdgcn_rsqf(ddb_row_sum(dv * dv, G) * inv_g + 1e-8f);
      if (own) R[f * G7 + (6 - k) * G + g] = ddb_prelu(dv * rstd * sml[2 * G + g] + sml[3 * G + g], alpha[k - 1]);
      if constexpr (one_wave) ddbz_wave_sync(); else ddbz_barrier();
    }
  }
  ddbz_barrier();
  DDBZ_T(4);
  // ---- ring slots <- in_k of this frame (every read of the old slots went through LDS); prev_out <- o_6
#pragma unroll
  for (int i = 0; i < (21 * fg4 + NT - 1) / NT; ++i) {
    const int q = tid + i * NT;
    if (q < 21 * fg4) {
      const int k = ddb_blk_of(q, fg4);
      const int kq = (k * G) >> 2;                                                      // float4s per row of block k
      const int local = q - fg4 * (k * (k - 1) / 2);
      const int fr = (local >= kq) + (local >= 2 * kq) + (local >= 3 * kq), c4 = local - fr * kq;      // F <= 4
      float* ring = k == 1 ? rp[0] : (k == 2 ? rp[1] : (k == 3 ? rp[2] : (k == 4 ? rp[3] : (k == 5 ? rp[4] : rp[5]))));
      *reinterpret_cast<ddb_f4*>(ring + 4 * local) = *reinterpret_cast<const ddb_f4*>(R + fr * G7 + (7 - k) * G + 4 * c4);
    }
  }
  if (tid < FG) pst_out[tid] = R[f * G7 + g];
  // ---- out conv over [prev_out ; o_6]
  {
    const ddb_f4 s = ddbz_dense_run<NT, F, G, C, G>(pout, R, G7, wo, part, tid);
    if (tid < FC / 4) {
      const int fo = tid >> (lg - 1), cq = tid & (C / 4 - 1);
      ddb_f4 r;
#pragma unroll
      for (int j = 0; j < 4; ++j) r[j] = ddb_prelu(s[j] + sm[25 * G + 4 * cq + j], a_out);
      *reinterpret_cast<ddb_f4*>(pdst + fo * dst_ld + 4 * cq) = r;
      put_y(fo, cq, r);
    }
  }
  DDBZ_T(5);
  ddbz_barrier();
  DDBZ_T(6);
#undef DDBZ_T
}

}  // namespace nutls
