// gfx950 (MI355X / CDNA4) kernels of the NUNet-TLS-LSTM frame step.  wave = 64 lanes.
//
// Data layout in HBM: every activation is channels-last fp32 rows, [stream][frequency][channel];
// a "row" is the C channels of one (stream, frequency bin).  All conv-like layers
// (reference: models/proposed.py:198-265) are evaluated as output-stationary GEMMs on the
// fp32 matrix cores, v_mfma_f32_32x32x2_f32 (exact fp32: bitwise an fmaf chain):
//
//     D[ch, pos] += W[ch, k] * X[k, pos]        ch -> MFMA rows (A operand = weights)
//                                               pos -> MFMA columns = lanes (B operand = activations)
//     k = (time tap, frequency tap, input channel)
//
// With positions on lanes, the 32 (or 64) channels of one position sit in the 16 accumulator
// registers of lanes l and l+32, so the per-position LayerNorm (over channels) is a register
// reduction plus ONE cross-lane exchange, and the result leaves as 16-byte channels-last stores.
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdlib>

#include "nutls_internal.hpp"
#include "ddb_device.hpp"

namespace nutls {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define LN_EPS 1e-8f   // models/proposed.py:202  LayerNormalization(epsilon=1e-8)

// ================================================================================================
//  Generic tap-GEMM convolution
//
//  One wave owns 32 consecutive output positions (flattened (stream, f_out)) x 32*NT output
//  channels.  A workgroup of NW waves stages, per (time tap, 64-channel chunk), the input rows its
//  32*NW positions touch into LDS (16-byte coalesced global reads, rows zero-filled outside the
//  stream = the ZeroPadding2D of proposed.py:210/:242), then every lane reads its B fragment with one
//  ds_read_b128 per 8 input channels:  lane (pos p, half h) gets channels 8g+4h..8g+4h+3 of row
//  (s*p + kf), which feed 4 MFMA k-steps.  The LDS row pitch CC+4 (stride 1) or 2*CC+4 per row
//  *pair* (stride 2: rows 2r and 2r+1 share a pitch unit) makes the 16-lane ds_read_b128 groups
//  hit 16 distinct 4-bank slots.  Weights stream from L2 in fragment order (1 KiB per wave load).
// ================================================================================================
template <int CIN, int NT, int STRIDE, int TT, int KF, int PADL, int EPI_LN, int G, int NW>
__global__ __launch_bounds__(64 * NW) void conv_mfma_kernel(const ConvParams p) {
  constexpr int CC = CIN < 64 ? CIN : 64;
  constexpr int NCH = CIN / CC;
  constexpr int TP = 32 * NW;
  constexpr int PITCH = (STRIDE == 1) ? (CC + 4) : (2 * CC + 4);
  constexpr int R = NT / G;
  static_assert(NT % G == 0, "groups must tile the channel tiles");
  extern __shared__ __attribute__((aligned(16))) float lds[];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int pl = lane & 31, h = lane >> 5;
  const int F_out = p.F_out, log2f = p.log2_fout;
  const int seg_len = F_out < TP ? F_out : TP;          // power of two
  const int nseg = TP / seg_len;
  const int RS = (STRIDE == 1) ? (seg_len + KF - 1) : (seg_len + (KF - 1) / 2);  // LDS rows(-pairs)/segment
  const int LR = (STRIDE == 1) ? RS : 2 * RS;                                    // input rows staged/segment
  const int P0 = blockIdx.x * TP;
  const int total_pos = p.B << log2f;

  const int ploc = wave * 32 + pl;
  const int myseg = ploc / seg_len, myfl = ploc - myseg * seg_len;
  const int lbase = (myseg * RS + myfl) * PITCH + 4 * h;

  f32x16 acc[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;

  const f32x4* wp = reinterpret_cast<const f32x4*>(p.wpk) + lane;
  const int rows_total = nseg * LR;

#pragma unroll 1
  for (int t = 0; t < TT; ++t) {
    const float* src = (TT == 2 && t == 1) ? p.src1 : p.src0;
#pragma unroll 1
    for (int ch = 0; ch < NCH; ++ch) {
      if (t + ch > 0) __syncthreads();   // all waves finished reading the previous phase
      // ---------------- stage input rows -> LDS -----------------------------------------------
      for (int q = tid; q < rows_total * (CC / 4); q += 64 * NW) {
        const int c4 = q % (CC / 4);
        const int rr = q / (CC / 4);
        const int sg = rr / LR, lr = rr - sg * LR;
        const int pseg = P0 + sg * seg_len;
        const int b = pseg >> log2f, f0 = pseg & (F_out - 1);
        const int gr = STRIDE * f0 - PADL + lr;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (b < p.B && gr >= 0 && gr < p.F_in)
          v = *reinterpret_cast<const f32x4*>(src + slot_of(b, p.sm) * p.sstride + static_cast<size_t>(gr) * p.src_ld + ch * CC + 4 * c4);
        const int la = (STRIDE == 1) ? ((sg * RS + lr) * PITCH + 4 * c4)
                                     : ((sg * RS + (lr >> 1)) * PITCH + (lr & 1) * CC + 4 * c4);
        *reinterpret_cast<f32x4*>(lds + la) = v;
      }
      __syncthreads();
      // ---------------- MFMA over (frequency tap, channel group) --------------------------------
#pragma unroll
      for (int kf = 0; kf < KF; ++kf) {
        const int koff = (STRIDE == 1) ? (kf * PITCH) : ((kf >> 1) * PITCH + (kf & 1) * CC);
#pragma unroll
        for (int g = 0; g < CC / 8; ++g) {
          const f32x4 b4 = *reinterpret_cast<const f32x4*>(lds + lbase + koff + 8 * g);
          f32x4 a4[NT];
#pragma unroll
          for (int n = 0; n < NT; ++n) a4[n] = wp[n * 64];
          wp += NT * 64;
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int n = 0; n < NT; ++n)
              acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[n][j], b4[j], acc[n], 0, 0, 0);
        }
      }
    }
  }

  // ---------------- epilogue: bias (+ LayerNorm over the group's channels + PReLU), store ------
  // accumulator register r of lane (pos, h) holds channel  tile*32 + 8*(r>>2) + 4*h + (r&3)
  const int P = P0 + ploc;
  const bool valid = P < total_pos;
  const int b = P >> log2f, f = P & (F_out - 1);
  const size_t soff = slot_of(b, p.sm) * p.sstride;
  const size_t row0 = static_cast<size_t>(f) * p.row_mul + p.row_add;
#pragma unroll
  for (int gi = 0; gi < R; ++gi) {
    float v[G][16];
#pragma unroll
    for (int tg = 0; tg < G; ++tg) {
      const int n = gi * G + tg;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 bb = *reinterpret_cast<const f32x4*>(p.bias + n * 32 + 8 * q + 4 * h);
#pragma unroll
        for (int i = 0; i < 4; ++i) v[tg][4 * q + i] = acc[n][4 * q + i] + bb[i];
      }
    }
    if (EPI_LN) {
      constexpr float inv_n = 1.0f / (32 * G);
      float s = 0.f;
#pragma unroll
      for (int tg = 0; tg < G; ++tg)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += v[tg][r];
      s += __shfl_xor(s, 32);
      const float mean = s * inv_n;
      float qv = 0.f;
#pragma unroll
      for (int tg = 0; tg < G; ++tg)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          v[tg][r] -= mean;
          qv += v[tg][r] * v[tg][r];
        }
      qv += __shfl_xor(qv, 32);
      const float rstd = 1.0f / sqrtf(qv * inv_n + LN_EPS);
#pragma unroll
      for (int tg = 0; tg < G; ++tg)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 gm = *reinterpret_cast<const f32x4*>(p.gamma + tg * 32 + 8 * q + 4 * h);
          const f32x4 bt = *reinterpret_cast<const f32x4*>(p.beta + tg * 32 + 8 * q + 4 * h);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float y = v[tg][4 * q + i] * rstd * gm[i] + bt[i];
            v[tg][4 * q + i] = y >= 0.f ? y : p.alpha * y;
          }
        }
    }
    if (valid) {
      const size_t row = row0 + gi;
#pragma unroll
      for (int tg = 0; tg < G; ++tg)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 o = {v[tg][4 * q], v[tg][4 * q + 1], v[tg][4 * q + 2], v[tg][4 * q + 3]};
          const int c = tg * 32 + 8 * q + 4 * h;
          *reinterpret_cast<f32x4*>(p.dst0 + soff + row * p.ld0 + c) = o;
          if (p.dst1) *reinterpret_cast<f32x4*>(p.dst1 + soff + row * p.ld1 + c) = o;
        }
    }
  }
}

// ================================================================================================
//  The same tap-GEMM on the bf16 matrix pipe (offline / block mode, int8 containers): fp32 results from three bf16 MFMAs per
//  product.  The weights are the container's int8 values (exact in bf16; the per-channel scale is applied to the fp32 sums
//  in the epilogue), every fp32 activation is split ERROR-FREE into three bf16 pieces when it is staged into LDS
//  (x = hi + mid + lo: 8 + 8 + 8 significand bits, see fused_step.hip), so every product w * piece is exact in fp32 and is
//  accumulated in fp32 -- the quantity an fp32 FMA chain rounds, at 16/3 of the fp32 MFMA's rate.
//  LDS row (stride 1) or row pair (stride 2): [plane hi | mid | lo][channel] bf16 + 16 bytes of padding (pitch = an odd
//  number of 16-byte slots); lane (pos p, half h) reads the 8 channels 16 g + 8 h .. + 7 of a K step with one ds_read_b128
//  per plane; v_mfma_f32_32x32x16_bf16 (A = weights of channel tile n, B = positions).
// ================================================================================================
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned kb_bits(float x) { return __builtin_bit_cast(unsigned, x); }
__device__ __forceinline__ float kb_float(unsigned b) { return __builtin_bit_cast(float, b); }
__device__ __forceinline__ unsigned kb_top16(unsigned a, unsigned b) { return __builtin_amdgcn_perm(b, a, 0x07060302u); }   // {top16(a), top16(b)}
__device__ __forceinline__ void kb_split4(const f32x4& v, u32x2& hi, u32x2& mid, u32x2& lo) {
  unsigned xb[4], rb[4], lb[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float x = v[e];
    xb[e] = kb_bits(x);
    const float r = x - kb_float(xb[e] & 0xffff0000u);
    rb[e] = kb_bits(r);
    lb[e] = kb_bits(r - kb_float(rb[e] & 0xffff0000u));
  }
  hi = u32x2{kb_top16(xb[0], xb[1]), kb_top16(xb[2], xb[3])};
  mid = u32x2{kb_top16(rb[0], rb[1]), kb_top16(rb[2], rb[3])};
  lo = u32x2{kb_top16(lb[0], lb[1]), kb_top16(lb[2], lb[3])};
}

template <int CIN, int NT, int STRIDE, int TT, int KF, int PADL, int EPI_LN, int G, int NW, bool ALL>
__global__ __launch_bounds__(64 * NW) void conv_bf16x3_kernel(const ConvParams p) {
  constexpr int CC = CIN < 64 ? CIN : 64;
  constexpr int NCH = CIN / CC;
  // ALL with four waves = K split: the four waves share ONE tile of 32 positions, wave w runs K steps w, w + 4, ... of all phases
  // (a quarter of the weights and of the MFMAs each, 256 threads stage the rows), the partial sums meet in LDS and wave 0
  // runs the epilogue -- the small layers are one latency chain per wave, this shortens the chain instead of widening the tile
  constexpr bool KS = ALL && NW == 4;
  constexpr int TP = KS ? 32 : 32 * NW;
  constexpr int PLANE_B = ((STRIDE == 1) ? CC : 2 * CC) * 2;      // bytes of one plane of a pitch unit
  constexpr int PITCH_B = 3 * PLANE_B + 16;
  constexpr int R = NT / G;
  static_assert(NT % G == 0, "groups must tile the channel tiles");
  static_assert(CC % 16 == 0, "a K step is 16 channels");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  char* const ldsb = reinterpret_cast<char*>(lds);

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int pl = lane & 31, h = lane >> 5;
  const int F_out = p.F_out, log2f = p.log2_fout;
  const int seg_len = F_out < TP ? F_out : TP;          // power of two
  const int nseg = TP / seg_len;
  const int RS = (STRIDE == 1) ? (seg_len + KF - 1) : (seg_len + (KF - 1) / 2);  // LDS rows(-pairs)/segment
  const int LR = (STRIDE == 1) ? RS : 2 * RS;                                    // input rows staged/segment
  const int P0 = blockIdx.x * TP;
  const int total_pos = p.B << log2f;

  const int ploc = KS ? pl : wave * 32 + pl;
  const int myseg = ploc / seg_len, myfl = ploc - myseg * seg_len;
  const int lbase = (myseg * RS + myfl) * PITCH_B + 16 * h;

  f32x16 acc[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;

  const f32x4* wp = reinterpret_cast<const f32x4*>(p.wbf) + lane;      // one fragment = 8 bf16 = 16 bytes per lane
  const int rows_total = nseg * LR;
  constexpr int WPH = KF * (CC / 16) * NT;      // weight fragments of one (time tap, channel chunk) phase
  constexpr int NPH = TT * NCH;

  // one phase = one (time tap, 64-channel chunk): its input rows -> LDS at byte offset `lo`, split into the three planes
  auto stage = [&](int t, int ch, int lo) {
    const float* src = (TT == 2 && t == 1) ? p.src1 : p.src0;
    // eight loads in flight per thread, then their splits and stores: one load per loop trip (load -> split -> store) paid a full
    // memory round trip per trip -- a dozen trips in the 1-wave workgroups of the small layers
    constexpr int U = 8;
    const int n_items = rows_total * (CC / 4);
    for (int q0 = tid; q0 < n_items; q0 += 64 * NW * U) {
      f32x4 v[U];
      int la[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int q = q0 + u * 64 * NW;
        const int c4 = q % (CC / 4);
        const int rr = q / (CC / 4);
        const int sg = rr / LR, lr = rr - sg * LR;
        const int pseg = P0 + sg * seg_len;
        const int b = pseg >> log2f, f0 = pseg & (F_out - 1);
        const int gr = STRIDE * f0 - PADL + lr;
        v[u] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (q < n_items && b < p.B && gr >= 0 && gr < p.F_in)
          v[u] = *reinterpret_cast<const f32x4*>(src + slot_of(b, p.sm) * p.sstride + static_cast<size_t>(gr) * p.src_ld + ch * CC + 4 * c4);
        la[u] = q < n_items ? lo + ((STRIDE == 1) ? ((sg * RS + lr) * PITCH_B + 8 * c4)
                                                  : ((sg * RS + (lr >> 1)) * PITCH_B + (lr & 1) * (CC * 2) + 8 * c4))
                            : -1;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (la[u] >= 0) {
          u32x2 hi, mid, lo3;
          kb_split4(v[u], hi, mid, lo3);
          *reinterpret_cast<u32x2*>(ldsb + la[u]) = hi;
          *reinterpret_cast<u32x2*>(ldsb + la[u] + PLANE_B) = mid;
          *reinterpret_cast<u32x2*>(ldsb + la[u] + 2 * PLANE_B) = lo3;
        }
      }
    }
  };
  // MFMA over (frequency tap, K step of 16 channels) of one phase; wreg: its WPH weight fragments
  auto mfma_phase = [&](int lo, const f32x4* wreg) {
#pragma unroll
    for (int kf = 0; kf < KF; ++kf) {
      const int koff = lo + ((STRIDE == 1) ? (kf * PITCH_B) : ((kf >> 1) * PITCH_B + (kf & 1) * (CC * 2)));
#pragma unroll
      for (int g = 0; g < CC / 16; ++g) {
        bf16x8 b3[3];
#pragma unroll
        for (int q = 0; q < 3; ++q)
          b3[q] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4*>(ldsb + lbase + koff + 32 * g + q * PLANE_B));
        bf16x8 a8[NT];
#pragma unroll
        for (int n = 0; n < NT; ++n) a8[n] = __builtin_bit_cast(bf16x8, wreg[(kf * (CC / 16) + g) * NT + n]);
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
          for (int n = 0; n < NT; ++n)
            acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8[n], b3[q], acc[n], 0, 0, 0);
      }
    }
  };

  if constexpr (KS) {
    constexpr int SPP = KF * (CC / 16);          // K steps per phase
    constexpr int S = NPH * SPP;
    static_assert(S % 4 == 0, "K steps must deal evenly to four waves");
    f32x4 wreg[(S / 4) * NT];
#pragma unroll
    for (int j = 0; j < S / 4; ++j)
#pragma unroll
      for (int n = 0; n < NT; ++n) wreg[j * NT + n] = wp[((4 * j + wave) * NT + n) * 64];
    const int phase_b = (nseg * RS * PITCH_B + 255) & ~255;
#pragma unroll
    for (int ph = 0; ph < NPH; ++ph) stage(ph / NCH, ph % NCH, ph * phase_b);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < S / 4; ++j) {
      const int st = 4 * j + wave;               // wave-uniform
      const int ph = st / SPP, r = st - ph * SPP, kf = r / (CC / 16), g = r - kf * (CC / 16);
      const int koff = ph * phase_b + ((STRIDE == 1) ? (kf * PITCH_B) : ((kf >> 1) * PITCH_B + (kf & 1) * (CC * 2)));
      bf16x8 b3[3];
#pragma unroll
      for (int q = 0; q < 3; ++q)
        b3[q] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4*>(ldsb + lbase + koff + 32 * g + q * PLANE_B));
#pragma unroll
      for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int n = 0; n < NT; ++n)
          acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wreg[j * NT + n]), b3[q], acc[n], 0, 0, 0);
    }
    __syncthreads();                             // every wave has read its B rows: the image becomes the exchange buffer
    if (wave != 0) {
#pragma unroll
      for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          *reinterpret_cast<f32x4*>(ldsb + ((((wave - 1) * NT + n) * 4 + q) * 64 + lane) * 16) = f32x4{acc[n][4 * q], acc[n][4 * q + 1], acc[n][4 * q + 2], acc[n][4 * q + 3]};
    }
    __syncthreads();
    if (wave != 0) return;
#pragma unroll
    for (int w = 0; w < 3; ++w)
#pragma unroll
      for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(ldsb + (((w * NT + n) * 4 + q) * 64 + lane) * 16);
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[n][4 * q + i] += v[i];
        }
  } else if constexpr (ALL) {
    // Small layers (all phases fit LDS side by side, all weights fit the registers): everything is requested up front and there
    // is ONE barrier -- these launches are nothing but latency (20-25 us for a few microseconds of work with a barrier pair and
    // an L2 round trip per phase).
    f32x4 wreg[NPH * WPH];
#pragma unroll
    for (int i = 0; i < NPH * WPH; ++i) wreg[i] = wp[i * 64];
    const int phase_b = (nseg * RS * PITCH_B + 255) & ~255;
#pragma unroll
    for (int ph = 0; ph < NPH; ++ph) stage(ph / NCH, ph % NCH, ph * phase_b);
    __syncthreads();
#pragma unroll
    for (int ph = 0; ph < NPH; ++ph) mfma_phase(ph * phase_b, wreg + ph * WPH);
  } else {
#pragma unroll 1
    for (int t = 0; t < TT; ++t) {
#pragma unroll 1
      for (int ch = 0; ch < NCH; ++ch) {
        // the phase's weights are requested BEFORE its input rows are staged: they do not depend on the input, and behind the
        // barriers below every K step would wait for its own L2 round trip
        f32x4 wreg[WPH];
#pragma unroll
        for (int i = 0; i < WPH; ++i) wreg[i] = wp[i * 64];
        wp += WPH * 64;
        if (t + ch > 0) __syncthreads();   // all waves finished reading the previous phase
        stage(t, ch, 0);
        __syncthreads();
        mfma_phase(0, wreg);
      }
    }
  }

  // ---------------- epilogue: weight scale, bias (+ LayerNorm over the group's channels + PReLU), store ------
  // accumulator register r of lane (pos, h) holds channel  tile*32 + 8*(r>>2) + 4*h + (r&3)
  const int P = P0 + ploc;
  const bool valid = P < total_pos;
  const int b = P >> log2f, f = P & (F_out - 1);
  const size_t soff = slot_of(b, p.sm) * p.sstride;
  const size_t row0 = static_cast<size_t>(f) * p.row_mul + p.row_add;
#pragma unroll
  for (int gi = 0; gi < R; ++gi) {
    float v[G][16];
#pragma unroll
    for (int tg = 0; tg < G; ++tg) {
      const int n = gi * G + tg;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 bb = *reinterpret_cast<const f32x4*>(p.bias + n * 32 + 8 * q + 4 * h);
        const f32x4 sc = *reinterpret_cast<const f32x4*>(p.wscale + n * 32 + 8 * q + 4 * h);
#pragma unroll
        for (int i = 0; i < 4; ++i) v[tg][4 * q + i] = acc[n][4 * q + i] * sc[i] + bb[i];
      }
    }
    if (EPI_LN) {
      constexpr float inv_n = 1.0f / (32 * G);
      float s = 0.f;
#pragma unroll
      for (int tg = 0; tg < G; ++tg)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += v[tg][r];
      s += __shfl_xor(s, 32);
      const float mean = s * inv_n;
      float qv = 0.f;
#pragma unroll
      for (int tg = 0; tg < G; ++tg)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          v[tg][r] -= mean;
          qv += v[tg][r] * v[tg][r];
        }
      qv += __shfl_xor(qv, 32);
      const float rstd = 1.0f / sqrtf(qv * inv_n + LN_EPS);
#pragma unroll
      for (int tg = 0; tg < G; ++tg)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 gm = *reinterpret_cast<const f32x4*>(p.gamma + tg * 32 + 8 * q + 4 * h);
          const f32x4 bt = *reinterpret_cast<const f32x4*>(p.beta + tg * 32 + 8 * q + 4 * h);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float y = v[tg][4 * q + i] * rstd * gm[i] + bt[i];
            v[tg][4 * q + i] = y >= 0.f ? y : p.alpha * y;
          }
        }
    }
    if (valid) {
      const size_t row = row0 + gi;
#pragma unroll
      for (int tg = 0; tg < G; ++tg)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 o = {v[tg][4 * q], v[tg][4 * q + 1], v[tg][4 * q + 2], v[tg][4 * q + 3]};
          const int c = tg * 32 + 8 * q + 4 * h;
          *reinterpret_cast<f32x4*>(p.dst0 + soff + row * p.ld0 + c) = o;
          if (p.dst1) *reinterpret_cast<f32x4*>(p.dst1 + soff + row * p.ld1 + c) = o;
        }
    }
  }
}

// ------------------------------------------------------------------------------------------------
ConvShape conv_shape(ConvKind k) {
  switch (k) {
    //                         cin  nt  s  tt kf padl ln g
    case CONV_EL_C32:   return {32,  1, 2, 2, 3, 1, 1, 1};
    case CONV_EL_C64:   return {64,  1, 2, 2, 3, 1, 1, 1};
    case CONV_EL_C128:  return {128, 1, 2, 2, 3, 1, 1, 1};
    case CONV_DL_N64:   return {64,  2, 1, 2, 3, 1, 1, 1};
    case CONV_DL_N128:  return {64,  4, 1, 2, 3, 1, 1, 2};
    case CONV_IN_C64:   return {64,  2, 1, 1, 1, 0, 1, 2};
    case CONV_IN_C128:  return {128, 2, 1, 1, 1, 0, 1, 2};
    case CONV_DOWN:     return {64,  2, 2, 1, 3, 0, 0, 2};
    case CONV_UP_EVEN:  return {128, 4, 1, 1, 2, 1, 0, 4};
    case CONV_UP_ODD:   return {128, 4, 1, 1, 1, 0, 0, 4};
    default:            return {0, 0, 0, 0, 0, 0, 0, 0};
  }
}

size_t conv_lds_bytes(ConvKind k, int f_out, int nw) {
  const ConvShape s = conv_shape(k);
  const int cc = s.cin < 64 ? s.cin : 64;
  const int tp = 32 * nw;
  const int seg_len = f_out < tp ? f_out : tp;
  const int nseg = tp / seg_len;
  const int pitch = s.stride == 1 ? cc + 4 : 2 * cc + 4;
  const int rs = s.stride == 1 ? seg_len + s.kf - 1 : seg_len + (s.kf - 1) / 2;
  return static_cast<size_t>(nseg) * rs * pitch * sizeof(float);
}

static size_t conv_lds_bytes_bf16(ConvKind k, int f_out, int nw) {
  const ConvShape s = conv_shape(k);
  const int cc = s.cin < 64 ? s.cin : 64;
  const int tp = 32 * nw;
  const int seg_len = f_out < tp ? f_out : tp;
  const int nseg = tp / seg_len;
  const int pitch_b = 3 * (s.stride == 1 ? cc : 2 * cc) * 2 + 16;
  const int rs = s.stride == 1 ? seg_len + s.kf - 1 : seg_len + (s.kf - 1) / 2;
  return static_cast<size_t>(nseg) * rs * pitch_b;
}

int conv_pick_nw(ConvKind, int B, int f_out) {
  // 4-wave workgroups (128 positions) once there are enough positions to give every CU several
  // workgroups; single-wave workgroups otherwise so small layers spread over more CUs.
  return (static_cast<long long>(B) * f_out >= 128LL * 512) ? 4 : 1;
}

constexpr int kMaxDevices = 64;

template <auto Kern>
static hipError_t launch_conv_variant(size_t lds, unsigned grid, unsigned threads, const ConvParams& p, hipStream_t s) {
  // raise the dynamic-LDS cap once per KERNEL and device (function attributes are per kernel and per device).  The kernel is a
  // non-type template parameter: every instantiation of conv_bf16x3_kernel / conv_mfma_kernel gets its own table -- with the
  // kernel passed as a function argument all of them shared one (they have the same pointer type), and an instantiation that
  // needed less than the largest cap raised so far never got its own hipFuncSetAttribute.
  static std::atomic<size_t> lds_cap[kMaxDevices] = {};
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::atomic<size_t>& cap = lds_cap[dev >= 0 && dev < kMaxDevices ? dev : 0];
  if (lds > 64 * 1024 && (lds > cap.load(std::memory_order_relaxed) || dev >= kMaxDevices)) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(Kern), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
    if (e != hipSuccess) return e;
    cap.store(lds, std::memory_order_relaxed);
  }
  hipLaunchKernelGGL(Kern, dim3(grid), dim3(threads), lds, s, p);
  return hipGetLastError();
}

template <int CIN, int NT, int STRIDE, int TT, int KF, int PADL, int EPI_LN, int G>
static hipError_t launch_conv_t(ConvKind k, const ConvParams& p, hipStream_t s) {
  if (p.use_bf16 && p.wbf && p.wscale) {      // block mode on an int8 container: the bf16-pipe kernel
    // 128-position tiles from 32 k positions per launch on, below that the K-split kernel on 32-position tiles (measured at
    // 16 k .. 1 M positions: 293 k frames/s at 16-32 k, 288 k at 64 k and above; one 1024-frame utterance, three chunks)
    const long long totalb = static_cast<long long>(p.B) * p.F_out;
    const int nw = totalb >= 64LL * 512 ? 4 : 1;
    const size_t ldsb = conv_lds_bytes_bf16(k, p.F_out, nw);
    if (nw == 4)
      return launch_conv_variant<conv_bf16x3_kernel<CIN, NT, STRIDE, TT, KF, PADL, EPI_LN, G, 4, false>>(ldsb, static_cast<unsigned>((totalb + 127) / 128), 256, p, s);
    // 1-wave workgroups (the layers with few positions): all phases at once where LDS and the registers allow it
    constexpr int cc = CIN < 64 ? CIN : 64, nph = TT * (CIN / cc), wph = KF * (cc / 16) * NT;
    const size_t phase_b = (ldsb + 255) & ~static_cast<size_t>(255);
    // four waves on the K steps of one 32-position tile where the K steps deal evenly and a wave's quarter of the weights fits
    // its registers; the exchange buffer (3 waves x NT x 4 KB) re-uses the image
    static const bool ksplit = [] { const char* v = getenv("NUTLS_OFFLINE_KSPLIT"); return !v || atoi(v) != 0; }();
    if constexpr ((nph * KF * (cc / 16)) % 4 == 0 && nph * wph <= 192) {
      const size_t need = std::max<size_t>(nph * phase_b, 3 * NT * 4096);
      if (ksplit && need <= 128 * 1024)
        return launch_conv_variant<conv_bf16x3_kernel<CIN, NT, STRIDE, TT, KF, PADL, EPI_LN, G, 4, true>>(need, static_cast<unsigned>((totalb + 31) / 32), 256, p, s);
    }
    if constexpr (nph > 1 && nph * wph <= 48) {
      if (nph * phase_b <= 64 * 1024)
        return launch_conv_variant<conv_bf16x3_kernel<CIN, NT, STRIDE, TT, KF, PADL, EPI_LN, G, 1, true>>(nph * phase_b, static_cast<unsigned>((totalb + 31) / 32), 64, p, s);
    }
    return launch_conv_variant<conv_bf16x3_kernel<CIN, NT, STRIDE, TT, KF, PADL, EPI_LN, G, 1, false>>(ldsb, static_cast<unsigned>((totalb + 31) / 32), 64, p, s);
  }
  const int nw = conv_pick_nw(k, p.B, p.F_out);
  const size_t lds = conv_lds_bytes(k, p.F_out, nw);
  const long long total = static_cast<long long>(p.B) * p.F_out;
  if (nw == 4)
    return launch_conv_variant<conv_mfma_kernel<CIN, NT, STRIDE, TT, KF, PADL, EPI_LN, G, 4>>(lds, static_cast<unsigned>((total + 127) / 128), 256, p, s);
  return launch_conv_variant<conv_mfma_kernel<CIN, NT, STRIDE, TT, KF, PADL, EPI_LN, G, 1>>(lds, static_cast<unsigned>((total + 31) / 32), 64, p, s);
}

hipError_t launch_conv(ConvKind k, const ConvParams& p, hipStream_t s) {
  switch (k) {
    case CONV_EL_C32:   return launch_conv_t<32,  1, 2, 2, 3, 1, 1, 1>(k, p, s);
    case CONV_EL_C64:   return launch_conv_t<64,  1, 2, 2, 3, 1, 1, 1>(k, p, s);
    case CONV_EL_C128:  return launch_conv_t<128, 1, 2, 2, 3, 1, 1, 1>(k, p, s);
    case CONV_DL_N64:   return launch_conv_t<64,  2, 1, 2, 3, 1, 1, 1>(k, p, s);
    case CONV_DL_N128:  return launch_conv_t<64,  4, 1, 2, 3, 1, 1, 2>(k, p, s);
    case CONV_IN_C64:   return launch_conv_t<64,  2, 1, 1, 1, 0, 1, 2>(k, p, s);
    case CONV_IN_C128:  return launch_conv_t<128, 2, 1, 1, 1, 0, 1, 2>(k, p, s);
    case CONV_DOWN:     return launch_conv_t<64,  2, 2, 1, 3, 0, 0, 2>(k, p, s);
    case CONV_UP_EVEN:  return launch_conv_t<128, 4, 1, 1, 2, 1, 0, 4>(k, p, s);
    case CONV_UP_ODD:   return launch_conv_t<128, 4, 1, 1, 1, 0, 0, 4>(k, p, s);
    default:            return hipErrorInvalidValue;
  }
}

// ================================================================================================
//  LSTM cell (21 units, gates i,f,g,o) + Dense  -- models/proposed.py:70-119, used at
//  converter_proposed.py:234-237.  One workgroup per stream; the flattened [F_D, 32] bottleneck
//  (F major, C minor) is gathered into LDS, thread n < 84 owns gate pre-activation n, weights are
//  stored transposed ([k][n]) so the 84 threads read consecutive floats.
// ================================================================================================
#define LSTM_UNITS 21
#define LSTM_GATES 84

__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + expf(-x)); }

__global__ __launch_bounds__(128) void lstm_dense_kernel(const LstmParams p) {
  __shared__ float v[256];
  __shared__ float hs[32];
  __shared__ float z[96];
  __shared__ float hn[32];
  const int b = blockIdx.x, tid = threadIdx.x;
  for (int k = tid; k < p.Din; k += 128) {
    const int f = k / p.x_cols, c = k - f * p.x_cols;
    v[k] = p.x[slot_of(b, p.sm) * p.sstride + static_cast<size_t>(f) * p.x_ld + c];
  }
  if (tid < LSTM_UNITS) hs[tid] = p.h_in[slot_of(b, p.sm) * p.sstride + tid];
  __syncthreads();
  if (tid < LSTM_GATES) {
    float a = p.bias[tid];
    for (int k = 0; k < p.Din; ++k) a = fmaf(p.wxT[k * LSTM_GATES + tid], v[k], a);
    float r = 0.f;
    for (int u = 0; u < LSTM_UNITS; ++u) r = fmaf(p.whT[u * LSTM_GATES + tid], hs[u], r);
    z[tid] = a + r;
  }
  __syncthreads();
  if (tid < LSTM_UNITS) {
    const float gi = sigmoid_f(z[tid]);
    const float gf = sigmoid_f(z[LSTM_UNITS + tid]);
    const float gg = tanhf(z[2 * LSTM_UNITS + tid]);
    const float go = sigmoid_f(z[3 * LSTM_UNITS + tid]);
    const float c_old = p.c_in[slot_of(b, p.sm) * p.sstride + tid];
    const float c_new = gf * c_old + gi * gg;
    const float h_new = go * tanhf(c_new);
    p.c_out[slot_of(b, p.sm) * p.sstride + tid] = c_new;
    p.h_out[slot_of(b, p.sm) * p.sstride + tid] = h_new;
    hn[tid] = h_new;
  }
  __syncthreads();
  for (int m = tid; m < p.Dout; m += 128) {
    float a = p.bd[m];
#pragma unroll
    for (int u = 0; u < LSTM_UNITS; ++u) a = fmaf(p.wdT[u * p.Dout + m], hn[u], a);
    const int f = m / p.dst_cols, c = m - f * p.dst_cols;
    p.dst[slot_of(b, p.sm) * p.sstride + static_cast<size_t>(f) * p.dst_ld + c] = a;
  }
}

hipError_t launch_lstm(const LstmParams& p, hipStream_t s) {
  if (p.Din > 256) return hipErrorInvalidValue;
  hipLaunchKernelGGL(lstm_dense_kernel, dim3(p.B), dim3(128), 0, s, p);
  return hipGetLastError();
}

// ================================================================================================
//  CTFA gate + residual -- ctfa_rt (models/proposed.py:162-196) as wired at
//  converter_proposed.py:258-262:   y = x * (TA * FA) + e0
//      TA = sigmoid(W2 relu(W1 mean_f(x) + b1) + b2)
//      FA = sigmoid(V2 relu(V1 (TA/32) + c1) + c2)      (T = 1: the 32-frame average pool sees
//                                                          31 zero frames + TA)
//  One workgroup per stream: 16 row-groups x 16 float4 channel groups reduce the mean over F in
//  LDS, 64 threads run the two 64->16->64 MLPs, then all threads apply the gate (x re-read from L2).
// ================================================================================================
__global__ __launch_bounds__(256) void ctfa_kernel(const CtfaParams p) {
  __shared__ __attribute__((aligned(16))) float part[16][64];
  __shared__ float m[64];
  __shared__ float hid[16];
  __shared__ float ta[64];
  __shared__ __attribute__((aligned(16))) float gate[64];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int c4 = tid & 15, rg = tid >> 4;
  const float* xb = p.x + slot_of(b, p.sm) * p.sstride;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  for (int f = rg; f < p.F; f += 16) s += *reinterpret_cast<const f32x4*>(xb + static_cast<size_t>(f) * p.x_ld + 4 * c4);
  *reinterpret_cast<f32x4*>(&part[rg][4 * c4]) = s;
  __syncthreads();
  if (tid < 64) {
    float a = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) a += part[r][tid];
    m[tid] = a / static_cast<float>(p.F);
  }
  __syncthreads();
  if (tid < 16) {
    float a = p.ta_b1[tid];
    for (int c = 0; c < 64; ++c) a = fmaf(p.ta_w1T[c * 16 + tid], m[c], a);
    hid[tid] = fmaxf(a, 0.f);
  }
  __syncthreads();
  if (tid < 64) {
    float a = p.ta_b2[tid];
#pragma unroll
    for (int u = 0; u < 16; ++u) a = fmaf(p.ta_w2T[u * 64 + tid], hid[u], a);
    ta[tid] = sigmoid_f(a);
  }
  __syncthreads();
  if (tid < 16) {
    float a = p.fa_b1[tid];
    for (int c = 0; c < 64; ++c) a = fmaf(p.fa_w1T[c * 16 + tid], ta[c] * (1.0f / 32.0f), a);
    hid[tid] = fmaxf(a, 0.f);
  }
  __syncthreads();
  if (tid < 64) {
    float a = p.fa_b2[tid];
#pragma unroll
    for (int u = 0; u < 16; ++u) a = fmaf(p.fa_w2T[u * 64 + tid], hid[u], a);
    gate[tid] = sigmoid_f(a) * ta[tid];
  }
  __syncthreads();
  const f32x4 g4 = *reinterpret_cast<const f32x4*>(&gate[4 * c4]);
  const float* eb = p.e0 + slot_of(b, p.sm) * p.sstride;
  float* yb = p.y + slot_of(b, p.sm) * p.sstride;
  for (int f = rg; f < p.F; f += 16) {
    const f32x4 xv = *reinterpret_cast<const f32x4*>(xb + static_cast<size_t>(f) * p.x_ld + 4 * c4);
    const f32x4 ev = *reinterpret_cast<const f32x4*>(eb + static_cast<size_t>(f) * p.e0_ld + 4 * c4);
    *reinterpret_cast<f32x4*>(yb + static_cast<size_t>(f) * p.y_ld + 4 * c4) = xv * g4 + ev;
  }
}

// ---- CTFA with the TRUE 32-frame causal average of the offline model (models/proposed.py:125-160 `ctfa`:
// ZeroPadding2D((31,0)) + AveragePooling1D(32, strides=1) over the time-attention vectors of real frames), offline / block
// mode only.  hist [31 + frames][64]: rows 0..30 = TA of the 31 frames before this block (zeros at the start of an
// utterance), row 31 + t = TA of block frame t.  Pass 1 (parallel over frames): TA.  Pass 2: FA from the mean of the last
// 32 TA rows, gate, residual.  Pass 3: the last 31 rows move to the front for the next block.
// (several utterances per handle: frame b = u * n + t of the launch -- p.sm -- keeps its time attention in utterance u's history, hist_ustride floats further)
__device__ __forceinline__ size_t hist_row(int b, const SlotMap& m, long long ustride) {
  if (!m.n) return static_cast<size_t>(31 + b) * 64;
  const unsigned u = utt_of(b, m);
  return static_cast<size_t>(u) * ustride + static_cast<size_t>(31 + b - static_cast<int>(u) * m.n) * 64;
}
__global__ __launch_bounds__(256) void ctfa_ta_kernel(const CtfaParams p, float* __restrict__ hist, long long hist_ustride) {
  __shared__ __attribute__((aligned(16))) float part[16][64];
  __shared__ float m[64];
  __shared__ float hid[16];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int c4 = tid & 15, rg = tid >> 4;
  const float* xb = p.x + slot_of(b, p.sm) * p.sstride;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  for (int f = rg; f < p.F; f += 16) s += *reinterpret_cast<const f32x4*>(xb + static_cast<size_t>(f) * p.x_ld + 4 * c4);
  *reinterpret_cast<f32x4*>(&part[rg][4 * c4]) = s;
  __syncthreads();
  if (tid < 64) {
    float a = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) a += part[r][tid];
    m[tid] = a / static_cast<float>(p.F);
  }
  __syncthreads();
  if (tid < 16) {
    float a = p.ta_b1[tid];
    for (int c = 0; c < 64; ++c) a = fmaf(p.ta_w1T[c * 16 + tid], m[c], a);
    hid[tid] = fmaxf(a, 0.f);
  }
  __syncthreads();
  if (tid < 64) {
    float a = p.ta_b2[tid];
#pragma unroll
    for (int u = 0; u < 16; ++u) a = fmaf(p.ta_w2T[u * 64 + tid], hid[u], a);
    hist[hist_row(b, p.sm, hist_ustride) + tid] = sigmoid_f(a);
  }
}

__global__ __launch_bounds__(256) void ctfa_apply_causal_kernel(const CtfaParams p, const float* __restrict__ hist, long long hist_ustride) {
  __shared__ float avg[64];
  __shared__ float hid[16];
  __shared__ __attribute__((aligned(16))) float gate[64];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int c4 = tid & 15, rg = tid >> 4;
  float ta = 0.f;
  if (tid < 64) {
    float a = 0.f;
    const size_t hr = hist_row(b, p.sm, hist_ustride);
    for (int k = 0; k < 32; ++k) a += hist[hr - static_cast<size_t>(k) * 64 + tid];      // oldest rows are zeros before the utterance starts
    avg[tid] = a * (1.0f / 32.0f);
    ta = hist[hr + tid];
  }
  __syncthreads();
  if (tid < 16) {
    float a = p.fa_b1[tid];
    for (int c = 0; c < 64; ++c) a = fmaf(p.fa_w1T[c * 16 + tid], avg[c], a);
    hid[tid] = fmaxf(a, 0.f);
  }
  __syncthreads();
  if (tid < 64) {
    float a = p.fa_b2[tid];
#pragma unroll
    for (int u = 0; u < 16; ++u) a = fmaf(p.fa_w2T[u * 64 + tid], hid[u], a);
    gate[tid] = sigmoid_f(a) * ta;
  }
  __syncthreads();
  const f32x4 g4 = *reinterpret_cast<const f32x4*>(&gate[4 * c4]);
  const float* xb = p.x + slot_of(b, p.sm) * p.sstride;
  const float* eb = p.e0 + slot_of(b, p.sm) * p.sstride;
  float* yb = p.y + slot_of(b, p.sm) * p.sstride;
  for (int f = rg; f < p.F; f += 16) {
    const f32x4 xv = *reinterpret_cast<const f32x4*>(xb + static_cast<size_t>(f) * p.x_ld + 4 * c4);
    const f32x4 ev = *reinterpret_cast<const f32x4*>(eb + static_cast<size_t>(f) * p.e0_ld + 4 * c4);
    *reinterpret_cast<f32x4*>(yb + static_cast<size_t>(f) * p.y_ld + 4 * c4) = xv * g4 + ev;
  }
}

__global__ __launch_bounds__(1024) void ctfa_hist_roll_kernel(float* __restrict__ hist, int frames, long long hist_ustride) {
  hist += static_cast<size_t>(blockIdx.x) * hist_ustride;      // (one workgroup per utterance)
  const int tid = threadIdx.x;          // 31 x 64 = 1984 values, two per thread: read everything, then write (the ranges overlap when frames < 31)
  float v[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int e = tid + 1024 * i;
    v[i] = e < 31 * 64 ? hist[static_cast<size_t>(frames) * 64 + e] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int e = tid + 1024 * i;
    if (e < 31 * 64) hist[e] = v[i];
  }
}

hipError_t launch_ctfa_causal(const CtfaParams& p, float* hist, bool roll, hipStream_t s, int utts, long long hist_ustride) {
  hipLaunchKernelGGL(ctfa_ta_kernel, dim3(p.B), dim3(256), 0, s, p, hist, hist_ustride);
  hipLaunchKernelGGL(ctfa_apply_causal_kernel, dim3(p.B), dim3(256), 0, s, p, hist, hist_ustride);
  if (roll) hipLaunchKernelGGL(ctfa_hist_roll_kernel, dim3(utts), dim3(1024), 0, s, hist, p.B / utts, hist_ustride);
  return hipGetLastError();
}

// the last 31 rows of a block of `frames` frames become rows 0..30 of the next block's history
hipError_t launch_ctfa_hist_roll(float* hist, int frames, hipStream_t s, int utts, long long hist_ustride) {
  hipLaunchKernelGGL(ctfa_hist_roll_kernel, dim3(utts), dim3(1024), 0, s, hist, frames, hist_ustride);
  return hipGetLastError();
}

hipError_t launch_ctfa(const CtfaParams& p, hipStream_t s) {
  hipLaunchKernelGGL(ctfa_kernel, dim3(p.B), dim3(256), 0, s, p);
  return hipGetLastError();
}

// ================================================================================================
//  input_layer: 1x1 conv 1 -> 64 + LN + PReLU (models/proposed.py:218-225 with Cin = 1).
//  16 lanes per position, 4 channels per lane; LayerNorm reduces across the 16 lanes.
// ================================================================================================
__global__ __launch_bounds__(256) void input_layer_kernel(const InLayerParams p) {
  const int gid = blockIdx.x * 256 + threadIdx.x;
  const int pos = gid >> 4, c4 = gid & 15;
  const bool valid = pos < p.n_pos;
  const float x = valid ? p.x[slot_of(pos >> 8, p.sm_io) * 256 + (pos & 255)] : 0.f;
  const f32x4 w = *reinterpret_cast<const f32x4*>(p.w + 4 * c4);
  const f32x4 bb = *reinterpret_cast<const f32x4*>(p.b + 4 * c4);
  f32x4 y = w * x + bb;
  float s = y[0] + y[1] + y[2] + y[3];
#pragma unroll
  for (int o = 1; o < 16; o <<= 1) s += __shfl_xor(s, o);
  const float mean = s * (1.0f / 64.0f);
  y -= mean;
  float q = y[0] * y[0] + y[1] * y[1] + y[2] * y[2] + y[3] * y[3];
#pragma unroll
  for (int o = 1; o < 16; o <<= 1) q += __shfl_xor(q, o);
  const float rstd = 1.0f / sqrtf(q * (1.0f / 64.0f) + LN_EPS);
  const f32x4 gm = *reinterpret_cast<const f32x4*>(p.gamma + 4 * c4);
  const f32x4 bt = *reinterpret_cast<const f32x4*>(p.beta + 4 * c4);
  f32x4 o;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float t = y[i] * rstd * gm[i] + bt[i];
    o[i] = t >= 0.f ? t : p.alpha * t;
  }
  if (valid) *reinterpret_cast<f32x4*>(p.y + slot_of(pos >> 8, p.sm) * p.sstride + static_cast<size_t>(pos & 255) * 64 + 4 * c4) = o;
}

hipError_t launch_input_layer(const InLayerParams& p, hipStream_t s) {
  const unsigned grid = static_cast<unsigned>((static_cast<long long>(p.n_pos) * 16 + 255) / 256);
  hipLaunchKernelGGL(input_layer_kernel, dim3(grid), dim3(256), 0, s, p);
  return hipGetLastError();
}

// ================================================================================================
//  output conv: 1x1, 64 -> 1, linear (models/proposed.py:65, :1146)
// ================================================================================================
__global__ __launch_bounds__(256) void out_conv_kernel(const OutConvParams p) {
  const int gid = blockIdx.x * 256 + threadIdx.x;
  const int pos = gid >> 4, c4 = gid & 15;
  const bool valid = pos < p.n_pos;
  float s = 0.f;
  if (valid) {
    const f32x4 xv = *reinterpret_cast<const f32x4*>(p.x + slot_of(pos >> 8, p.sm) * p.sstride + static_cast<size_t>(pos & 255) * p.x_ld + 4 * c4);
    const f32x4 w = *reinterpret_cast<const f32x4*>(p.w + 4 * c4);
    s = xv[0] * w[0] + xv[1] * w[1] + xv[2] * w[2] + xv[3] * w[3];
  }
#pragma unroll
  for (int o = 1; o < 16; o <<= 1) s += __shfl_xor(s, o);
  if (valid && c4 == 0) p.y[slot_of(pos >> 8, p.sm_io) * 256 + (pos & 255)] = s + p.bias;
}

hipError_t launch_out_conv(const OutConvParams& p, hipStream_t s) {
  const unsigned grid = static_cast<unsigned>((static_cast<long long>(p.n_pos) * 16 + 255) / 256);
  hipLaunchKernelGGL(out_conv_kernel, dim3(grid), dim3(256), 0, s, p);
  return hipGetLastError();
}

// ================================================================================================
//  Baseline variant: dilated-dense bottleneck, one workgroup per stream (ddb_device.hpp)
// ================================================================================================
__global__ __launch_bounds__(256) void ddb_kernel(const DdbParams p) {
  __shared__ float lds[2 * 4 * 64 + 8 * 4 * 32];
  ddb_block(p, blockIdx.x, lds, threadIdx.x, 256);
}

hipError_t launch_ddb(const DdbParams& p, hipStream_t s) {
  if (p.F * p.C > 256) return hipErrorInvalidValue;
  hipLaunchKernelGGL(ddb_kernel, dim3(p.B), dim3(256), 0, s, p);
  return hipGetLastError();
}

__global__ void incr_step_kernel(int* step) { *step += 1; }

hipError_t launch_incr_step(int* step, hipStream_t s) {
  hipLaunchKernelGGL(incr_step_kernel, dim3(1), dim3(1), 0, s, step);
  return hipGetLastError();
}

// Carried partial sums of the fused kernel's two-tap convs, rebuilt from the conv-input state tensors (rare: after nutls_state_set or
// a step of another mode): S[pos][n] = sum over frequency taps k and channels c of W[n][tap 0][k][c] * x[stride pos - 1 + k][c],
// rows outside the tensor are the zero padding (models/proposed.py:208-216, :240-251).  grid = (ops, streams).
__global__ __launch_bounds__(256) void ysum_refresh_kernel(const float* arena, long long sstride, int x_block_off, int ys_block_off,
                                                           const YsOp* ops, const float* w) {
  const YsOp o = ops[blockIdx.x];
  const float* slice = arena + static_cast<size_t>(blockIdx.y) * sstride;
  const float* x = slice + x_block_off + o.xs_off;
  float* y = const_cast<float*>(slice) + ys_block_off + o.ys_off;
  const float* wq = w + o.w_off;
  const int rows = o.stride * o.P;
  for (int idx = threadIdx.x; idx < o.P * o.N; idx += 256) {
    const int pos = idx / o.N, n = idx - pos * o.N;
    float a = 0.f;
    for (int k = 0; k < 3; ++k) {
      const int row = o.stride * pos - 1 + k;
      if (row < 0 || row >= rows) continue;
      const float* xr = x + static_cast<size_t>(row) * o.xs_ld;
      const float* wr = wq + (static_cast<size_t>(n) * 3 + k) * o.cin;
      for (int c = 0; c < o.cin; ++c) a = fmaf(wr[c], xr[c], a);
    }
    int dst = idx;
    if (o.r32) {      // accumulator order of the 32x32 tiles: lane (pos & 31, half h) register 4 q + e holds channel 32 T + 8 q + 4 h + e
      const int tp = pos >> 5, j = pos & 31, T = n >> 5, c32 = n & 31;
      const int task = tp / o.PT + o.PG * (T / o.NT);
      dst = ((((task * o.PT + tp % o.PT) * o.NT + T % o.NT) * 4 + (c32 >> 3)) * 64 + ((c32 >> 2) & 1) * 32 + j) * 4 + (c32 & 3);
    }
    y[dst] = a;
  }
}

hipError_t launch_ysum_refresh(const float* arena, long long sstride, int x_block_off, int ys_block_off, const YsOp* ops, const float* w,
                               int n_ops, int B, hipStream_t s) {
  hipLaunchKernelGGL(ysum_refresh_kernel, dim3(n_ops, B), dim3(256), 0, s, arena, sstride, x_block_off, ys_block_off, ops, w);
  return hipGetLastError();
}

// Causal32 mode of the streaming CTFA (the offline model's `ctfa`, models/proposed.py:143-147: the frequency branch sees the mean of the
// time attention over the last 32 frames): the sum over the 31 frames BEFORE the one about to be stepped, from the history ring
// [B][12 stages][32 frames][64] -- every row but `slot`, the one this frame's time attention will replace.  grid = B * 12, 64 threads.
__global__ __launch_bounds__(64) void ta_sum_kernel(const float* ring, float* sum, int slot) {
  const float* r = ring + static_cast<size_t>(blockIdx.x) * (32 * 64) + threadIdx.x;
  float a = 0.f;
#pragma unroll
  for (int k = 0; k < 32; ++k) a += k == slot ? 0.f : r[k * 64];
  sum[static_cast<size_t>(blockIdx.x) * 64 + threadIdx.x] = a;
}

hipError_t launch_ta_sum(const float* ring, float* sum, int slot, int B, hipStream_t s) {
  hipLaunchKernelGGL(ta_sum_kernel, dim3(B * 12), dim3(64), 0, s, ring, sum, slot);
  return hipGetLastError();
}

// Lazily written state tensors of the fused kernel (fused_plan.hpp OpD::d0_on = 2, engine.cpp states_materialize): the input state of a
// strided conv holds the same rows as the skip-connection slice of the stage's sub-pixel conv input, which the kernel does write.
// grid = (entries, B), 256 threads: float4 copies of rows x width floats.
__global__ __launch_bounds__(256) void lazy_states_kernel(float* arena, long long sstride, int block_off, const LazyCopy* tab) {
  const LazyCopy c = tab[blockIdx.x];
  float* base = arena + static_cast<size_t>(blockIdx.y) * sstride + block_off;
  const int w4 = c.width / 4, n = c.rows * w4;
  for (int i = threadIdx.x; i < n; i += 256) {
    const int r = i / w4, q = i % w4;
    *reinterpret_cast<float4*>(base + c.dst_off + r * c.dst_ld + 4 * q) = *reinterpret_cast<const float4*>(base + c.src_off + r * c.src_ld + 4 * q);
  }
}

hipError_t launch_lazy_states(float* arena, long long sstride, int block_off, const LazyCopy* tab, int n, int B, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(lazy_states_kernel, dim3(n, B), dim3(256), 0, s, arena, sstride, block_off, tab);
  return hipGetLastError();
}

__global__ void set_step_kernel(int* step, int value) { *step = value; }

hipError_t launch_set_step(int* step, int value, hipStream_t s) {
  hipLaunchKernelGGL(set_step_kernel, dim3(1), dim3(1), 0, s, step, value);
  return hipGetLastError();
}

}  // namespace nutls
