// Persistent "one workgroup per stream" kernel: the whole NUNet-TLS-LSTM frame step of one stream
// runs inside ONE 1024-thread workgroup (16 waves = 4 per SIMD, one workgroup per CU), layer after
// layer, driven by the device-resident launch plan.  Streams are independent,
// so no inter-workgroup synchronisation exists: a layer boundary is a __syncthreads(), not a
// kernel boundary.  At B = 256 streams this is exactly one stream per CU of the MI355X.
//
// Per conv-like layer (reference blocks: models/proposed.py:198-265) the workgroup
//   1. stages the input rows of the current (time tap, 64-channel chunk) in LDS; the global loads
//      of the NEXT phase -- and, across layers, of the next layer's previous-frame tap, which
//      never depends on the current frame -- are issued before the MFMA loop and land in registers
//      while the matrix cores work (register-staged prefetch);
//   2. splits the GEMM  D[ch,pos] = W[ch,k] X[k,pos]  into (position tile, channel tile, K slice)
//      tasks of 32x32 outputs, one task per wave, so even a layer with 4 output positions keeps
//      8..16 waves busy (split-K); v_mfma_f32_32x32x2_f32, exact fp32;
//   3. drops the partial tiles into an LDS exchange buffer [k-slice][position][channel];
//   4. re-reads it row-wise (8/16/32 lanes per output row, float4 per lane): sum of K slices + bias,
//      LayerNorm over the row's channels with DPP shuffles, PReLU, and writes full 128-byte
//      channels-last rows to the (up to two) destination state tensors.
#include <hip/hip_runtime.h>

#include "nutls_internal.hpp"

namespace nutls {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define MK_LN_EPS 1e-8f
constexpr int MK_THREADS = 1024;
constexpr int MK_WAVES = 16;
constexpr int MK_MAXPF = 5;                   // float4 prefetch registers per thread
constexpr int MK_LDS_IN = 17408;              // floats: 256 rows x (64+4)  (>= 129 row pairs x 132)
constexpr int MK_LDS_OUT = 16 * 32 * 36;      // floats: 16 tasks x 32 positions x (32+4)
constexpr size_t MK_LDS_BYTES = (MK_LDS_IN + MK_LDS_OUT) * sizeof(float);

struct StageGeom {   // how one (time tap, channel chunk) of a conv input is laid out in LDS
  int rows;          // input rows to stage (incl. halo)
  int cc;            // channels per chunk (32 or 64)
  int stride, padl, F_in, pitch;
};

__device__ __forceinline__ StageGeom make_geom(const ConvShape& sh, const ConvParams& p) {
  StageGeom g;
  g.cc = sh.cin < 64 ? sh.cin : 64;
  g.stride = sh.stride;
  g.padl = sh.padl;
  g.F_in = p.F_in;
  g.pitch = sh.stride == 1 ? g.cc + 4 : 2 * g.cc + 4;
  g.rows = sh.stride == 1 ? p.F_out + sh.kf - 1 : 2 * (p.F_out + (sh.kf - 1) / 2);
  return g;
}

// global -> registers (issue only; the wait happens at the first use in stage_store)
__device__ __forceinline__ void stage_load(const float* src, int src_ld, const StageGeom& g, int tid, f32x4 (&pf)[MK_MAXPF]) {
  const int cc4 = g.cc >> 2;
  const int n = g.rows * cc4;
#pragma unroll
  for (int i = 0; i < MK_MAXPF; ++i) {
    const int q = tid + i * MK_THREADS;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (q < n) {
      const int c4 = q & (cc4 - 1);
      const int lr = q / cc4;
      const int gr = lr - g.padl;
      if (gr >= 0 && gr < g.F_in) v = *reinterpret_cast<const f32x4*>(src + static_cast<size_t>(gr) * src_ld + 4 * c4);
    }
    pf[i] = v;
  }
}

// registers -> LDS
__device__ __forceinline__ void stage_store(float* lds_in, const StageGeom& g, int tid, const f32x4 (&pf)[MK_MAXPF]) {
  const int cc4 = g.cc >> 2;
  const int n = g.rows * cc4;
#pragma unroll
  for (int i = 0; i < MK_MAXPF; ++i) {
    const int q = tid + i * MK_THREADS;
    if (q < n) {
      const int c4 = q & (cc4 - 1);
      const int lr = q / cc4;
      const int la = g.stride == 1 ? lr * g.pitch + 4 * c4 : (lr >> 1) * g.pitch + (lr & 1) * g.cc + 4 * c4;
      *reinterpret_cast<f32x4*>(lds_in + la) = pf[i];
    }
  }
}

__device__ __forceinline__ ConvShape dev_conv_shape(int k) {
  //                     cin  nt  s  tt kf padl ln g      (mirror of conv_shape() in kernels.hip)
  switch (k) {
    case CONV_EL_C32:   return {32,  1, 2, 2, 3, 1, 1, 1};
    case CONV_EL_C64:   return {64,  1, 2, 2, 3, 1, 1, 1};
    case CONV_EL_C128:  return {128, 1, 2, 2, 3, 1, 1, 1};
    case CONV_DL_N64:   return {64,  2, 1, 2, 3, 1, 1, 1};
    case CONV_DL_N128:  return {64,  4, 1, 2, 3, 1, 1, 2};
    case CONV_IN_C64:   return {64,  2, 1, 1, 1, 0, 1, 2};
    case CONV_IN_C128:  return {128, 2, 1, 1, 1, 0, 1, 2};
    case CONV_DOWN:     return {64,  2, 2, 1, 3, 0, 0, 2};
    case CONV_UP_EVEN:  return {128, 4, 1, 1, 2, 1, 0, 4};
    default:            return {128, 4, 1, 1, 1, 0, 0, 4};   // CONV_UP_ODD
  }
}

// Row-wise epilogue: LPG lanes per output row (row = LPG*4 channels), optional LN + PReLU.
template <int LPG, bool LN>
__device__ __forceinline__ void conv_epilogue(const ConvParams& p, int stream, const float* lds_out, int KS, int slot_floats,
                                              int opitch, int R, int tid) {
  constexpr int GC = LPG * 4;
  const int li = tid & (LPG - 1);
  const int units = p.F_out * R;
  const f32x4 bias_dummy = {0.f, 0.f, 0.f, 0.f};
  (void)bias_dummy;
  f32x4 gm = {1.f, 1.f, 1.f, 1.f}, bt = {0.f, 0.f, 0.f, 0.f};
  if (LN) {
    gm = *reinterpret_cast<const f32x4*>(p.gamma + 4 * li);
    bt = *reinterpret_cast<const f32x4*>(p.beta + 4 * li);
  }
  float* d0 = p.dst0 + static_cast<size_t>(stream) * (p.F_out * p.row_mul) * p.ld0;
  float* d1 = p.dst1 ? p.dst1 + static_cast<size_t>(stream) * (p.F_out * p.row_mul) * p.ld1 : nullptr;
  for (int u = tid / LPG; u < units; u += MK_THREADS / LPG) {
    const int pos = u / R, gi = u - pos * R;      // R is 1 or 2
    const int ch = gi * GC + 4 * li;
    f32x4 v = *reinterpret_cast<const f32x4*>(p.bias + ch);
    const float* o = lds_out + pos * opitch + ch;
    for (int ks = 0; ks < KS; ++ks) v += *reinterpret_cast<const f32x4*>(o + ks * slot_floats);
    if (LN) {
      float s = v[0] + v[1] + v[2] + v[3];
#pragma unroll
      for (int m = 1; m < LPG; m <<= 1) s += __shfl_xor(s, m);
      const float mean = s * (1.0f / GC);
      v -= mean;
      float q = v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
#pragma unroll
      for (int m = 1; m < LPG; m <<= 1) q += __shfl_xor(q, m);
      const float rstd = 1.0f / sqrtf(q * (1.0f / GC) + MK_LN_EPS);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float y = v[i] * rstd * gm[i] + bt[i];
        v[i] = y >= 0.f ? y : p.alpha * y;
      }
    }
    const size_t row = static_cast<size_t>(pos) * p.row_mul + p.row_add + gi;
    *reinterpret_cast<f32x4*>(d0 + row * p.ld0 + 4 * li) = v;
    if (d1) *reinterpret_cast<f32x4*>(d1 + row * p.ld1 + 4 * li) = v;
  }
}

// One conv-like layer for one stream.  `pf` holds (on entry) the already-issued loads of this
// layer's phase 0 when `have_pf`; on exit it may hold the next layer's phase-0 loads (see caller).
__device__ __forceinline__ void conv_layer(const ConvParams& p, int kind, int stream, float* lds_in, float* lds_out, int tid,
                                           f32x4 (&pf)[MK_MAXPF], bool have_pf, const DevLaunch* next, int next_stream_ok) {
  const ConvShape sh = dev_conv_shape(kind);
  const StageGeom g = make_geom(sh, p);
  const int lane = tid & 63, wave = tid >> 6;
  const int pl = lane & 31, h = lane >> 5;
  const int nch = sh.cin / g.cc;
  const int nph = sh.tt * nch;
  const int gshift = g.cc == 64 ? 3 : 2;          // log2(cc/8)
  const int gpp = sh.kf << gshift;                 // channel groups (8 ch) per phase
  const int F_out = p.F_out;
  const int PT = (F_out + 31) >> 5;
  const int NT = sh.nt;
  const int tiles = PT * NT;
  // split K so that up to 16 waves have a task; the slice count must divide the groups per phase
  int KS = MK_WAVES / tiles;
  const int ksmax = (gpp % 16 == 0) ? 16 : (gpp % 8 == 0) ? 8 : 4;
  if (KS > ksmax) KS = ksmax;
  if (KS < 1) KS = 1;
  const int gpk = gpp / KS;
  const bool active = wave < tiles * KS;
  const int ks = wave / tiles, tl = wave - ks * tiles;
  const int pt = tl / NT, nt = tl - pt * NT;
  int pc = pt * 32 + pl;
  if (pc > F_out - 1) pc = F_out - 1;             // padding lanes recompute the last position
  const int lbase = pc * g.pitch + 4 * h;

  const float* s0 = p.src0 + static_cast<size_t>(stream) * p.F_in * p.src_ld;
  const float* s1 = p.src1 ? p.src1 + static_cast<size_t>(stream) * p.F_in * p.src_ld : s0;

  if (!have_pf) stage_load(s0, p.src_ld, g, tid, pf);
  stage_store(lds_in, g, tid, pf);
  __syncthreads();

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const f32x4* wbase = reinterpret_cast<const f32x4*>(p.wpk) + lane;

#pragma unroll 1
  for (int ph = 0; ph < nph; ++ph) {
    const bool more = ph + 1 < nph;
    if (more) {
      const int t = (ph + 1) / nch, ch = (ph + 1) - t * nch;
      stage_load((t ? s1 : s0) + ch * g.cc, p.src_ld, g, tid, pf);
    }
    if (active) {
      const int g0 = ks * gpk;
      const f32x4* wp = wbase + (static_cast<size_t>(ph * gpp + g0) * NT + nt) * 64;
      const int wstep = NT * 64;
      // depth-2 software pipeline over the 8-channel groups
      f32x4 a0 = wp[0], a1 = {0.f, 0.f, 0.f, 0.f};
      int kf = g0 >> gshift, gg = g0 & ((1 << gshift) - 1);
      int koff = (g.stride == 1) ? kf * g.pitch : ((kf >> 1) * g.pitch + (kf & 1) * g.cc);
      f32x4 b0 = *reinterpret_cast<const f32x4*>(lds_in + lbase + koff + 8 * gg), b1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
      for (int gi = 0; gi < gpk; ++gi) {
        if (gi + 1 < gpk) {
          const int gn = g0 + gi + 1;
          kf = gn >> gshift;
          gg = gn & ((1 << gshift) - 1);
          koff = (g.stride == 1) ? kf * g.pitch : ((kf >> 1) * g.pitch + (kf & 1) * g.cc);
          a1 = wp[(gi + 1) * wstep];
          b1 = *reinterpret_cast<const f32x4*>(lds_in + lbase + koff + 8 * gg);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], b0[j], acc, 0, 0, 0);
        a0 = a1;
        b0 = b1;
      }
    }
    __syncthreads();             // every wave is done reading this phase's LDS rows
    if (more) {
      stage_store(lds_in, g, tid, pf);
      __syncthreads();
    }
  }

  // cross-layer prefetch: the next conv layer's previous-frame tap does not depend on this frame
  bool next_pf = false;
  if (next && next->op == DEV_OP_CONV && next_stream_ok) {
    const ConvShape nsh = dev_conv_shape(next->ck);
    if (nsh.tt == 2) {
      const StageGeom ng = make_geom(nsh, next->conv);
      stage_load(next->conv.src0 + static_cast<size_t>(stream) * next->conv.F_in * next->conv.src_ld, next->conv.src_ld, ng, tid, pf);
      next_pf = true;
    }
  }
  (void)next_pf;

  // partial tiles -> LDS exchange buffer [ks][pos][32*NT (+4)]
  const int opitch = 32 * NT + 4;
  const int slot_floats = PT * 32 * opitch;
  if (active) {
    float* o = lds_out + ks * slot_floats + (pt * 32 + pl) * opitch + nt * 32 + 4 * h;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 v = {acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]};
      *reinterpret_cast<f32x4*>(o + 8 * q) = v;
    }
  }
  __syncthreads();
  const int R = NT / sh.g;
  if (sh.epi_ln) {
    if (sh.g == 1) conv_epilogue<8, true>(p, stream, lds_out, KS, slot_floats, opitch, R, tid);
    else conv_epilogue<16, true>(p, stream, lds_out, KS, slot_floats, opitch, R, tid);
  } else {
    if (sh.g == 2) conv_epilogue<16, false>(p, stream, lds_out, KS, slot_floats, opitch, R, tid);
    else conv_epilogue<32, false>(p, stream, lds_out, KS, slot_floats, opitch, R, tid);
  }
  __syncthreads();               // stores visible to the whole workgroup, LDS free for the next layer
}

__device__ __forceinline__ float mk_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

// LSTM cell + Dense for one stream (models/proposed.py:70-119; converter_proposed.py:234-237):
// 8 K-slices x 128 gate slots, reduced through LDS.
__device__ __forceinline__ void lstm_layer(const LstmParams& p, int stream, float* lds, int tid) {
  float* v = lds;              // [256]
  float* hs = lds + 256;       // [32]
  float* part = lds + 288;     // [8][84]
  float* z = lds + 288 + 8 * 84;   // [96]
  float* hn = z + 96;          // [32]
  for (int k = tid; k < p.Din; k += MK_THREADS) {
    const int f = k / p.x_cols, c = k - f * p.x_cols;
    v[k] = p.x[(static_cast<size_t>(stream) * p.x_rows + f) * p.x_ld + c];
  }
  if (tid < 21) hs[tid] = p.h_in[static_cast<size_t>(stream) * 21 + tid];
  __syncthreads();
  {
    const int n = tid & 127, sl = tid >> 7;
    if (n < 84) {
      const int kn = p.Din >> 3, k0 = sl * kn;
      float a = 0.f;
      for (int k = k0; k < k0 + kn; ++k) a = fmaf(p.wxT[k * 84 + n], v[k], a);
      part[sl * 84 + n] = a;
    }
  }
  __syncthreads();
  if (tid < 84) {
    // same association as the per-layer kernel / oracle: (bias + Wx v) + Wh h
    float a = p.bias[tid];
#pragma unroll
    for (int s = 0; s < 8; ++s) a += part[s * 84 + tid];
    float r = 0.f;
    for (int u = 0; u < 21; ++u) r = fmaf(p.whT[u * 84 + tid], hs[u], r);
    z[tid] = a + r;
  }
  __syncthreads();
  if (tid < 21) {
    const float gi = mk_sigmoid(z[tid]), gf = mk_sigmoid(z[21 + tid]);
    const float gg = tanhf(z[42 + tid]), go = mk_sigmoid(z[63 + tid]);
    const float c_new = gf * p.c_in[static_cast<size_t>(stream) * 21 + tid] + gi * gg;
    const float h_new = go * tanhf(c_new);
    p.c_out[static_cast<size_t>(stream) * 21 + tid] = c_new;
    p.h_out[static_cast<size_t>(stream) * 21 + tid] = h_new;
    hn[tid] = h_new;
  }
  __syncthreads();
  for (int m = tid; m < p.Dout; m += MK_THREADS) {
    float a = p.bd[m];
#pragma unroll
    for (int u = 0; u < 21; ++u) a = fmaf(p.wdT[u * p.Dout + m], hn[u], a);
    const int f = m / p.dst_cols, c = m - f * p.dst_cols;
    p.dst[(static_cast<size_t>(stream) * p.dst_rows + f) * p.dst_ld + c] = a;
  }
  __syncthreads();
}

// CTFA gate + residual for one stream (ctfa_rt, models/proposed.py:162-196).
__device__ __forceinline__ void ctfa_layer(const CtfaParams& p, int stream, float* lds, int tid) {
  float* part = lds;               // [64][64]
  float* m = lds + 4096;           // [64]
  float* hid = m + 64;             // [16]
  float* ta = hid + 16;            // [64]
  float* gate = ta + 64;           // [64]
  const int c4 = tid & 15, rg = tid >> 4;
  const float* xb = p.x + static_cast<size_t>(stream) * p.F * p.x_ld;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  for (int f = rg; f < p.F; f += 64) s += *reinterpret_cast<const f32x4*>(xb + static_cast<size_t>(f) * p.x_ld + 4 * c4);
  *reinterpret_cast<f32x4*>(part + rg * 64 + 4 * c4) = s;
  __syncthreads();
  if (tid < 64) {
    float a = 0.f;
    for (int r = 0; r < 64; ++r) a += part[r * 64 + tid];
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
    ta[tid] = mk_sigmoid(a);
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
    gate[tid] = mk_sigmoid(a) * ta[tid];
  }
  __syncthreads();
  const f32x4 g4 = *reinterpret_cast<const f32x4*>(gate + 4 * c4);
  const float* eb = p.e0 + static_cast<size_t>(stream) * p.F * p.e0_ld;
  float* yb = p.y + static_cast<size_t>(stream) * p.F * p.y_ld;
  for (int f = rg; f < p.F; f += 64) {
    const f32x4 xv = *reinterpret_cast<const f32x4*>(xb + static_cast<size_t>(f) * p.x_ld + 4 * c4);
    const f32x4 ev = *reinterpret_cast<const f32x4*>(eb + static_cast<size_t>(f) * p.e0_ld + 4 * c4);
    *reinterpret_cast<f32x4*>(yb + static_cast<size_t>(f) * p.y_ld + 4 * c4) = xv * g4 + ev;
  }
  __syncthreads();
}

__device__ __forceinline__ void input_layer_op(const InLayerParams& p, int stream, int tid) {
  const int c4 = tid & 15;
  const f32x4 w = *reinterpret_cast<const f32x4*>(p.w + 4 * c4);
  const f32x4 bb = *reinterpret_cast<const f32x4*>(p.b + 4 * c4);
  const f32x4 gm = *reinterpret_cast<const f32x4*>(p.gamma + 4 * c4);
  const f32x4 bt = *reinterpret_cast<const f32x4*>(p.beta + 4 * c4);
  for (int pos = tid >> 4; pos < NUTLS_DEV_BINS; pos += MK_THREADS / 16) {
    const float x = p.x[static_cast<size_t>(stream) * NUTLS_DEV_BINS + pos];
    f32x4 y = w * x + bb;
    float s = y[0] + y[1] + y[2] + y[3];
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) s += __shfl_xor(s, o);
    y -= s * (1.0f / 64.0f);
    float q = y[0] * y[0] + y[1] * y[1] + y[2] * y[2] + y[3] * y[3];
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) q += __shfl_xor(q, o);
    const float rstd = 1.0f / sqrtf(q * (1.0f / 64.0f) + MK_LN_EPS);
    f32x4 o4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float t = y[i] * rstd * gm[i] + bt[i];
      o4[i] = t >= 0.f ? t : p.alpha * t;
    }
    *reinterpret_cast<f32x4*>(p.y + (static_cast<size_t>(stream) * NUTLS_DEV_BINS + pos) * 64 + 4 * c4) = o4;
  }
  __syncthreads();
}

__device__ __forceinline__ void out_conv_op(const OutConvParams& p, int stream, int tid) {
  const int c4 = tid & 15;
  const f32x4 w = *reinterpret_cast<const f32x4*>(p.w + 4 * c4);
  for (int pos = tid >> 4; pos < NUTLS_DEV_BINS; pos += MK_THREADS / 16) {
    const f32x4 xv = *reinterpret_cast<const f32x4*>(p.x + (static_cast<size_t>(stream) * NUTLS_DEV_BINS + pos) * p.x_ld + 4 * c4);
    float s = xv[0] * w[0] + xv[1] * w[1] + xv[2] * w[2] + xv[3] * w[3];
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) s += __shfl_xor(s, o);
    if (c4 == 0) p.y[static_cast<size_t>(stream) * NUTLS_DEV_BINS + pos] = s + p.bias;
  }
  __syncthreads();
}

__global__ __launch_bounds__(MK_THREADS) void nutls_stream_step_kernel(const DevLaunch* __restrict__ plan, int n_ops, int B,
                                                                       unsigned long long* prof) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* lds_in = lds;
  float* lds_out = lds + MK_LDS_IN;
  const int tid = threadIdx.x;
  for (int stream = blockIdx.x; stream < B; stream += gridDim.x) {
    f32x4 pf[MK_MAXPF];
    bool have_pf = false;
#pragma unroll 1
    for (int i = 0; i < n_ops; ++i) {
      const DevLaunch& L = plan[i];
      if (prof && blockIdx.x == 0 && tid == 0) prof[i] = wall_clock64();
      switch (L.op) {
        case DEV_OP_CONV: {
          const DevLaunch* next = (i + 1 < n_ops) ? &plan[i + 1] : nullptr;
          conv_layer(L.conv, L.ck, stream, lds_in, lds_out, tid, pf, have_pf, next, 1);
          have_pf = next && next->op == DEV_OP_CONV && dev_conv_shape(next->ck).tt == 2;
          break;
        }
        case DEV_OP_LSTM: lstm_layer(L.lstm, stream, lds_out, tid); have_pf = false; break;
        case DEV_OP_CTFA: ctfa_layer(L.ctfa, stream, lds_out, tid); have_pf = false; break;
        case DEV_OP_INLAYER: input_layer_op(L.inl, stream, tid); have_pf = false; break;
        default: out_conv_op(L.outc, stream, tid); have_pf = false; break;
      }
    }
    if (prof && blockIdx.x == 0 && tid == 0) prof[n_ops] = wall_clock64();
  }
}

hipError_t launch_stream_step(const DevLaunch* plan, int n_ops, int B, int grid, unsigned long long* prof, hipStream_t s) {
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(nutls_stream_step_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(MK_LDS_BYTES));
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  hipLaunchKernelGGL(nutls_stream_step_kernel, dim3(grid), dim3(MK_THREADS), MK_LDS_BYTES, s, plan, n_ops, B, prof);
  return hipGetLastError();
}

}  // namespace nutls
