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

// Pointers read out of the device-resident plan have no provable address space, so plain
// dereferences become FLAT accesses -- which also tick lgkmcnt and would make every LDS wait drain
// the in-flight weight prefetch.  Everything that lives in HBM is therefore accessed through
// explicit global (address_space(1)) pointers.
typedef const f32x4 __attribute__((address_space(1))) * gc4_t;
typedef f32x4 __attribute__((address_space(1))) * g4_t;
typedef const float __attribute__((address_space(1))) * gcf_t;
typedef float __attribute__((address_space(1))) * gf_t;
__device__ __forceinline__ gc4_t G4(const float* p) { return (gc4_t)(unsigned long long)p; }
__device__ __forceinline__ gc4_t G4(const f32x4* p) { return (gc4_t)(unsigned long long)p; }
__device__ __forceinline__ g4_t G4W(float* p) { return (g4_t)(unsigned long long)p; }
__device__ __forceinline__ gcf_t GF(const float* p) { return (gcf_t)(unsigned long long)p; }
__device__ __forceinline__ gf_t GFW(float* p) { return (gf_t)(unsigned long long)p; }

#define MK_LN_EPS 1e-8f
constexpr int MK_THREADS = 1024;
constexpr int MK_WAVES = 16;
constexpr int MK_MAXPF = 4;                   // float4 prefetch registers per thread (256 rows x 16 chunks / 1024)
constexpr int MK_LDS_IN = 17920;              // floats: >= 256 rows x 68, 129 row pairs x 132, 2 x 130 rows x 68
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

// global -> registers (issue only; the wait happens at the first use in stage_store).  Only rows
// that exist in the stream travel through registers; halo rows are zero-filled by stage_store.
__device__ __forceinline__ void stage_load(const float* src, int src_ld, const StageGeom& g, int tid, f32x4 (&pf)[MK_MAXPF]) {
  const int cc4 = g.cc >> 2;
  const int nvalid = (g.rows - g.padl < g.F_in ? g.rows - g.padl : g.F_in) * cc4;
#pragma unroll
  for (int i = 0; i < MK_MAXPF; ++i) {
    const int q = tid + i * MK_THREADS;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (q < nvalid) {
      const int c4 = q & (cc4 - 1);
      const int gr = q / cc4;
      v = *G4(src + static_cast<size_t>(gr) * src_ld + 4 * c4);
    }
    pf[i] = v;
  }
}

__device__ __forceinline__ int stage_lds_addr(const StageGeom& g, int lr, int c4) {
  return g.stride == 1 ? lr * g.pitch + 4 * c4 : (lr >> 1) * g.pitch + (lr & 1) * g.cc + 4 * c4;
}

// registers -> LDS (+ zero halo rows: the ZeroPadding2D of proposed.py:210/:242 and the SAME pad of :255)
__device__ __forceinline__ void stage_store(float* lds_in, const StageGeom& g, int tid, const f32x4 (&pf)[MK_MAXPF]) {
  const int cc4 = g.cc >> 2;
  const int vrows = g.rows - g.padl < g.F_in ? g.rows - g.padl : g.F_in;
  const int nvalid = vrows * cc4;
#pragma unroll
  for (int i = 0; i < MK_MAXPF; ++i) {
    const int q = tid + i * MK_THREADS;
    if (q < nvalid) *reinterpret_cast<f32x4*>(lds_in + stage_lds_addr(g, g.padl + q / cc4, q & (cc4 - 1))) = pf[i];
  }
  const int nhalo = (g.rows - vrows) * cc4;          // <= 3 rows
  if (tid < nhalo) {
    const int hr = tid / cc4;
    const int lr = hr < g.padl ? hr : vrows + hr;    // top halo rows first, then the bottom ones
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    *reinterpret_cast<f32x4*>(lds_in + stage_lds_addr(g, lr, tid & (cc4 - 1))) = z;
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
    gm = *G4(p.gamma + 4 * li);
    bt = *G4(p.beta + 4 * li);
  }
  float* d0 = p.dst0 + static_cast<size_t>(stream) * (p.F_out * p.row_mul) * p.ld0;
  float* d1 = p.dst1 ? p.dst1 + static_cast<size_t>(stream) * (p.F_out * p.row_mul) * p.ld1 : nullptr;
  for (int u = tid / LPG; u < units; u += MK_THREADS / LPG) {
    const int pos = u / R, gi = u - pos * R;      // R is 1 or 2
    const int ch = gi * GC + 4 * li;
    f32x4 v = *G4(p.bias + ch);
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
    *G4W(d0 + row * p.ld0 + 4 * li) = v;
    if (d1) *G4W(d1 + row * p.ld1 + 4 * li) = v;
  }
}

// This wave's weight stream: fragments are consumed in order (round, group); a RING-deep register
// ring holds the next RING fragments, whose loads were issued RING groups earlier.  All cursor
// state is wave-uniform (SGPRs); the only per-lane part of an address is lane*16 bytes.
template <int RING>
struct WeightStream {
  gc4_t base;          // wave-uniform: packed weights + this task's (k-slice, channel tile) offset
  int wstep;           // float4 between consecutive groups  (NT * 64)
  int gpk, rounds;     // groups per slice, slices (one per round) in this wave's stream
  int round_step;      // float4 between the starts of consecutive slices (RG * wstep)
  int pre_gi, pre_r;   // prefetch cursor
  int pre_off;         // float4 offset of the prefetch cursor from base
  f32x4 ring[RING];

  __device__ __forceinline__ void init(gc4_t b, int wstep_, int gpk_, int rounds_, int round_step_, int lane) {
    base = b; wstep = wstep_; gpk = gpk_; rounds = rounds_; round_step = round_step_;
    pre_gi = 0; pre_r = 0; pre_off = 0;
#pragma unroll
    for (int u = 0; u < RING; ++u) { ring[u] = base[pre_off + lane]; advance(); }
  }
  // move the prefetch cursor one group ahead; past the end of the stream it parks on the last
  // fragment (re-loading it is harmless and keeps the loop free of divergent control flow)
  __device__ __forceinline__ void advance() {
    int gi = pre_gi + 1, r = pre_r, off = pre_off + wstep;
    if (gi == gpk) { gi = 0; r += 1; off += round_step - gpk * wstep; }
    const bool end = r >= rounds;
    pre_gi = end ? pre_gi : gi;
    pre_r = end ? pre_r : r;
    pre_off = end ? pre_off : off;
  }
};

// The MFMA part of one round of one task: `gpk` consecutive 8-channel groups starting at group
// `g_first` of the resident LDS phases.
template <int RING>
__device__ __forceinline__ void mfma_slice(f32x16& acc, WeightStream<RING>& ws, const float* lds_lane, int g_first, int gpp,
                                           int gpc, const StageGeom& g, int phase_floats, int lane) {
  // wave-uniform cursor over (local phase, frequency tap, channel group)
  int phl = g_first / gpp;
  int rem = g_first - phl * gpp;
  int kf = rem / gpc, gg = rem - kf * gpc;
  auto b_off = [&]() {
    const int koff = (g.stride == 1) ? kf * g.pitch : ((kf >> 1) * g.pitch + (kf & 1) * g.cc);
    return phl * phase_floats + koff + 8 * gg;
  };
  auto step = [&]() {
    int ngg = gg + 1, nkf = kf, nph = phl;
    if (ngg == gpc) { ngg = 0; nkf += 1; }
    if (nkf * gpc == gpp) { nkf = 0; nph += 1; }
    gg = ngg; kf = nkf; phl = nph;
  };
  f32x4 b_cur = *reinterpret_cast<const f32x4*>(lds_lane + b_off());
#pragma unroll 1
  for (int gi = 0; gi < ws.gpk; gi += RING) {
#pragma unroll
    for (int u = 0; u < RING; ++u) {
      const f32x4 a = ws.ring[u];
      ws.ring[u] = ws.base[ws.pre_off + lane];     // fragment RING groups ahead (parked at the end)
      ws.advance();
      const f32x4 b = b_cur;
      step();
      // next B fragment (the read after the slice's last group is unused; its address stays inside LDS)
      b_cur = *reinterpret_cast<const f32x4*>(lds_lane + ((gi + u + 1 < ws.gpk) ? b_off() : 0));
#pragma unroll
      for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[j], acc, 0, 0, 0);
    }
  }
}

struct ConvTask {        // how one layer is spread over the 16 waves
  int nstage;            // phases resident in LDS at once (1, or all of them for small layers)
  int rounds;            // nph / nstage
  int RG;                // 8-channel groups per round = nstage * gpp
  int KS, gpk;           // K slices per round, groups per slice
  int PT, NT, tiles;
  int phase_floats;      // LDS floats of one staged phase
  int per_thread;        // float4 staged per thread per phase
};

__device__ __forceinline__ ConvTask plan_task(const ConvShape& sh, const StageGeom& g, int F_out, int ring) {
  ConvTask t;
  const int nch = sh.cin / g.cc, nph = sh.tt * nch;
  const int gpp = sh.kf * (g.cc >> 3);
  t.phase_floats = (sh.stride == 1 ? g.rows : (g.rows >> 1)) * g.pitch;
  const int n4 = (g.rows - g.padl < g.F_in ? g.rows - g.padl : g.F_in) * (g.cc >> 2);
  t.per_thread = (n4 + MK_THREADS - 1) / MK_THREADS;
  // small layers: keep every (time tap, channel chunk) resident -> one barrier, deeper split-K
  const bool merge = (nph * t.phase_floats <= MK_LDS_IN) && (nch * t.per_thread <= MK_MAXPF);
  t.nstage = merge ? nph : 1;
  t.rounds = nph / t.nstage;
  t.RG = t.nstage * gpp;
  t.PT = (F_out + 31) >> 5;
  t.NT = sh.nt;
  t.tiles = t.PT * t.NT;
  int ks = MK_WAVES / t.tiles;
  if (ks < 1) ks = 1;
  int kmax = t.RG / ring;              // slices must hold whole rings
  kmax = kmax & (-kmax);               // largest power-of-two divisor
  t.KS = ks < kmax ? ks : kmax;
  t.gpk = t.RG / t.KS;
  return t;
}

// One conv-like layer for one stream.  On entry `pf` may already hold the issued loads of this
// layer's previous-frame tap (`have_pf`); on exit it holds the next layer's when that layer is a
// two-tap conv (its previous-frame rows never depend on the current frame).
template <int RING>
__device__ __forceinline__ void conv_layer(const ConvParams& p, const ConvShape& sh, int stream, float* lds_in, float* lds_out,
                                           int tid, f32x4 (&pf)[MK_MAXPF], bool have_pf, const DevLaunch* next) {
  const StageGeom g = make_geom(sh, p);
  const ConvTask T = plan_task(sh, g, p.F_out, RING);
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // wave-uniform -> SGPR
  const int pl = lane & 31, h = lane >> 5;
  const int nch = sh.cin / g.cc;
  const int gpc = g.cc >> 3;                       // 8-channel groups per frequency tap
  const int gpp = sh.kf * gpc;
  const bool active = wave < T.tiles * T.KS;
  const int ks = wave / T.tiles, tl = wave - ks * T.tiles;
  const int pt = tl / T.NT, nt = tl - pt * T.NT;
  int pc = pt * 32 + pl;
  if (pc > p.F_out - 1) pc = p.F_out - 1;          // padding lanes recompute the last position
  const float* lds_lane = lds_in + pc * g.pitch + 4 * h;
  const int wstep = T.NT * 64;

  const float* s0 = p.src0 + static_cast<size_t>(stream) * p.F_in * p.src_ld;
  const float* s1 = p.src1 ? p.src1 + static_cast<size_t>(stream) * p.F_in * p.src_ld : s0;

  // ---- weight ring: first RING fragments of this wave's stream, issued before anything else
  WeightStream<RING> ws;
  if (active)
    ws.init(G4(p.wpk) + (static_cast<size_t>(ks * T.gpk) * T.NT + nt) * 64, wstep, T.gpk, T.rounds,
            T.RG * wstep, lane);

  // ---- stage round 0
  if (T.nstage == 1) {
    if (!have_pf) stage_load(s0, p.src_ld, g, tid, pf);
    stage_store(lds_in, g, tid, pf);
  } else {
    // all phases resident: phase ph = t*nch + ch at lds_in + ph*phase_floats
    for (int ph = 0; ph < T.nstage; ++ph) {
      const int t = ph / nch, ch = ph - t * nch;
      if (!(have_pf && ph == 0)) stage_load((t ? s1 : s0) + ch * g.cc, p.src_ld, g, tid, pf);
      stage_store(lds_in + ph * T.phase_floats, g, tid, pf);
    }
  }
  __syncthreads();

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;

#pragma unroll 1
  for (int rd = 0; rd < T.rounds; ++rd) {
    const bool more = rd + 1 < T.rounds;
    if (more) {   // nstage == 1 here: prefetch the next phase's rows into registers
      const int t = (rd + 1) / nch, ch = (rd + 1) - t * nch;
      stage_load((t ? s1 : s0) + ch * g.cc, p.src_ld, g, tid, pf);
    }
    if (active) mfma_slice<RING>(acc, ws, lds_lane, ks * T.gpk, gpp, gpc, g, T.phase_floats, lane);
    if (more) {
      __syncthreads();           // every wave is done reading this phase's LDS rows
      stage_store(lds_in, g, tid, pf);
      __syncthreads();
    }
  }

  // ---- cross-layer prefetch of the next layer's previous-frame tap (independent of this frame)
  if (next && next->op == DEV_OP_CONV) {
    const ConvShape nsh = dev_conv_shape(next->ck);
    if (nsh.tt == 2) {
      const StageGeom ng = make_geom(nsh, next->conv);
      stage_load(next->conv.src0 + static_cast<size_t>(stream) * next->conv.F_in * next->conv.src_ld, next->conv.src_ld, ng, tid, pf);
    }
  }

  // ---- partial tiles -> LDS exchange buffer [ks][pos][32*NT (+4)]
  const int opitch = 32 * T.NT + 4;
  const int slot_floats = T.PT * 32 * opitch;
  if (active) {
    float* o = lds_out + ks * slot_floats + (pt * 32 + pl) * opitch + nt * 32 + 4 * h;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 v = {acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]};
      *reinterpret_cast<f32x4*>(o + 8 * q) = v;
    }
  }
  __syncthreads();
  const int R = T.NT / sh.g;
  if (sh.epi_ln) {
    if (sh.g == 1) conv_epilogue<8, true>(p, stream, lds_out, T.KS, slot_floats, opitch, R, tid);
    else conv_epilogue<16, true>(p, stream, lds_out, T.KS, slot_floats, opitch, R, tid);
  } else {
    if (sh.g == 2) conv_epilogue<16, false>(p, stream, lds_out, T.KS, slot_floats, opitch, R, tid);
    else conv_epilogue<32, false>(p, stream, lds_out, T.KS, slot_floats, opitch, R, tid);
  }
  __syncthreads();               // stores visible to the whole workgroup, LDS free for the next layer
}

__device__ __forceinline__ float mk_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
  return v;
}

// LSTM cell + Dense for one stream (models/proposed.py:70-119; converter_proposed.py:234-237):
// 8 K-slices x 128 gate slots, reduced through LDS.  All weight loads are issued up front
// (independent of the gathered input) so only one memory latency is exposed.
__device__ __forceinline__ void lstm_layer(const LstmParams& p, int stream, float* lds, int tid) {
  float* v = lds;              // [256]
  float* hs = lds + 256;       // [32]
  float* part = lds + 288;     // [8][84]
  float* z = lds + 288 + 8 * 84;   // [96]
  float* hn = z + 96;          // [32]
  const int n = tid & 127, sl = tid >> 7;
  const int kn = p.Din >> 3, k0 = sl * kn;       // kn in {4, 8, 16, 32}
  // first batch of input weights + recurrent weights: issued before the input gather's barrier
  float w[8];
  if (n < 84) {
#pragma unroll
    for (int k = 0; k < 8; ++k) w[k] = (k < kn) ? GF(p.wxT)[(k0 + k) * 84 + n] : 0.f;
  }
  for (int k = tid; k < p.Din; k += MK_THREADS) {
    const int f = k / p.x_cols, c = k - f * p.x_cols;
    v[k] = GF(p.x)[(static_cast<size_t>(stream) * p.x_rows + f) * p.x_ld + c];
  }
  float c_old = 0.f;
  if (tid < 21) {
    hs[tid] = GF(p.h_in)[static_cast<size_t>(stream) * 21 + tid];
    c_old = GF(p.c_in)[static_cast<size_t>(stream) * 21 + tid];
  }
  __syncthreads();
  if (n < 84) {
    float a = 0.f;
#pragma unroll 1
    for (int kb = 0; kb < kn; kb += 8) {
      float wn[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) wn[k] = (kb + 8 + k < kn) ? GF(p.wxT)[(k0 + kb + 8 + k) * 84 + n] : 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (kb + k < kn) a = fmaf(w[k], v[k0 + kb + k], a);
#pragma unroll
      for (int k = 0; k < 8; ++k) w[k] = wn[k];
    }
    part[sl * 84 + n] = a;
  }
  float bias = 0.f, r = 0.f;
  if (tid < 84) {
    bias = GF(p.bias)[tid];
#pragma unroll
    for (int u = 0; u < 21; ++u) r = fmaf(GF(p.whT)[u * 84 + tid], hs[u], r);
  }
  __syncthreads();
  if (tid < 84) {
    float a = bias;
#pragma unroll
    for (int s = 0; s < 8; ++s) a += part[s * 84 + tid];
    z[tid] = a + r;
  }
  // dense weights for the output rows this thread owns (Dout <= 256 -> one row per thread)
  float wd[21];
  float bd = 0.f;
  if (tid < p.Dout) {
#pragma unroll
    for (int u = 0; u < 21; ++u) wd[u] = GF(p.wdT)[u * p.Dout + tid];
    bd = GF(p.bd)[tid];
  }
  __syncthreads();
  if (tid < 21) {
    const float gi = mk_sigmoid(z[tid]), gf = mk_sigmoid(z[21 + tid]);
    const float gg = tanhf(z[42 + tid]), go = mk_sigmoid(z[63 + tid]);
    const float c_new = gf * c_old + gi * gg;
    const float h_new = go * tanhf(c_new);
    GFW(p.c_out)[static_cast<size_t>(stream) * 21 + tid] = c_new;
    GFW(p.h_out)[static_cast<size_t>(stream) * 21 + tid] = h_new;
    hn[tid] = h_new;
  }
  __syncthreads();
  if (tid < p.Dout) {
    float a = bd;
#pragma unroll
    for (int u = 0; u < 21; ++u) a = fmaf(wd[u], hn[u], a);
    const int f = tid / p.dst_cols, c = tid - f * p.dst_cols;
    GFW(p.dst)[(static_cast<size_t>(stream) * p.dst_rows + f) * p.dst_ld + c] = a;
  }
  __syncthreads();
}

// CTFA gate + residual for one stream (ctfa_rt, models/proposed.py:162-196).
// Wave u computes hidden unit u of the 64->16 layers (one product per lane + wave reduction);
// the MLP weights are fetched before the mean-over-F reduction so their latency is hidden.
__device__ __forceinline__ void ctfa_layer(const CtfaParams& p, int stream, float* lds, int tid) {
  float* part = lds;               // [64][64]
  float* m = lds + 4096;           // [64]
  float* hid = m + 64;             // [16]
  float* ta = hid + 16;            // [64]
  float* gate = ta + 64;           // [64]
  const int c4 = tid & 15, rg = tid >> 4;
  const int lane = tid & 63, wave = tid >> 6;
  const float w1_ta = GF(p.ta_w1T)[lane * 16 + wave], w1_fa = GF(p.fa_w1T)[lane * 16 + wave];
  const float b1_ta = GF(p.ta_b1)[wave], b1_fa = GF(p.fa_b1)[wave];
  float w2_ta[16], w2_fa[16];
  float b2_ta = 0.f, b2_fa = 0.f;
  if (tid < 64) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      w2_ta[u] = GF(p.ta_w2T)[u * 64 + tid];
      w2_fa[u] = GF(p.fa_w2T)[u * 64 + tid];
    }
    b2_ta = GF(p.ta_b2)[tid];
    b2_fa = GF(p.fa_b2)[tid];
  }
  const float* xb = p.x + static_cast<size_t>(stream) * p.F * p.x_ld;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  for (int f = rg; f < p.F; f += 64) s += *G4(xb + static_cast<size_t>(f) * p.x_ld + 4 * c4);
  *reinterpret_cast<f32x4*>(part + rg * 64 + 4 * c4) = s;
  __syncthreads();
  if (tid < 64) {
    float a = 0.f;
#pragma unroll 16
    for (int r = 0; r < 64; ++r) a += part[r * 64 + tid];
    m[tid] = a / static_cast<float>(p.F);
  }
  __syncthreads();
  {
    const float t = wave_sum(w1_ta * m[lane]);
    if (lane == 0) hid[wave] = fmaxf(t + b1_ta, 0.f);
  }
  __syncthreads();
  float ta_c = 0.f;
  if (tid < 64) {
    float a = b2_ta;
#pragma unroll
    for (int u = 0; u < 16; ++u) a = fmaf(w2_ta[u], hid[u], a);
    ta_c = mk_sigmoid(a);
    ta[tid] = ta_c;
  }
  __syncthreads();
  {
    const float t = wave_sum(w1_fa * (ta[lane] * (1.0f / 32.0f)));
    __syncthreads();             // everyone has read hid (TA pass) before it is overwritten
    if (lane == 0) hid[wave] = fmaxf(t + b1_fa, 0.f);
  }
  __syncthreads();
  if (tid < 64) {
    float a = b2_fa;
#pragma unroll
    for (int u = 0; u < 16; ++u) a = fmaf(w2_fa[u], hid[u], a);
    gate[tid] = mk_sigmoid(a) * ta_c;
  }
  __syncthreads();
  const f32x4 g4 = *reinterpret_cast<const f32x4*>(gate + 4 * c4);
  const float* eb = p.e0 + static_cast<size_t>(stream) * p.F * p.e0_ld;
  float* yb = p.y + static_cast<size_t>(stream) * p.F * p.y_ld;
  for (int f = rg; f < p.F; f += 64) {
    const f32x4 xv = *G4(xb + static_cast<size_t>(f) * p.x_ld + 4 * c4);
    const f32x4 ev = *G4(eb + static_cast<size_t>(f) * p.e0_ld + 4 * c4);
    *G4W(yb + static_cast<size_t>(f) * p.y_ld + 4 * c4) = xv * g4 + ev;
  }
  __syncthreads();
}

__device__ __forceinline__ void input_layer_op(const InLayerParams& p, int stream, int tid) {
  const int c4 = tid & 15;
  const f32x4 w = *G4(p.w + 4 * c4);
  const f32x4 bb = *G4(p.b + 4 * c4);
  const f32x4 gm = *G4(p.gamma + 4 * c4);
  const f32x4 bt = *G4(p.beta + 4 * c4);
  for (int pos = tid >> 4; pos < NUTLS_DEV_BINS; pos += MK_THREADS / 16) {
    const float x = GF(p.x)[static_cast<size_t>(stream) * NUTLS_DEV_BINS + pos];
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
    *G4W(p.y + (static_cast<size_t>(stream) * NUTLS_DEV_BINS + pos) * 64 + 4 * c4) = o4;
  }
  __syncthreads();
}

__device__ __forceinline__ void out_conv_op(const OutConvParams& p, int stream, int tid) {
  const int c4 = tid & 15;
  const f32x4 w = *G4(p.w + 4 * c4);
  for (int pos = tid >> 4; pos < NUTLS_DEV_BINS; pos += MK_THREADS / 16) {
    const f32x4 xv = *G4(p.x + (static_cast<size_t>(stream) * NUTLS_DEV_BINS + pos) * p.x_ld + 4 * c4);
    float s = xv[0] * w[0] + xv[1] * w[1] + xv[2] * w[2] + xv[3] * w[3];
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) s += __shfl_xor(s, o);
    if (c4 == 0) GFW(p.y)[static_cast<size_t>(stream) * NUTLS_DEV_BINS + pos] = s + p.bias;
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
    int i = 0;
#pragma unroll 1
    while (i < n_ops) {
      if (plan[i].op == DEV_OP_CONV) {
        // a run of consecutive conv layers: the activation prefetch registers live only here
        f32x4 pf[MK_MAXPF];
        bool have_pf = false;
#pragma unroll 1
        while (i < n_ops && plan[i].op == DEV_OP_CONV) {
          const DevLaunch& L = plan[i];
          if (prof && blockIdx.x == 0 && tid == 0) prof[i] = wall_clock64();
          const DevLaunch* next = (i + 1 < n_ops) ? &plan[i + 1] : nullptr;
          const ConvShape sh = dev_conv_shape(L.ck);
          if (sh.kf == 3) conv_layer<3>(L.conv, sh, stream, lds_in, lds_out, tid, pf, have_pf, next);
          else conv_layer<4>(L.conv, sh, stream, lds_in, lds_out, tid, pf, have_pf, next);
          have_pf = next && next->op == DEV_OP_CONV && dev_conv_shape(next->ck).tt == 2;
          ++i;
        }
      } else {
        const DevLaunch& L = plan[i];
        if (prof && blockIdx.x == 0 && tid == 0) prof[i] = wall_clock64();
        switch (L.op) {
          case DEV_OP_LSTM: lstm_layer(L.lstm, stream, lds_out, tid); break;
          case DEV_OP_CTFA: ctfa_layer(L.ctfa, stream, lds_out, tid); break;
          case DEV_OP_INLAYER: input_layer_op(L.inl, stream, tid); break;
          default: out_conv_op(L.outc, stream, tid); break;
        }
        ++i;
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
