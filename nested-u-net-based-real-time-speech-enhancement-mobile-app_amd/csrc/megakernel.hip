// Persistent "one workgroup per stream" kernel: the whole NUNet-TLS-LSTM frame step of one stream
// runs inside ONE 512-thread workgroup (8 waves = 2 per SIMD, one workgroup per CU), layer after
// layer, driven by the device-resident plan (DevLaunch[]).  Streams are independent (SURVEY.md
// section 8e), so no inter-workgroup synchronisation exists: a layer boundary is a
// __syncthreads(), not a kernel boundary.  At B = 256 this is exactly one stream per CU.
//
// Per conv-like layer (reference blocks: models/proposed.py:198-265):
//   * the input rows ([prev ; cur] time taps x 64-channel chunks, zero halo) live in an LDS image;
//     for small layers the WHOLE image (all phases) is resident and is completed by the PREVIOUS
//     layer: its epilogue forwards the rows it just produced straight into the next layer's image
//     and stores the rest (previous-frame tap, skip-connection channels) from registers whose
//     global loads were issued before that layer's MFMA work ("hand-off": no global round trip and
//     no staging step on the critical path);
//   * the GEMM  D[ch,pos] = W[ch,k] X[k,pos]  is cut into (position tile, channel tile, K slice)
//     tasks of 32x32 outputs (two position tiles per wave, sharing the weight fragments, when a layer
//     has 16 tiles), so even a 4-position layer keeps 6..8 waves busy
//     (split-K); v_mfma_f32_32x32x2_f32 (exact fp32); weight fragments stream from L2 through a
//     double-buffered 4-fragment register ring, the first chunk already fetched by the previous layer;
//   * partial tiles meet in an LDS exchange buffer [k-slice][position][channel];
//   * the epilogue re-reads it row-wise (8/16/32 lanes per output row, float4 per lane): K-slice sum +
//     bias, LayerNorm over the row's channels with DPP shuffles, PReLU, full-line channels-last stores
//     to the (up to two) destination state tensors in HBM.
#include <hip/hip_runtime.h>

#include "nutls_internal.hpp"
#include "ddb_device.hpp"

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
__device__ __forceinline__ g4_t G4W(float* p) { return (g4_t)(unsigned long long)p; }
__device__ __forceinline__ gcf_t GF(const float* p) { return (gcf_t)(unsigned long long)p; }
__device__ __forceinline__ gf_t GFW(float* p) { return (gf_t)(unsigned long long)p; }

// Wave-uniform base (SGPR pair) + 32-bit unsigned float offset (VGPR): lowers to the
// "global_load v, v_off, s[base]" addressing form -- no 64-bit VALU address arithmetic.
typedef const char __attribute__((address_space(1))) * gcb_t;
typedef char __attribute__((address_space(1))) * gb_t;
__device__ __forceinline__ f32x4 ld4(const float* ubase, unsigned foff) {
  return *(gc4_t)((gcb_t)(unsigned long long)ubase + static_cast<unsigned long long>(foff * 4u));
}
__device__ __forceinline__ float ld1(const float* ubase, unsigned foff) {
  return *(gcf_t)((gcb_t)(unsigned long long)ubase + static_cast<unsigned long long>(foff * 4u));
}
__device__ __forceinline__ void st4(float* ubase, unsigned foff, f32x4 v) {
  *(g4_t)((gb_t)(unsigned long long)ubase + static_cast<unsigned long long>(foff * 4u)) = v;
}
__device__ __forceinline__ void st1(float* ubase, unsigned foff, float v) {
  *(gf_t)((gb_t)(unsigned long long)ubase + static_cast<unsigned long long>(foff * 4u)) = v;
}

#define MK_LN_EPS 1e-8f
constexpr int MK_WAVES = MK_NWAVES;           // 8 waves = 2 per SIMD: 256 VGPRs per lane, half the per-layer bookkeeping of 16
constexpr int MK_THREADS = 64 * MK_WAVES;
constexpr int MK_MAXPF = MK_STAGE_ITEMS / MK_THREADS;   // float4 prefetch registers per thread (8)
constexpr int MK_LDS_IN = MK_LDS_IN_FLOATS;
constexpr int MK_LDS_OUT = 16 * 32 * 36;      // floats: 16 tasks x 32 positions x (32+4)
constexpr int MK_LDS_PLAN = MK_MAX_OPS * MK_OP_WORDS;   // dwords: the whole compact plan
constexpr size_t MK_LDS_BYTES = (MK_LDS_IN + MK_LDS_OUT + MK_LDS_PLAN) * sizeof(float);

// Workgroup barrier that orders LDS traffic only: unlike __syncthreads() it does NOT drain vmcnt,
// so global loads issued earlier (next layer's image rows, weight fragments) stay in flight across it.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

#define MK_STAMP(k) do { if (sub && tid == 0) sub[k] = wall_clock64(); } while (0)
// fine-grained cycle stamps of ONE chosen op (debug builds of the profile path only)
#define MK_T(k) do { if (dbg && (tid & 63) == 0) dbg[(tid >> 6) * 16 + k] = clock64(); } while (0)

// ---------------------------------------------------------------------------------------------
//  Compact plan decode (layout: encode_op() in weights.cpp).  All values are wave-uniform; the
//  readfirstlane moves them to SGPRs so the layer code branches and addresses on scalars.
// ---------------------------------------------------------------------------------------------
struct OpWords { unsigned w[MK_OP_WORDS]; };
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ OpWords load_op(const unsigned* lds_plan, int i) {
  OpWords o;
  const u32x4* q = reinterpret_cast<const u32x4*>(lds_plan + i * MK_OP_WORDS);
#pragma unroll
  for (int k = 0; k < MK_OP_WORDS / 4; ++k) {
    const u32x4 v = q[k];
#pragma unroll
    for (int j = 0; j < 4; ++j) o.w[4 * k + j] = __builtin_amdgcn_readfirstlane(v[j]);
  }
  return o;
}
__device__ __forceinline__ int b0(unsigned w) { return w & 255; }
__device__ __forceinline__ int b1(unsigned w) { return (w >> 8) & 255; }
__device__ __forceinline__ int b2(unsigned w) { return (w >> 16) & 255; }
__device__ __forceinline__ int b3(unsigned w) { return w >> 24; }
__device__ __forceinline__ int h0(unsigned w) { return w & 0xFFFF; }
__device__ __forceinline__ int h1(unsigned w) { return w >> 16; }
__device__ __forceinline__ float* aptr(float* base, unsigned off) { return off == MK_NULL_OFF ? nullptr : base + off; }
__device__ __forceinline__ const float* wptr(const float* base, unsigned off) { return off == MK_NULL_OFF ? nullptr : base + off; }

__device__ __forceinline__ void decode_conv(const OpWords& o, const StepArgs& a, ConvParams& p, ConvPlan& c) {
  p.src0 = aptr(a.arena, o.w[0]); p.src1 = aptr(a.arena, o.w[1]);
  p.dst0 = aptr(a.arena, o.w[2]); p.dst1 = aptr(a.arena, o.w[3]);
  p.wpk = wptr(a.wbase, o.w[4]); p.bias = wptr(a.wbase, o.w[5]); p.gamma = wptr(a.wbase, o.w[6]); p.beta = wptr(a.wbase, o.w[7]);
  p.alpha = __uint_as_float(o.w[8]);
  p.src_ld = b0(o.w[9]); p.ld0 = b1(o.w[9]); p.ld1 = b2(o.w[9]);
  const int fl = b3(o.w[9]);
  p.row_mul = fl & 3; p.row_add = (fl >> 2) & 1;
  c.stride = ((fl >> 3) & 1) + 1; c.tt = ((fl >> 4) & 1) + 1; c.epi_ln = (fl >> 5) & 1; c.merged = (fl >> 6) & 1;
  c.staged_by_prev = (fl >> 7) & 1;
  p.F_in = h0(o.w[10]); p.F_out = h1(o.w[10]);
  p.B = a.B; p.sstride = a.sstride; p.log2_fout = 0;
  c.kf = b0(o.w[11]); c.padl = b1(o.w[11]); c.g = b2(o.w[11]); c.nt = b3(o.w[11]);
  c.cc = b0(o.w[12]); c.cc4_shift = b1(o.w[12]); c.n4p_shift = b2(o.w[12]); c.nch_shift = b3(o.w[12]);
  c.pitch = h0(o.w[13]); c.rows = h1(o.w[13]);
  c.vrows = h0(o.w[14]); c.nph = b2(o.w[14]); c.rounds = b3(o.w[14]);
  c.phase_floats = h0(o.w[15]); c.slot_floats = h1(o.w[15]);
  c.RG = b0(o.w[16]); c.KS = b1(o.w[16]); c.gpk = b2(o.w[16]); c.gpc = b3(o.w[16]);
  c.PT = b0(o.w[17]); c.tiles = b1(o.w[17]); c.nt_shift = b2(o.w[17]); c.tw = b3(o.w[17]);
  c.tasks = b0(o.w[18]); c.tasks_shift = b1(o.w[18]); c.opitch = h1(o.w[18]);
  c.R = b0(o.w[19]); c.lpg = b1(o.w[19]); c.hand_next = b2(o.w[19]); c.fwd_sel = b3(o.w[19]);
  c.fwd_coff4 = b0(o.w[20]); c.fwd_rmul = b1(o.w[20]); c.fwd_radd = b2(o.w[20]); c.cin = b3(o.w[20]);
  c.pf_phase0_ready = 0; c.pre_next_phase0 = 0;
}

__device__ __forceinline__ void decode_lstm(const OpWords& o, const StepArgs& a, LstmParams& p) {
  p.x = aptr(a.arena, o.w[0]); p.x_ld = h0(o.w[1]); p.x_cols = h1(o.w[1]); p.x_rows = 0;
  p.wxT = wptr(a.wbase, o.w[2]); p.whT = wptr(a.wbase, o.w[3]); p.bias = wptr(a.wbase, o.w[4]);
  p.wdT = wptr(a.wbase, o.w[5]); p.bd = wptr(a.wbase, o.w[6]);
  p.h_in = aptr(a.arena, o.w[7]); p.c_in = aptr(a.arena, o.w[8]); p.h_out = aptr(a.arena, o.w[9]); p.c_out = aptr(a.arena, o.w[10]);
  p.dst = aptr(a.arena, o.w[11]); p.dst_ld = h0(o.w[12]); p.dst_cols = h1(o.w[12]); p.dst_rows = 0;
  p.Din = h0(o.w[13]); p.Dout = h1(o.w[13]); p.B = a.B; p.sstride = a.sstride;
}

__device__ __forceinline__ void decode_ctfa(const OpWords& o, const StepArgs& a, CtfaParams& p) {
  p.x = aptr(a.arena, o.w[0]); p.e0 = aptr(a.arena, o.w[1]); p.y = aptr(a.arena, o.w[2]);
  p.x_ld = b0(o.w[3]); p.e0_ld = b1(o.w[3]); p.y_ld = b2(o.w[3]);
  p.ta_w1T = wptr(a.wbase, o.w[4]); p.ta_b1 = wptr(a.wbase, o.w[5]); p.ta_w2T = wptr(a.wbase, o.w[6]); p.ta_b2 = wptr(a.wbase, o.w[7]);
  p.fa_w1T = wptr(a.wbase, o.w[8]); p.fa_b1 = wptr(a.wbase, o.w[9]); p.fa_w2T = wptr(a.wbase, o.w[10]); p.fa_b2 = wptr(a.wbase, o.w[11]);
  p.F = static_cast<int>(o.w[12]); p.B = a.B; p.sstride = a.sstride;
}

// LDS float offset of (image row lr, float4 column c4) inside one phase
__device__ __forceinline__ int img_addr(const ConvPlan& c, int lr, int c4) {
  return c.stride == 1 ? lr * c.pitch + 4 * c4 : (lr >> 1) * c.pitch + (lr & 1) * c.cc + 4 * c4;
}

// ---------------------------------------------------------------------------------------------
//  Staging of a layer's LDS image through registers.  Item q (float4) -> (phase, row, column):
//  phase = (time tap t, channel chunk ch); only rows that exist in the stream travel through
//  registers, halo rows are zero-filled (ZeroPadding2D of proposed.py:210/:242, SAME pad of :255).
//  float4 columns [fwd_lo4, fwd_hi4) of the CURRENT-frame tap are excluded: the producing
//  layer's epilogue forwards them.
// ---------------------------------------------------------------------------------------------
// Item q = tid + 512*i (+ first phase) has disjoint bit fields (float4 column | row | channel chunk |
// time tap), all sizes powers of two, and tid < 512 <= the step of i: the global offset and the LDS
// address of an item are (per-lane part, computed once) + (per-pass part, wave-uniform -> SALU).
struct FwdWin { int lo4, hi4, rmul, radd; bool on; };   // float4 columns [lo4,hi4) of rows (row % rmul == radd) are forwarded
struct ItemBits { int ph, row, c4, t, ch; };
__device__ __forceinline__ ItemBits item_bits(const ConvPlan& c, int q) {
  ItemBits a;
  a.ph = q >> c.n4p_shift;
  const int r = q & ((1 << c.n4p_shift) - 1);
  a.row = r >> c.cc4_shift;
  a.c4 = r & ((1 << c.cc4_shift) - 1);
  a.t = a.ph >> c.nch_shift;
  a.ch = a.ph & ((1 << c.nch_shift) - 1);
  return a;
}

// what image_load needs of a layer; built with field-wise (scalar) selects where one load site serves two layers
struct ImgSrc { const float* src0; const float* src1; long long sstride; int ld, n4p_shift, cc4_shift, nch_shift; };
__device__ __forceinline__ ImgSrc img_src(const ConvParams& p, const ConvPlan& c) {
  return ImgSrc{p.src0, p.src1, p.sstride, p.src_ld, c.n4p_shift, c.cc4_shift, c.nch_shift};
}
__device__ __forceinline__ ItemBits item_bits(const ImgSrc& c, int q) {
  ItemBits a;
  a.ph = q >> c.n4p_shift;
  const int r = q & ((1 << c.n4p_shift) - 1);
  a.row = r >> c.cc4_shift;
  a.c4 = r & ((1 << c.cc4_shift) - 1);
  a.t = a.ph >> c.nch_shift;
  a.ch = a.ph & ((1 << c.nch_shift) - 1);
  return a;
}

// phases [ph0, ph0+nphases) of a layer image -> registers
__device__ __forceinline__ void image_load(const ImgSrc& c, int stream, int ph0, int nphases, int tid, f32x4 (&pf)[MK_MAXPF]) {
  // (opaque copy: pins the per-lane address arithmetic to this call site -- hoisted out of the
  //  caller's loops and branches it would run, for all MK_MAXPF items, in layers that never stage)
  asm volatile("" : "+v"(tid));
  // both time taps live in the same stream slice: one uniform base, the tap picks a 32-bit offset
  const float* lo = (c.src1 && c.src1 < c.src0) ? c.src1 : c.src0;
  const float* s0 = lo + static_cast<size_t>(stream) * c.sstride;
  const unsigned tap0 = static_cast<unsigned>(c.src0 - lo);
  const unsigned dtap = (c.src1 ? static_cast<unsigned>(c.src1 - lo) : tap0) - tap0;     // (mod 2^32)
  const unsigned ld = static_cast<unsigned>(c.ld);
  const int n = nphases << c.n4p_shift;
  const int wave_q0 = __builtin_amdgcn_readfirstlane(tid) & ~63;
  // lanes past the end (only when n < 512) re-load the last item: nothing selects on a pending load
  const ItemBits l = item_bits(c, tid < n ? tid : n - 1);
  const unsigned g_lo = tap0 + static_cast<unsigned>(l.t) * dtap + static_cast<unsigned>(l.row) * ld +
                        4u * static_cast<unsigned>((l.ch << c.cc4_shift) + l.c4);
#pragma unroll
  for (int i = 0; i < MK_MAXPF; ++i) {
    if (wave_q0 + i * MK_THREADS < n) {       // wave-uniform guard (scalar branch)
      const ItemBits h = item_bits(c, i * MK_THREADS + (ph0 << c.n4p_shift));           // wave-uniform
      unsigned g_hi = static_cast<unsigned>(h.t) * dtap + static_cast<unsigned>(h.row) * ld +
                      4u * static_cast<unsigned>(h.ch << c.cc4_shift);
      asm volatile("" : "+s"(g_hi));          // stays one SGPR: no re-association with the lane part
      pf[i] = ld4(s0, g_lo + g_hi);
    }
  }
}

// registers -> LDS image (phase ph lands at lds_in + (ph - ph_base) * phase_floats) + zero halo rows
__device__ __forceinline__ void image_store(const ConvPlan& c, float* lds_in, int ph0, int nphases, int ph_base, const FwdWin& fw,
                                            int tid, const f32x4 (&pf)[MK_MAXPF]) {
  asm volatile("" : "+v"(tid));
  const int n = nphases << c.n4p_shift;
  const int wave_q0 = __builtin_amdgcn_readfirstlane(tid) & ~63;
  const ItemBits l = item_bits(c, tid < n ? tid : n - 1);
  const int l_lo = l.ph * c.phase_floats + img_addr(c, c.padl + l.row, l.c4);
  const int chan4_lo = (l.ch << c.cc4_shift) + l.c4;
  const bool fwd_lane = fw.on && (l.row & (fw.rmul - 1)) == fw.radd;
#pragma unroll
  for (int i = 0; i < MK_MAXPF; ++i) {
    if (wave_q0 + i * MK_THREADS < n) {
      const ItemBits h = item_bits(c, i * MK_THREADS + (ph0 << c.n4p_shift));           // wave-uniform
      // (h.row is 0 or a multiple of 512 / cc4 >= 32: even, so it moves whole stride-2 row pairs)
      int l_hi = (h.ph - ph_base) * c.phase_floats + (c.stride == 1 ? h.row : (h.row >> 1)) * c.pitch;
      asm volatile("" : "+s"(l_hi));
      // forwarded by the producing layer's epilogue: current-frame tap, float4 columns [lo4,hi4), matching rows
      const int t = l.t | h.t;
      const unsigned col = static_cast<unsigned>(chan4_lo + (h.ch << c.cc4_shift) - fw.lo4);
      const bool skip = fwd_lane && t == c.tt - 1 && col < static_cast<unsigned>(fw.hi4 - fw.lo4);
      if ((i > 0 || tid < n) && !skip) *reinterpret_cast<f32x4*>(lds_in + l_lo + l_hi) = pf[i];
    }
  }
  // halo rows (<= 3 per phase, 4 slots): thread -> (phase, slot, float4 column), no division
  const int hrows = c.rows - c.vrows;
  const int c4 = tid & ((1 << c.cc4_shift) - 1);
  const int slot = tid >> c.cc4_shift;
  const int hr = slot & 3, phl = slot >> 2;
  if (hr < hrows && phl < nphases) {
    const int lr = hr < c.padl ? hr : c.vrows + hr;         // top halo rows first, then the bottom ones
    // (a zero the compiler cannot hoist out of the layer loop -- it would spill it, and a scratch
    //  reload costs a vmcnt(0) wait right here)
    float zf = 0.f;
    asm volatile("" : "+v"(zf));
    const f32x4 z = {zf, zf, zf, zf};
    *reinterpret_cast<f32x4*>(lds_in + (ph0 - ph_base + phl) * c.phase_floats + img_addr(c, lr, c4)) = z;
  }
}

// Direct writes into the NEXT layer's LDS image (current-frame tap): a float4 column / a single channel of row `row`
__device__ __forceinline__ void img_put4(const ConvPlan& n, float* lds_in, int row, int chan4, f32x4 v) {
  const int ph = ((n.tt - 1) << n.nch_shift) + (chan4 >> n.cc4_shift);
  *reinterpret_cast<f32x4*>(lds_in + ph * n.phase_floats + img_addr(n, n.padl + row, chan4 & ((1 << n.cc4_shift) - 1))) = v;
}
__device__ __forceinline__ void img_put1(const ConvPlan& n, float* lds_in, int row, int chan, float v) {
  const int chan4 = chan >> 2;
  const int ph = ((n.tt - 1) << n.nch_shift) + (chan4 >> n.cc4_shift);
  lds_in[ph * n.phase_floats + img_addr(n, n.padl + row, chan4 & ((1 << n.cc4_shift) - 1)) + (chan & 3)] = v;
}

// ---------------------------------------------------------------------------------------------
//  MFMA part of one round of one task.  4-group chunks: the next chunk's 4 weight fragments
//  (1 KiB per wave load, L2) are in flight while the current chunk's 16 MFMAs issue.
// ---------------------------------------------------------------------------------------------
struct WeightCursor {   // wave-uniform
  gc4_t base;           // packed weights + (k-slice, channel tile) offset of this task, round 0
  int wstep;            // float4 between consecutive groups (NT * 64)
  int round_step;       // float4 between consecutive rounds of this task (RG * wstep)
};

__device__ __forceinline__ void load_chunk(f32x4 (&w)[4], gc4_t ptr, int wstep, int lane) {
  // ptr is wave-uniform; the per-lane part is a 32-bit byte offset
#pragma unroll
  for (int u = 0; u < 4; ++u)
    w[u] = *(gc4_t)((gcb_t)ptr + static_cast<unsigned long long>(static_cast<unsigned>(u * wstep + lane) * 16u));
}

// first weight chunk of the next conv layer for this wave's task (single static load site per op kind)
__device__ __forceinline__ void load_next_weights(const ConvParams& np, const ConvPlan& ncp, int wave, int lane, f32x4 (&wnext)[4]) {
  const bool nact = wave < ncp.tasks * ncp.KS;
  const int nks = nact ? (wave >> ncp.tasks_shift) : 0;
  const int nnt = nact ? (wave & (ncp.nt - 1)) : 0;
  load_chunk(wnext, (gc4_t)(unsigned long long)np.wpk + (static_cast<size_t>(nks * ncp.gpk) * ncp.nt + nnt) * 64, ncp.nt * 64, lane);
}

// cursor over the resident image for one task: (local phase, frequency tap, channel group), wave-uniform
struct KCursor { int phl, kf, gg; };

__device__ __forceinline__ KCursor kcursor_init(const ConvPlan& c, int ks) {
  // (no integer divisions: gpc is 4 or 8, kf is 1, 2 or 3, seg < 64)
  const int g_first = ks * c.gpk;
  const int gshift = c.gpc == 8 ? 3 : 2;
  const int seg = g_first >> gshift;
  KCursor k;
  k.gg = g_first & (c.gpc - 1);
  k.phl = c.kf == 3 ? ((seg * 43) >> 7) : (c.kf == 2 ? (seg >> 1) : seg);
  k.kf = seg - k.phl * c.kf;
  return k;
}

// The 16 (x TW) MFMAs of one 4-group chunk (TW = 1 or 2 position tiles sharing the weight fragments).
template <int TW>
__device__ __forceinline__ void chunk_mfma(f32x16 (&acc)[2], const f32x4 (&wa)[4], const float* lds_lane0, const float* lds_lane1, int boff) {
  // B fragments: one ds_read_b128 per tile per group, fetched one group ahead of its MFMAs
  f32x4 b0 = *reinterpret_cast<const f32x4*>(lds_lane0 + boff), b1 = b0;
  if (TW == 2) b1 = *reinterpret_cast<const f32x4*>(lds_lane1 + boff);
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const f32x4 c0 = b0, c1 = b1;
    if (u < 3) {
      b0 = *reinterpret_cast<const f32x4*>(lds_lane0 + boff + 8 * (u + 1));
      if (TW == 2) b1 = *reinterpret_cast<const f32x4*>(lds_lane1 + boff + 8 * (u + 1));
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[u][j], c0[j], acc[0], 0, 0, 0);
      if (TW == 2) acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[u][j], c1[j], acc[1], 0, 0, 0);
    }
  }
}

// ---------------------------------------------------------------------------------------------
//  Row-wise epilogue: LPG lanes per output row (row = LPG*4 channels), optional LN + PReLU; writes
//  HBM destinations and, when `nx` is given, forwards the rows into the next layer's LDS image.
// ---------------------------------------------------------------------------------------------
template <int LPG, bool LN>
__device__ __forceinline__ void conv_epilogue(const ConvParams& p, const ConvPlan& c, int stream, const float* lds_out, int tid,
                                              f32x4 bias, f32x4 gm, f32x4 bt, bool do_fwd, const ConvPlan& nx, float* lds_next,
                                              int fwd_coff4) {
  constexpr int GC = LPG * 4;
  const int li = tid & (LPG - 1);
  const int units = p.F_out * c.R;
  float* d0 = p.dst0 + static_cast<size_t>(stream) * p.sstride;
  float* d1 = p.dst1 ? p.dst1 + static_cast<size_t>(stream) * p.sstride : nullptr;
  // forwarded block inside the next image: float4 column (coff4 + li) of the current-frame phase
  int f_ph = 0, f_c4 = 0;
  if (do_fwd) {
    const int chan4 = fwd_coff4 + li;
    f_ph = ((nx.tt - 1) << nx.nch_shift) + (chan4 >> nx.cc4_shift);
    f_c4 = chan4 & ((1 << nx.cc4_shift) - 1);
  }
  for (int u = tid / LPG; u < units; u += MK_THREADS / LPG) {
    const int pos = (c.R == 2) ? (u >> 1) : u, gi = (c.R == 2) ? (u & 1) : 0;
    const float* o = lds_out + pos * c.opitch + gi * GC + 4 * li;
    f32x4 v = bias;
    for (int ks = 0; ks < c.KS; ++ks) v += *reinterpret_cast<const f32x4*>(o + ks * c.slot_floats);
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
    const int row = pos * p.row_mul + p.row_add + gi;
    st4(d0, static_cast<unsigned>(row * p.ld0 + 4 * li), v);
    if (d1) st4(d1, static_cast<unsigned>(row * p.ld1 + 4 * li), v);
    if (do_fwd) *reinterpret_cast<f32x4*>(lds_next + f_ph * nx.phase_floats + img_addr(nx, nx.padl + row, f_c4)) = v;
  }
}

// One conv-like layer for one stream.
//   wnext  in : this task's first weight chunk (fetched while the previous layer ran) when `have_w`
//          out: the next layer's first chunk (single static load site -> no copies of pending loads)
__device__ __forceinline__ void conv_layer(const ConvParams& p, const ConvPlan& c, bool nconv, const ConvParams& np, const ConvPlan& ncp,
                                           int stream, float* lds_in, float* lds_out, int tid, f32x4 (&wnext)[4], bool& have_w,
                                           unsigned long long* sub, unsigned long long* dbg) {
  MK_T(0);
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // wave-uniform -> SGPR
  const int pl = lane & 31, h = lane >> 5;
  const bool active = wave < c.tasks * c.KS;
  const int ks = wave >> c.tasks_shift, tl = wave & (c.tasks - 1);
  const int pt = (tl >> c.nt_shift) * c.tw, nt = tl & (c.nt - 1);     // first position tile, channel tile
  int pc0 = pt * 32 + pl, pc1 = pc0 + 32;
  if (pc0 > p.F_out - 1) pc0 = p.F_out - 1;        // padding lanes recompute the last position
  if (pc1 > p.F_out - 1) pc1 = p.F_out - 1;
  const float* lds_lane0 = lds_in + pc0 * c.pitch + 4 * h;
  const float* lds_lane1 = lds_in + pc1 * c.pitch + 4 * h;
  const FwdWin nofw = {0, 0, 1, 0, false};

  WeightCursor wc;
  wc.wstep = c.nt * 64;
  wc.round_step = c.RG * wc.wstep;
  wc.base = (gc4_t)(unsigned long long)p.wpk + (static_cast<size_t>(active ? ks * c.gpk : 0) * c.nt + (active ? nt : 0)) * 64;

  // ---- first weight chunk: fetched by the previous layer, or (first layer of a run) right now
  f32x4 wa[4];
  if (have_w) {
#pragma unroll
    for (int u = 0; u < 4; ++u) wa[u] = wnext[u];
  } else {
    load_chunk(wa, wc.base, wc.wstep, lane);
  }
  MK_STAMP(0);

  // ---- LDS image of round 0 (skipped when the previous layer handed it over complete)
  const int nstage = c.merged ? c.nph : 1;
  if (!c.staged_by_prev) {
    f32x4 pf[MK_MAXPF];
    image_load(img_src(p, c), stream, 0, nstage, tid, pf);
    image_store(c, lds_in, 0, nstage, 0, nofw, tid, pf);
    lds_barrier();
  }
  MK_STAMP(1);
  MK_T(2);

  // ---- what this layer owes the next one
  const bool hand = nconv && c.hand_next;
  FwdWin fw = nofw;
  if (hand && c.fwd_sel) { fw.lo4 = c.fwd_coff4; fw.hi4 = c.fwd_coff4 + c.lpg; fw.rmul = c.fwd_rmul; fw.radd = c.fwd_radd; fw.on = true; }
  const int n_hand = hand ? (ncp.merged ? ncp.nph : 1) : 0;

  // Global loads are issued oldest-needed-first (vmcnt retires in order): epilogue parameters, then
  // the next layer's first weight chunk, then -- after this layer's own last weight prefetch -- the
  // rows of the next layer's LDS image that this layer does not produce (previous-frame tap,
  // skip-connection channels; HBM, long latency).  None of the barriers in between drains vmcnt.
  const int lpg = c.lpg;
  const int li = tid & (lpg - 1);
  const int gi_mine = (c.R == 2) ? ((tid / lpg) & 1) : 0;
  const f32x4 bias = *G4(p.bias + gi_mine * (4 * lpg) + 4 * li);
  // (always loaded -- a value select on a pending load would wait for it right here; layers without
  //  LayerNorm re-read the bias)
  const f32x4 gm = *G4((c.epi_ln ? p.gamma : p.bias) + 4 * li);
  const f32x4 bt = *G4((c.epi_ln ? p.beta : p.bias) + 4 * li);
  MK_STAMP(6);
  MK_T(3);
  f32x16 acc[2];
#pragma unroll
  for (int r = 0; r < 16; ++r) { acc[0][r] = 0.f; acc[1][r] = 0.f; }
  // One register set and ONE static load site serve both image prefetches -- the next phase of THIS
  // layer (rounds before the last: issued with the round's first chunk) and, in the last round, the
  // next layer's image rows (issued right after the layer's last weight prefetch: the youngest loads,
  // vmcnt retires in order).  Two load sites would meet in phi copies of pending loads, and the copy
  // costs a full vmcnt(0) wait.
  f32x4 pfx[MK_MAXPF];
  const int nchunks = active ? (c.gpk >> 2) : 1;     // idle waves run one empty chunk per round: they stage too
#pragma unroll 1
  for (int rd = 0; rd < c.rounds; ++rd) {
    const bool last = rd + 1 == c.rounds;
    const int hook_ch = last ? nchunks - 1 : 0;
    KCursor k = kcursor_init(c, ks);
    gc4_t wp = wc.base + static_cast<size_t>(rd) * wc.round_step;
#pragma unroll 1
    for (int ch = 0; ch < nchunks; ++ch) {
      // prefetch the next chunk: next in this slice, else the first chunk of the next round's slice
      // (wave-uniform condition -> scalar branch; nothing is fetched after the layer's last chunk)
      const bool more = active && ((ch + 1 < nchunks) || !last);
      f32x4 wb[4];
      if (more) {
        gc4_t nxt = (ch + 1 < nchunks) ? wp + 4 * wc.wstep : wc.base + static_cast<size_t>(rd + 1) * wc.round_step;
        load_chunk(wb, nxt, wc.wstep, lane);
      }
      if (ch == hook_ch && (!last || hand)) {
        ImgSrc is;
        is.src0 = last ? np.src0 : p.src0; is.src1 = last ? np.src1 : p.src1; is.sstride = p.sstride;
        is.ld = last ? np.src_ld : p.src_ld;
        is.n4p_shift = last ? ncp.n4p_shift : c.n4p_shift; is.cc4_shift = last ? ncp.cc4_shift : c.cc4_shift;
        is.nch_shift = last ? ncp.nch_shift : c.nch_shift;
        image_load(is, stream, last ? 0 : rd + 1, last ? n_hand : 1, tid, pfx);
        MK_STAMP(7);
        MK_T(4);
      }
      if (active) {
        const int koff = (c.stride == 1) ? k.kf * c.pitch : ((k.kf >> 1) * c.pitch + (k.kf & 1) * c.cc);
        const int boff = k.phl * c.phase_floats + koff + 8 * k.gg;
        if (c.tw == 2) chunk_mfma<2>(acc, wa, lds_lane0, lds_lane1, boff);
        else chunk_mfma<1>(acc, wa, lds_lane0, lds_lane1, boff);
      }
      if (more) {
#pragma unroll
        for (int u = 0; u < 4; ++u) wa[u] = wb[u];
      }
      wp += 4 * wc.wstep;
      k.gg += 4;
      if (k.gg == c.gpc) { k.gg = 0; if (++k.kf == c.kf) { k.kf = 0; ++k.phl; } }
    }
    if (!last) {
      lds_barrier();             // every wave is done reading this phase's LDS rows
      image_store(c, lds_in, rd + 1, 1, rd + 1, nofw, tid, pfx);
      lds_barrier();
    }
  }
  MK_T(5);
  {   // next layer's first weight chunk (single load site; a layer with no conv successor re-reads its own)
    // (field-wise selects: a reference/pointer select between the two structs would force both to memory)
    const float* nw = nconv ? np.wpk : p.wpk;
    const int n_tasks = nconv ? ncp.tasks : c.tasks, n_ks = nconv ? ncp.KS : c.KS, n_shift = nconv ? ncp.tasks_shift : c.tasks_shift;
    const int n_nt = nconv ? ncp.nt : c.nt, n_gpk = nconv ? ncp.gpk : c.gpk;
    const bool nact = wave < n_tasks * n_ks;
    const int nks = nact ? (wave >> n_shift) : 0;
    const int nnt = nact ? (wave & (n_nt - 1)) : 0;
    load_chunk(wnext, (gc4_t)(unsigned long long)nw + (static_cast<size_t>(nks * n_gpk) * n_nt + nnt) * 64, n_nt * 64, lane);
    have_w = nconv;
  }
  MK_STAMP(2);
  MK_T(6);

  // ---- partial tiles -> LDS exchange buffer [ks][pos][32*NT (+4)]
  if (active) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      if (t < c.tw) {
        float* o = lds_out + ks * c.slot_floats + ((pt + t) * 32 + pl) * c.opitch + nt * 32 + 4 * h;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 v = {acc[t][4 * q], acc[t][4 * q + 1], acc[t][4 * q + 2], acc[t][4 * q + 3]};
          *reinterpret_cast<f32x4*>(o + 8 * q) = v;
        }
      }
    }
  }
  MK_T(7);
  lds_barrier();                 // also: every wave has finished reading lds_in
  MK_STAMP(3);
  MK_T(8);

  const bool do_fwd = hand && c.fwd_sel;
  if (c.epi_ln) {
    if (c.g == 1) conv_epilogue<8, true>(p, c, stream, lds_out, tid, bias, gm, bt, do_fwd, ncp, lds_in, c.fwd_coff4);
    else conv_epilogue<16, true>(p, c, stream, lds_out, tid, bias, gm, bt, do_fwd, ncp, lds_in, c.fwd_coff4);
  } else {
    if (c.g == 2) conv_epilogue<16, false>(p, c, stream, lds_out, tid, bias, gm, bt, do_fwd, ncp, lds_in, c.fwd_coff4);
    else conv_epilogue<32, false>(p, c, stream, lds_out, tid, bias, gm, bt, do_fwd, ncp, lds_in, c.fwd_coff4);
  }
  MK_T(9);
  // ---- hand-off: the prefetched part of the next layer's image (the forwarded rows were written above)
  if (hand) image_store(ncp, lds_in, 0, n_hand, 0, fw, tid, pfx);
  MK_STAMP(4);
  MK_T(10);
  __syncthreads();               // HBM stores visible to the workgroup; next image complete; exchange buffer free
  MK_STAMP(5);
  MK_T(11);
}

__device__ __forceinline__ float mk_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
  return v;
}

// LSTM cell + Dense for one stream (models/proposed.py:70-119; converter_proposed.py:234-237).
// Thread (n4, sl): gate outputs 4*n4..4*n4+3 (84 = 21 float4) x K slice sl of 16; every global load of
// the op (input, weights of all three products, states, and the next conv layer's image rows) is issued
// before the first barrier, so one memory latency is exposed.  When `hand`, the op also completes the
// next conv layer's LDS image: its own output is written straight into it.
__device__ __forceinline__ void lstm_layer(const LstmParams& p, int stream, float* lds, float* lds_in, int tid, bool hand, int fwd_coff,
                                           const ConvParams& np, const ConvPlan& ncp) {
  float* part = lds;               // [16][84]
  float* z = lds + 16 * 84;        // [96]
  float* hn = z + 96;              // [32]
  const float* sb = p.x + static_cast<size_t>(stream) * p.sstride;         // stream slice of the input tensor
  const int n4 = tid % 21, sl = tid / 21;                                    // sl < 16 for tid < 336
  const int kn = p.Din >> 4, k0 = sl * kn;                                   // kn in {2, 4, 8, 16}
  const bool mv = tid < 336;
  // ---- x . Wx: weights + inputs of this thread's K slice
  f32x4 w4[16];
  float xv[16];
  if (mv) {
#pragma unroll
    for (int j = 0; j < 16; ++j)
      if (j < kn) {
        const int k = k0 + j;
        w4[j] = ld4(p.wxT, static_cast<unsigned>(k * 84 + 4 * n4));
        const int f = k / p.x_cols, c = k - f * p.x_cols;
        xv[j] = ld1(sb, static_cast<unsigned>(f * p.x_ld + c));
      }
  }
  // ---- recurrent product, bias, cell state, dense weights (threads < 84 / < 21 / < Dout)
  float wh[21], hprev[21];
  float bias = 0.f, c_old = 0.f;
  if (tid < 84) {
#pragma unroll
    for (int u = 0; u < 21; ++u) {
      wh[u] = ld1(p.whT, static_cast<unsigned>(u * 84 + tid));
      hprev[u] = ld1(p.h_in + static_cast<size_t>(stream) * p.sstride, static_cast<unsigned>(u));
    }
    bias = ld1(p.bias, static_cast<unsigned>(tid));
  }
  if (tid < 21) c_old = ld1(p.c_in + static_cast<size_t>(stream) * p.sstride, static_cast<unsigned>(tid));
  float wd[21];
  float bd = 0.f;
  if (tid < p.Dout) {
#pragma unroll
    for (int u = 0; u < 21; ++u) wd[u] = ld1(p.wdT, static_cast<unsigned>(u * p.Dout + tid));
    bd = ld1(p.bd, static_cast<unsigned>(tid));
  }
  // ---- next conv layer's image rows this op does not produce
  f32x4 pfx[MK_MAXPF];
  if (hand) image_load(img_src(np, ncp), stream, 0, ncp.nph, tid, pfx);

  if (mv) {
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 16; ++j)
      if (j < kn) a += w4[j] * xv[j];
    *reinterpret_cast<f32x4*>(part + sl * 84 + 4 * n4) = a;
  }
  lds_barrier();
  if (tid < 84) {
    float a = bias;
#pragma unroll
    for (int s2 = 0; s2 < 16; ++s2) a += part[s2 * 84 + tid];
    float r = 0.f;
#pragma unroll
    for (int u = 0; u < 21; ++u) r = fmaf(wh[u], hprev[u], r);
    z[tid] = a + r;
  }
  lds_barrier();
  if (tid < 21) {
    const float gi = mk_sigmoid(z[tid]), gf = mk_sigmoid(z[21 + tid]);
    const float gg = tanhf(z[42 + tid]), go = mk_sigmoid(z[63 + tid]);
    const float c_new = gf * c_old + gi * gg;
    const float h_new = go * tanhf(c_new);
    st1(p.c_out + static_cast<size_t>(stream) * p.sstride, static_cast<unsigned>(tid), c_new);
    st1(p.h_out + static_cast<size_t>(stream) * p.sstride, static_cast<unsigned>(tid), h_new);
    hn[tid] = h_new;
  }
  lds_barrier();
  if (tid < p.Dout) {
    float a = bd;
#pragma unroll
    for (int u = 0; u < 21; ++u) a = fmaf(wd[u], hn[u], a);
    const int f = tid / p.dst_cols, c = tid - f * p.dst_cols;
    st1(p.dst + static_cast<size_t>(stream) * p.sstride, static_cast<unsigned>(f * p.dst_ld + c), a);
    if (hand) img_put1(ncp, lds_in, f, fwd_coff + c, a);
  }
  if (hand) {
    const FwdWin fw = {fwd_coff >> 2, (fwd_coff + p.dst_cols) >> 2, 1, 0, true};
    image_store(ncp, lds_in, 0, ncp.nph, 0, fw, tid, pfx);
  }
  __syncthreads();
}

// CTFA gate + residual for one stream (ctfa_rt, models/proposed.py:162-196).
// Wave u computes hidden unit u of the 64->16 layers (one product per lane + wave reduction);
// the MLP weights are fetched before the mean-over-F reduction so their latency is hidden.
__device__ __forceinline__ void ctfa_layer(const CtfaParams& p, int stream, float* lds, float* lds_in, int tid, bool hand, int fwd_coff,
                                           const ConvParams& np, const ConvPlan& ncp) {
  f32x4 pfx[MK_MAXPF];
  if (hand) image_load(img_src(np, ncp), stream, 0, ncp.nph, tid, pfx);
  float* part = lds;               // [32][64]
  float* m = lds + 4096;           // [64]
  float* hid = m + 64;             // [16]
  float* ta = hid + 16;            // [64]
  float* gate = ta + 64;           // [64]
  const int c4 = tid & 15, rg = tid >> 4;            // 32 row groups x 16 float4 columns
  const int lane = tid & 63, wave = tid >> 6;        // wave w owns hidden units w and w + 8
  const float w1_ta0 = GF(p.ta_w1T)[lane * 16 + wave], w1_ta1 = GF(p.ta_w1T)[lane * 16 + wave + 8];
  const float w1_fa0 = GF(p.fa_w1T)[lane * 16 + wave], w1_fa1 = GF(p.fa_w1T)[lane * 16 + wave + 8];
  const float b1_ta0 = GF(p.ta_b1)[wave], b1_ta1 = GF(p.ta_b1)[wave + 8];
  const float b1_fa0 = GF(p.fa_b1)[wave], b1_fa1 = GF(p.fa_b1)[wave + 8];
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
  const float* xb = p.x + static_cast<size_t>(stream) * p.sstride;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  for (int f = rg; f < p.F; f += 32) s += *G4(xb + static_cast<size_t>(f) * p.x_ld + 4 * c4);
  *reinterpret_cast<f32x4*>(part + rg * 64 + 4 * c4) = s;
  lds_barrier();
  if (tid < 64) {
    float a = 0.f;
#pragma unroll 16
    for (int r = 0; r < 32; ++r) a += part[r * 64 + tid];
    m[tid] = a / static_cast<float>(p.F);
  }
  lds_barrier();
  {
    const float t0 = wave_sum(w1_ta0 * m[lane]), t1 = wave_sum(w1_ta1 * m[lane]);
    if (lane == 0) { hid[wave] = fmaxf(t0 + b1_ta0, 0.f); hid[wave + 8] = fmaxf(t1 + b1_ta1, 0.f); }
  }
  lds_barrier();
  float ta_c = 0.f;
  if (tid < 64) {
    float a = b2_ta;
#pragma unroll
    for (int u = 0; u < 16; ++u) a = fmaf(w2_ta[u], hid[u], a);
    ta_c = mk_sigmoid(a);
    ta[tid] = ta_c;
  }
  lds_barrier();
  {
    const float tv = ta[lane] * (1.0f / 32.0f);
    const float t0 = wave_sum(w1_fa0 * tv), t1 = wave_sum(w1_fa1 * tv);
    lds_barrier();             // everyone has read hid (TA pass) before it is overwritten
    if (lane == 0) { hid[wave] = fmaxf(t0 + b1_fa0, 0.f); hid[wave + 8] = fmaxf(t1 + b1_fa1, 0.f); }
  }
  lds_barrier();
  if (tid < 64) {
    float a = b2_fa;
#pragma unroll
    for (int u = 0; u < 16; ++u) a = fmaf(w2_fa[u], hid[u], a);
    gate[tid] = mk_sigmoid(a) * ta_c;
  }
  lds_barrier();
  const f32x4 g4 = *reinterpret_cast<const f32x4*>(gate + 4 * c4);
  const float* eb = p.e0 + static_cast<size_t>(stream) * p.sstride;
  float* yb = p.y + static_cast<size_t>(stream) * p.sstride;
  for (int f = rg; f < p.F; f += 32) {
    const f32x4 xv = *G4(xb + static_cast<size_t>(f) * p.x_ld + 4 * c4);
    const f32x4 ev = *G4(eb + static_cast<size_t>(f) * p.e0_ld + 4 * c4);
    const f32x4 yv = xv * g4 + ev;
    *G4W(yb + static_cast<size_t>(f) * p.y_ld + 4 * c4) = yv;
    if (hand) img_put4(ncp, lds_in, f, (fwd_coff >> 2) + c4, yv);
  }
  if (hand) {
    const FwdWin fw = {fwd_coff >> 2, (fwd_coff >> 2) + 16, 1, 0, true};
    image_store(ncp, lds_in, 0, ncp.nph, 0, fw, tid, pfx);
  }
  __syncthreads();
}

__device__ __forceinline__ void input_layer_op(const OpWords& o, const StepArgs& a, int stream, float* lds_in, int tid, bool hand,
                                               const ConvPlan& ncp) {
  InLayerParams p;
  p.x = a.io_in; p.y = aptr(a.arena, o.w[0]);
  p.w = wptr(a.wbase, o.w[1]); p.b = wptr(a.wbase, o.w[2]); p.gamma = wptr(a.wbase, o.w[3]); p.beta = wptr(a.wbase, o.w[4]);
  p.alpha = __uint_as_float(o.w[5]); p.sstride = a.sstride;
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
    *G4W(p.y + static_cast<size_t>(stream) * p.sstride + static_cast<size_t>(pos) * 64 + 4 * c4) = o4;
    if (hand) img_put4(ncp, lds_in, pos, c4, o4);     // the whole image of msfe6_en_in is this op's output
  }
  __syncthreads();
}

__device__ __forceinline__ void out_conv_op(const OpWords& o, const StepArgs& a, int stream, int tid) {
  OutConvParams p;
  p.x = aptr(a.arena, o.w[0]); p.x_ld = static_cast<int>(o.w[1]); p.y = a.io_out;
  p.w = wptr(a.wbase, o.w[2]); p.bias = __uint_as_float(o.w[3]); p.sstride = a.sstride;
  const int c4 = tid & 15;
  const f32x4 w = *G4(p.w + 4 * c4);
  for (int pos = tid >> 4; pos < NUTLS_DEV_BINS; pos += MK_THREADS / 16) {
    const f32x4 xv = *G4(p.x + static_cast<size_t>(stream) * p.sstride + static_cast<size_t>(pos) * p.x_ld + 4 * c4);
    float s = xv[0] * w[0] + xv[1] * w[1] + xv[2] * w[2] + xv[3] * w[3];
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) s += __shfl_xor(s, o);
    if (c4 == 0) GFW(p.y)[static_cast<size_t>(stream) * NUTLS_DEV_BINS + pos] = s + p.bias;
  }
  __syncthreads();
}

__global__ __launch_bounds__(MK_THREADS) void nutls_stream_step_kernel(const StepArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* lds_in = lds;
  float* lds_out = lds + MK_LDS_IN;
  unsigned* lds_plan = reinterpret_cast<unsigned*>(lds + MK_LDS_IN + MK_LDS_OUT);
  const int n_ops = a.n_ops;
  unsigned long long* prof = a.prof;
  // the whole compact plan -> LDS, once
  {
    const u32x4* src = reinterpret_cast<const u32x4*>(a.plan);
    for (int q = threadIdx.x; q < n_ops * (MK_OP_WORDS / 4); q += MK_THREADS)
      reinterpret_cast<u32x4*>(lds_plan)[q] = src[q];
    __syncthreads();
  }
  // `fresh_tid()` re-materialises the thread id behind an opaque asm at every layer: without it the
  // compiler hoists dozens of tid-derived per-thread constants out of the layer loop, keeps them live
  // across the whole kernel and spills them -- and a scratch reload is a vmcnt wait, which in this
  // kernel means "wait for every prefetch in flight".
  auto fresh_tid = []() { int t = threadIdx.x; asm volatile("" : "+v"(t)); return t; };
  for (int stream = blockIdx.x; stream < a.B; stream += gridDim.x) {
    if (prof && blockIdx.x == 0 && threadIdx.x == 0) prof[n_ops * 9 + 1] = clock64();
    // the first weight chunk of the next conv layer travels in registers from op to op
    f32x4 wnext[4];
    bool have_w = false;
    OpWords cur = load_op(lds_plan, 0);
#pragma unroll 1
    for (int i = 0; i < n_ops; ++i) {
      const int tid = fresh_tid();
      const int lane = tid & 63;
      const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
      if (prof && blockIdx.x == 0 && tid == 0) prof[i] = wall_clock64();
      const OpWords nxt = load_op(lds_plan, i + 1 < n_ops ? i + 1 : i);
      const bool nconv = (i + 1 < n_ops) && static_cast<int>(nxt.w[23]) == DEV_OP_CONV;
      ConvParams np;
      ConvPlan ncp;
      decode_conv(nxt, a, np, ncp);
      const int op = static_cast<int>(cur.w[23]);
      const bool nc_hand = nconv && b0(cur.w[22]) != 0;      // non-conv op -> conv hand-off
      const int nc_coff = b1(cur.w[22]);
      if (op == DEV_OP_CONV) {
        unsigned long long* sub = (prof && blockIdx.x == 0) ? prof + (n_ops + 1) + 8 * i : nullptr;
        unsigned long long* dbg = (sub && i == a.dbg_op) ? prof + n_ops * 9 + 3 : nullptr;
        if (dbg && (tid & 63) == 0) dbg[(tid >> 6) * 16 + 15] = clock64();
        ConvParams p;
        ConvPlan c;
        decode_conv(cur, a, p, c);
        if (dbg && (tid & 63) == 0) dbg[(tid >> 6) * 16 + 14] = clock64();
        conv_layer(p, c, nconv, np, ncp, stream, lds_in, lds_out, tid, wnext, have_w, sub, dbg);
      } else {
        if (op == DEV_OP_LSTM) {
          LstmParams p;
          decode_lstm(cur, a, p);
          lstm_layer(p, stream, lds_out, lds_in, tid, nc_hand, nc_coff, np, ncp);
        } else if (op == DEV_OP_CTFA) {
          CtfaParams p;
          decode_ctfa(cur, a, p);
          ctfa_layer(p, stream, lds_out, lds_in, tid, nc_hand, nc_coff, np, ncp);
        } else if (op == DEV_OP_INLAYER) {
          input_layer_op(cur, a, stream, lds_in, tid, nc_hand, ncp);
        } else if (op == DEV_OP_DDB) {
          ddb_block(a.ddb[cur.w[0]], stream, lds_out, tid, MK_THREADS);
        } else {
          out_conv_op(cur, a, stream, tid);
        }
        have_w = false;
        if (nconv) {
          load_next_weights(np, ncp, wave, lane, wnext);
          have_w = true;
        }
      }
      cur = nxt;
    }
    if (prof && blockIdx.x == 0 && threadIdx.x == 0) { prof[n_ops] = wall_clock64(); prof[n_ops * 9 + 2] = clock64(); }
  }
}

hipError_t launch_stream_step(const StepArgs& a, int grid, hipStream_t s) {
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(nutls_stream_step_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(MK_LDS_BYTES));
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  hipLaunchKernelGGL(nutls_stream_step_kernel, dim3(grid), dim3(MK_THREADS), MK_LDS_BYTES, s, a);
  return hipGetLastError();
}

}  // namespace nutls
