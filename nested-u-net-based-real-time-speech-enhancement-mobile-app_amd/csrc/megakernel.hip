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
constexpr int MK_LDS_DBG = 256;               // floats: 8 waves x 16 cycle stamps (profiling only)
constexpr size_t MK_LDS_BYTES = (MK_LDS_IN + MK_LDS_OUT + MK_LDS_DBG) * sizeof(float);

// Workgroup barrier that orders LDS traffic only: unlike __syncthreads() it does NOT drain vmcnt,
// so global loads issued earlier (next layer's image rows, weight fragments) stay in flight across it.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

#define MK_STAMP(k) do { if (PROF && sub && tid == 0) sub[k] = wall_clock64(); } while (0)
// fine-grained cycle stamps of ONE chosen op (debug builds of the profile path only)
// (stamps go to LDS and are copied out after the op: a global store here would sit in vmcnt and stretch
//  every vmcnt(0) wait that follows)
#define MK_T(k) do { if (PROF && dbg && (tid & 63) == 0) dbg_lds[(tid >> 6) * 16 + k] = clock64(); } while (0)

// ---------------------------------------------------------------------------------------------
//  Compact plan (layout: encode_op() in weights.cpp).  Ops are read with scalar loads straight into
//  SGPRs (constant address space, wave-uniform index) and conv fields are extracted where they are
//  used: a bit-field extract is one SALU instruction, an eagerly decoded struct is ~70 live SGPRs
//  per op and a v_readlane / v_writelane spill pair for most of them.  All conv offsets are BYTES
//  from the stream slice (`sb`) or the weight arena (`wb`): every global access is
//  "SGPR base + 32-bit VGPR offset", no 64-bit arithmetic.
// ---------------------------------------------------------------------------------------------
struct OpWords { unsigned w[MK_OP_WORDS]; };
typedef const unsigned __attribute__((address_space(4))) * cplan_t;

__device__ __forceinline__ OpWords load_op(cplan_t plan, int i) {
  OpWords o;
#pragma unroll
  for (int k = 0; k < MK_OP_WORDS; ++k) o.w[k] = plan[i * MK_OP_WORDS + k];
  return o;
}
// Same words again, through a fresh scalar load the compiler cannot merge with the first one: lets the
// first copy die before a register-hungry region instead of being spilled to VGPR lanes across it.
__device__ __forceinline__ OpWords reload_op(cplan_t plan, int i) {
  asm volatile("" : "+s"(i));
  return load_op(plan, i);
}
__device__ __forceinline__ int b0(unsigned w) { return w & 255; }
__device__ __forceinline__ int b1(unsigned w) { return (w >> 8) & 255; }
__device__ __forceinline__ int b2(unsigned w) { return (w >> 16) & 255; }
__device__ __forceinline__ int b3(unsigned w) { return w >> 24; }
__device__ __forceinline__ int h0(unsigned w) { return w & 0xFFFF; }
__device__ __forceinline__ int h1(unsigned w) { return w >> 16; }
__device__ __forceinline__ float* aptr(float* base, unsigned off) { return off == MK_NULL_OFF ? nullptr : base + off; }
__device__ __forceinline__ const float* wptr(const float* base, unsigned off) { return off == MK_NULL_OFF ? nullptr : base + off; }

// conv op fields
#define CV(name, expr) __device__ __forceinline__ int cv_##name(const OpWords& o) { return (expr); }
CV(src_ld_b, h0(o.w[9]))   CV(ld0_b, h1(o.w[9]))   CV(ld1_b, h0(o.w[10]))
CV(row_mul, (o.w[10] >> 16) & 3)   CV(row_add, (o.w[10] >> 18) & 1)   CV(stride2, (o.w[10] >> 19) & 1)   CV(tt2, (o.w[10] >> 20) & 1)
CV(epi_ln, (o.w[10] >> 21) & 1)    CV(merged, (o.w[10] >> 22) & 1)    CV(staged_by_prev, (o.w[10] >> 23) & 1)
CV(hand_next, (o.w[10] >> 24) & 1) CV(fwd_sel, (o.w[10] >> 25) & 1)   CV(fwd_rmul2, (o.w[10] >> 26) & 1)   CV(fwd_radd, (o.w[10] >> 27) & 1)
CV(R2, (o.w[10] >> 28) & 1)        CV(has_dst1, (o.w[10] >> 29) & 1)  CV(gcode, o.w[10] >> 30)
CV(F_in, h0(o.w[11]))      CV(F_out, h1(o.w[11]))
CV(kf, b0(o.w[12]))        CV(padl, b1(o.w[12]))       CV(lpg, b2(o.w[12]))        CV(nt, b3(o.w[12]))
CV(cc4_shift, b0(o.w[13])) CV(n4p_shift, b1(o.w[13]))  CV(nch_shift, b2(o.w[13]))  CV(nt_shift, b3(o.w[13]))
CV(pitch_b, h0(o.w[14]))   CV(rows, h1(o.w[14]))
CV(vrows, h0(o.w[15]))     CV(nph, b2(o.w[15]))        CV(rounds, b3(o.w[15]))
CV(phase_b, o.w[16])
CV(RG, b0(o.w[18]))        CV(KS, b1(o.w[18]))         CV(gpk, b2(o.w[18]))        CV(gpc, b3(o.w[18]))
CV(tasks, b0(o.w[19]))     CV(tasks_shift, b1(o.w[19])) CV(tw, b2(o.w[19]))        CV(fwd_coff4, b3(o.w[19]))
CV(opitch_b, h0(o.w[20]))  CV(units, h1(o.w[20]))
CV(hx_nhand, (o.w[21] >> 1) & 7)   CV(hx_cc4_shift, (o.w[21] >> 4) & 15)   CV(hx_n4p_shift, (o.w[21] >> 8) & 15)
CV(hx_nch_shift, (o.w[21] >> 12) & 15)   CV(hx_ld_b, o.w[21] >> 16)
CV(s16, o.w[21] & 1)       // F_out <= 16: 16x16x4 MFMA tiles, fragments are 16-channel K groups, nt counts 16-channel tiles
#undef CV
__device__ __forceinline__ int cv_cc_b(const OpWords& o) { return 16 << cv_cc4_shift(o); }      // bytes of one channel chunk
// bytes of one K-slice slot of the exchange buffer: PT * 32 positions (PT = 1 for 16-position tiles)
__device__ __forceinline__ int cv_slot_b(const OpWords& o) { return ((cv_F_out(o) + 31) >> 5) * 32 * cv_opitch_b(o); }

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
  p.ta_w2 = wptr(a.wbase, o.w[13]); p.fa_w2 = wptr(a.wbase, o.w[14]);
}

// global / LDS accesses by byte offset
__device__ __forceinline__ f32x4 ldb(gcb_t base, unsigned boff) { return *(gc4_t)(base + static_cast<unsigned long long>(boff)); }
__device__ __forceinline__ void stb(gcb_t base, unsigned boff, f32x4 v) {
  *(g4_t)((gb_t)(unsigned long long)base + static_cast<unsigned long long>(boff)) = v;
}
__device__ __forceinline__ f32x4& lds4(float* base, int boff) { return *reinterpret_cast<f32x4*>(reinterpret_cast<char*>(base) + boff); }
__device__ __forceinline__ float& lds1(float* base, int boff) { return *reinterpret_cast<float*>(reinterpret_cast<char*>(base) + boff); }

// LDS byte offset of (image row lr, float4 column c4) inside one phase
__device__ __forceinline__ int img_addr_b(const OpWords& o, int lr, int c4) {
  const int pitch = cv_pitch_b(o);
  return cv_stride2(o) ? (lr >> 1) * pitch + (lr & 1) * cv_cc_b(o) + 16 * c4 : lr * pitch + 16 * c4;
}

// ---------------------------------------------------------------------------------------------
//  Staging of a layer's LDS image through registers.  Item q (float4) -> (phase, row, column):
//  phase = (time tap t, channel chunk ch); only rows that exist in the stream travel through
//  registers, halo rows are zero-filled (ZeroPadding2D of proposed.py:210/:242, SAME pad of :255).
//  Item q = tid + 512*i (+ first phase) has disjoint bit fields (float4 column | row | channel chunk |
//  time tap), all sizes powers of two, and tid < 512 <= the step of i: the global offset and the LDS
//  address of an item are (per-lane part, computed once) + (per-pass part, wave-uniform -> SALU).
//  float4 columns [lo4, hi4) of the CURRENT-frame tap can be excluded from the store: the producing
//  layer's epilogue forwards them.
// ---------------------------------------------------------------------------------------------
struct FwdWin { int lo4, hi4, rmul, radd; bool on; };   // float4 columns [lo4,hi4) of rows (row % rmul == radd) are forwarded
struct ImgSrc { unsigned src0, dtap, ld; int n4p_shift, cc4_shift, nch_shift; };   // what image_load needs of a layer
__device__ __forceinline__ ImgSrc img_src(const OpWords& o) {
  return ImgSrc{o.w[0], o.w[1], static_cast<unsigned>(cv_src_ld_b(o)), cv_n4p_shift(o), cv_cc4_shift(o), cv_nch_shift(o)};
}
struct ItemBits { int ph, row, c4, t, ch; };
__device__ __forceinline__ ItemBits item_bits(int n4p_shift, int cc4_shift, int nch_shift, int q) {
  ItemBits a;
  a.ph = q >> n4p_shift;
  const int r = q & ((1 << n4p_shift) - 1);
  a.row = r >> cc4_shift;
  a.c4 = r & ((1 << cc4_shift) - 1);
  a.t = a.ph >> nch_shift;
  a.ch = a.ph & ((1 << nch_shift) - 1);
  return a;
}

// phases [ph0, ph0+nphases) of a layer image -> registers.  Pass 0 is unconditional (every image has
// one); passes 1..7 exist only in images of more than 512 float4 and their (scalar) address parts are
// computed behind the guards.
__device__ __forceinline__ void image_load(const ImgSrc& c, gcb_t sb, int ph0, int nphases, int tid, f32x4 (&pf)[MK_MAXPF]) {
  // (opaque copy: pins the per-lane address arithmetic to this call site -- hoisted out of the
  //  caller's loops and branches it would run in layers that never stage)
  asm volatile("" : "+v"(tid));
  const int n = nphases << c.n4p_shift;
  const int wave_q0 = __builtin_amdgcn_readfirstlane(tid) & ~63;
  // lanes past the end (only when n < 512) re-load the last item: nothing selects on a pending load
  const ItemBits l = item_bits(c.n4p_shift, c.cc4_shift, c.nch_shift, tid < n ? tid : n - 1);
  const unsigned g_lo = c.src0 + static_cast<unsigned>(l.t) * c.dtap + static_cast<unsigned>(l.row) * c.ld +
                        16u * static_cast<unsigned>((l.ch << c.cc4_shift) + l.c4);
  auto pass = [&](int i) {
    int qh = i * MK_THREADS + (ph0 << c.n4p_shift);
    if (i > 0) asm volatile("" : "+s"(qh));       // not speculated above the guards
    const ItemBits h = item_bits(c.n4p_shift, c.cc4_shift, c.nch_shift, qh);            // wave-uniform
    unsigned g_hi = static_cast<unsigned>(h.t) * c.dtap + static_cast<unsigned>(h.row) * c.ld +
                    16u * static_cast<unsigned>(h.ch << c.cc4_shift);
    asm volatile("" : "+s"(g_hi));                // stays one SGPR: no re-association with the lane part
    pf[i] = ldb(sb, g_lo + g_hi);
  };
  pass(0);
  if (n > MK_THREADS) {                           // one scalar branch skips all of this for the small images
#pragma unroll
    for (int i = 1; i < MK_MAXPF; ++i)
      if (wave_q0 + i * MK_THREADS < n) pass(i);  // wave-uniform guard
  }
}

// registers -> LDS image of layer o (phase ph lands at lds_in + (ph - ph_base) * phase bytes) + zero halo rows
__device__ __forceinline__ void image_store(const OpWords& o, float* lds_in, int ph0, int nphases, int ph_base, const FwdWin& fw,
                                            int tid, const f32x4 (&pf)[MK_MAXPF]) {
  asm volatile("" : "+v"(tid));
  const int n4p_shift = cv_n4p_shift(o), cc4_shift = cv_cc4_shift(o), nch_shift = cv_nch_shift(o);
  const int n = nphases << n4p_shift;
  const int wave_q0 = __builtin_amdgcn_readfirstlane(tid) & ~63;
  const int phase_b = cv_phase_b(o), padl = cv_padl(o);
  const ItemBits l = item_bits(n4p_shift, cc4_shift, nch_shift, tid < n ? tid : n - 1);
  const int l_lo = l.ph * phase_b + img_addr_b(o, padl + l.row, l.c4);
  const int chan4_lo = (l.ch << cc4_shift) + l.c4;
  const bool fwd_lane = fw.on && (l.row & (fw.rmul - 1)) == fw.radd;
  const int t_cur = cv_tt2(o);
  auto pass = [&](int i) {
    int qh = i * MK_THREADS + (ph0 << n4p_shift);
    if (i > 0) asm volatile("" : "+s"(qh));
    const ItemBits h = item_bits(n4p_shift, cc4_shift, nch_shift, qh);                 // wave-uniform
    // (h.row is 0 or a multiple of 512 / cc4 >= 32: even, so it moves whole stride-2 row pairs)
    int l_hi = (h.ph - ph_base) * phase_b + (cv_stride2(o) ? (h.row >> 1) : h.row) * cv_pitch_b(o);
    asm volatile("" : "+s"(l_hi));
    // forwarded by the producing layer's epilogue: current-frame tap, float4 columns [lo4,hi4), matching rows
    const int t = l.t | h.t;
    const unsigned col = static_cast<unsigned>(chan4_lo + (h.ch << cc4_shift) - fw.lo4);
    const bool skip = fwd_lane && t == t_cur && col < static_cast<unsigned>(fw.hi4 - fw.lo4);
    if ((i > 0 || tid < n) && !skip) lds4(lds_in, l_lo + l_hi) = pf[i];
  };
  pass(0);
  if (n > MK_THREADS) {
#pragma unroll
    for (int i = 1; i < MK_MAXPF; ++i)
      if (wave_q0 + i * MK_THREADS < n) pass(i);
  }
  // halo rows (<= 3 per phase, 4 slots): thread -> (phase, slot, float4 column), no division
  const int vrows = cv_vrows(o);
  const int hrows = cv_rows(o) - vrows;
  const int c4 = tid & ((1 << cc4_shift) - 1);
  const int slot = tid >> cc4_shift;
  const int hr = slot & 3, phl = slot >> 2;
  if (hr < hrows && phl < nphases) {
    const int lr = hr < padl ? hr : vrows + hr;         // top halo rows first, then the bottom ones
    // (a zero the compiler cannot hoist out of the layer loop -- it would spill it, and a scratch
    //  reload costs a vmcnt(0) wait right here)
    float zf = 0.f;
    asm volatile("" : "+v"(zf));
    const f32x4 z = {zf, zf, zf, zf};
    lds4(lds_in, (ph0 - ph_base + phl) * phase_b + img_addr_b(o, lr, c4)) = z;
  }
}

// Direct writes into the NEXT layer's LDS image (current-frame tap): a float4 column / a single channel of row `row`
__device__ __forceinline__ void img_put4(const OpWords& n, float* lds_in, int row, int chan4, f32x4 v) {
  const int cc4_shift = cv_cc4_shift(n);
  const int ph = (cv_tt2(n) << cv_nch_shift(n)) + (chan4 >> cc4_shift);
  lds4(lds_in, ph * cv_phase_b(n) + img_addr_b(n, cv_padl(n) + row, chan4 & ((1 << cc4_shift) - 1))) = v;
}
__device__ __forceinline__ void img_put1(const OpWords& n, float* lds_in, int row, int chan, float v) {
  const int chan4 = chan >> 2;
  const int cc4_shift = cv_cc4_shift(n);
  const int ph = (cv_tt2(n) << cv_nch_shift(n)) + (chan4 >> cc4_shift);
  lds1(lds_in, ph * cv_phase_b(n) + img_addr_b(n, cv_padl(n) + row, chan4 & ((1 << cc4_shift) - 1)) + 4 * (chan & 3)) = v;
}

// ---------------------------------------------------------------------------------------------
//  What travels in registers from op to op: the next conv layer's first weight chunk (this wave's
//  task) and its epilogue parameters (this lane's float4 of bias / gamma / beta).  Fetched while
//  the current op runs its epilogue, complete at the op boundary.
// ---------------------------------------------------------------------------------------------
struct Carry { f32x4 w[4]; f32x4 e[3]; };

__device__ __forceinline__ void prefetch_conv(const OpWords& n, gcb_t wb, int wave, int tid, Carry& cy) {
  const int nt = cv_nt(n);
  const bool act = wave < cv_tasks(n) * cv_KS(n);
  const int ks = act ? (wave >> cv_tasks_shift(n)) : 0;
  const int ntile = act ? (wave & (nt - 1)) : 0;
  const unsigned w0 = n.w[4] + static_cast<unsigned>(ks * cv_gpk(n) * nt + ntile) * 1024u;      // wave-uniform
  const unsigned wstep_b = static_cast<unsigned>(nt) << 10;
  const unsigned lane16 = static_cast<unsigned>(tid & 63) * 16u;
#pragma unroll
  for (int u = 0; u < 4; ++u) cy.w[u] = ldb(wb, w0 + u * wstep_b + lane16);
  const int lpg = cv_lpg(n);
  const unsigned li16 = static_cast<unsigned>(tid & (lpg - 1)) * 16u;
  const unsigned gi = cv_R2(n) ? static_cast<unsigned>((tid >> (3 + cv_gcode(n))) & 1) : 0u;     // lpg = 8 << gcode
  cy.e[0] = ldb(wb, n.w[5] + gi * static_cast<unsigned>(lpg * 16) + li16);
  cy.e[1] = ldb(wb, n.w[6] + li16);      // (layers without LayerNorm: gamma = beta = bias offset)
  cy.e[2] = ldb(wb, n.w[7] + li16);
}

// cursor over the resident image for one task: (local phase, frequency tap, channel group), wave-uniform
struct KCursor { int phl, kf, gg; };

__device__ __forceinline__ KCursor kcursor_init(const OpWords& o, int ks) {
  // (no integer divisions: gpc is 4 or 8, kf is 1, 2 or 3, seg < 64)
  const int gpc = cv_gpc(o), kf = cv_kf(o);
  const int g_first = ks * cv_gpk(o);
  const int gshift = gpc == 8 ? 3 : (gpc == 4 ? 2 : 1);
  const int seg = g_first >> gshift;
  KCursor k;
  k.gg = g_first & (gpc - 1);
  k.phl = kf == 3 ? ((seg * 43) >> 7) : (kf == 2 ? (seg >> 1) : seg);
  k.kf = seg - k.phl * kf;
  return k;
}

// MFMAs of one 4-fragment chunk = two fragment pairs (a pair never straddles a frequency-tap segment;
// lbA / lbB: B-operand offsets of the pairs).
//   32-position tiles (v_mfma_f32_32x32x2_f32): a fragment is 8 channels, 4 MFMAs per fragment and tile; the
//     first tile's accumulator has ONE update site, the second tile's (16-tile layers) a conditional in-place
//     one -- any second site (a two-sided branch, a conditional tail) makes the register allocator copy the
//     accumulators in and out of temporaries around every chunk, each copy waiting for the MFMA chain.
//   16-position tiles (v_mfma_f32_16x16x4_f32, layers with F_out <= 16): a fragment is 16 channels.
__device__ __forceinline__ void chunk_mfma32(f32x16& acc0, f32x16& acc1, bool two, const f32x4 (&w)[4], float* lds_in, int lbA0, int lbB0,
                                             int lbA1, int lbB1) {
  f32x4 b[4];
  b[0] = lds4(lds_in, lbA0); b[1] = lds4(lds_in, lbA0 + 32); b[2] = lds4(lds_in, lbB0); b[3] = lds4(lds_in, lbB0 + 32);
  f32x4 d[4] = {b[0], b[1], b[2], b[3]};
  if (two) {
    d[0] = lds4(lds_in, lbA1); d[1] = lds4(lds_in, lbA1 + 32); d[2] = lds4(lds_in, lbB1); d[3] = lds4(lds_in, lbB1 + 32);
  }
  // the second tile's MFMAs are interleaved with the first tile's (two independent accumulator chains keep the
  // pipe full from one wave), each behind its own wave-uniform branch: acc0 keeps a single update site
#pragma unroll
  for (int u = 0; u < 4; ++u) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(w[u][j], b[u][j], acc0, 0, 0, 0);
      if (two) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(w[u][j], d[u][j], acc1, 0, 0, 0);
    }
  }
}
__device__ __forceinline__ void chunk_mfma16(f32x4& acc, const f32x4 (&w)[4], float* lds_in, int lbA, int lbB) {
  f32x4 b[4];
  b[0] = lds4(lds_in, lbA); b[1] = lds4(lds_in, lbA + 64); b[2] = lds4(lds_in, lbB); b[3] = lds4(lds_in, lbB + 64);
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w[u][j], b[u][j], acc, 0, 0, 0);
}

// DPP lane exchange (no LDS crossbar round trip): quad xor 1, quad xor 2, mirror inside 8 / 16 lanes
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
template <int LPG>
__device__ __forceinline__ float group_sum(float s) {      // sum over LPG consecutive lanes (aligned), result in all of them
  s += dpp_mov<0xB1>(s);                  // quad_perm [1,0,3,2]
  s += dpp_mov<0x4E>(s);                  // quad_perm [2,3,0,1]
  if (LPG >= 8) s += dpp_mov<0x141>(s);   // row_half_mirror: the other quad of the 8
  if (LPG >= 16) s += dpp_mov<0x140>(s);  // row_mirror: the other 8 of the 16
  if (LPG >= 32) s += __shfl_xor(s, 16);
  return s;
}

// ---------------------------------------------------------------------------------------------
//  Row-wise epilogue: LPG lanes per output row (row = LPG*4 channels), optional LN + PReLU; writes
//  HBM destinations and, when `do_fwd`, forwards the rows into the next layer's LDS image.
// ---------------------------------------------------------------------------------------------
template <int LPG, bool LN>
__device__ __forceinline__ void conv_epilogue(const OpWords& o, gcb_t sb, float* lds_out, int tid, f32x4 bias, f32x4 gm, f32x4 bt,
                                              bool do_fwd, const OpWords& nx, float* lds_next) {
  constexpr int GC = LPG * 4;
  const int units = cv_units(o);
  int u = tid / LPG;
  if (u >= units) return;
  const int li = tid & (LPG - 1);
  const int R2 = cv_R2(o), KS = cv_KS(o), slot_b = cv_slot_b(o), opitch_b = cv_opitch_b(o);
  const unsigned d0 = o.w[2] + 16u * li, d1 = o.w[3] + 16u * li;
  const int ld0 = cv_ld0_b(o), ld1 = cv_ld1_b(o), row_mul = cv_row_mul(o), row_add = cv_row_add(o);
  const bool has1 = cv_has_dst1(o);
  const float alpha = __uint_as_float(o.w[8]);
  // forwarded block inside the next image: float4 column (coff4 + li) of the current-frame phase
  int f_base = 0, f_c4 = 0;
  if (do_fwd) {
    const int chan4 = cv_fwd_coff4(o) + li;
    const int cs = cv_cc4_shift(nx);
    f_base = ((cv_tt2(nx) << cv_nch_shift(nx)) + (chan4 >> cs)) * cv_phase_b(nx);
    f_c4 = chan4 & ((1 << cs) - 1);
  }
  for (; u < units; u += MK_THREADS / LPG) {
    const int pos = R2 ? (u >> 1) : u, gi = R2 ? (u & 1) : 0;
    const int ob = pos * opitch_b + gi * (GC * 4) + 16 * li;
    f32x4 v = bias;
    for (int ks = 0; ks < KS; ++ks) v += lds4(lds_out, ob + ks * slot_b);
    if (LN) {
      const float mean = group_sum<LPG>(v[0] + v[1] + v[2] + v[3]) * (1.0f / GC);
      v -= mean;
      const float q = group_sum<LPG>(v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3]);
      const float rstd = __builtin_amdgcn_rsqf(q * (1.0f / GC) + MK_LN_EPS);     // v_rsq_f32, 1 ulp
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float y = v[i] * rstd * gm[i] + bt[i];
        v[i] = y >= 0.f ? y : alpha * y;
      }
    }
    const int row = pos * row_mul + row_add + gi;
    stb(sb, d0 + static_cast<unsigned>(row * ld0), v);
    if (has1) stb(sb, d1 + static_cast<unsigned>(row * ld1), v);
    if (do_fwd) lds4(lds_next, f_base + img_addr_b(nx, cv_padl(nx) + row, f_c4)) = v;
  }
}

// One conv-like layer for one stream.
//   cy  in : this wave's first weight chunk + this lane's epilogue parameters (fetched by the previous op)
//       out: the same for the next conv layer (a layer with no conv successor re-reads its own)
template <bool PROF>
__device__ __forceinline__ void conv_layer(const OpWords& o_in, OpWords& n, cplan_t plan, int op_i, int nxt_i, gcb_t sb, gcb_t wb,
                                           float* lds_in,
                                           float* lds_out, int tid, Carry& cy, unsigned long long* sub, unsigned long long* dbg) {
  unsigned long long* dbg_lds = reinterpret_cast<unsigned long long*>(lds_out + MK_LDS_OUT);
  MK_T(0);
  OpWords o = o_in;                // (re-read after the MFMA loop, see below; the next op's words arrive there too)
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // wave-uniform -> SGPR
  const bool s16 = cv_s16(o);
  const int pl = s16 ? (lane & 15) : (lane & 31), h = s16 ? (lane >> 4) : (lane >> 5);     // tile column, K sub-block of the lane
  const int tasks = cv_tasks(o), nt = cv_nt(o), tw = cv_tw(o), F_out = cv_F_out(o);
  const bool active = wave < tasks * cv_KS(o);
  const int ks = wave >> cv_tasks_shift(o), tl = wave & (tasks - 1);
  const int pt = (tl >> cv_nt_shift(o)) * tw, ntile = tl & (nt - 1);     // first position tile, channel tile
  int pc0 = pt * 32 + pl, pc1 = pc0 + 32;
  if (pc0 > F_out - 1) pc0 = F_out - 1;        // padding lanes recompute the last position
  if (pc1 > F_out - 1) pc1 = F_out - 1;
  const int pitch_b = cv_pitch_b(o);
  const int lane_b0 = pc0 * pitch_b + 16 * h, lane_b1 = pc1 * pitch_b + 16 * h;
  const FwdWin nofw = {0, 0, 1, 0, false};

  // weight stream of this task: byte offsets from the weight arena (wave-uniform) + lane part
  const unsigned wstep_b = static_cast<unsigned>(nt) << 10;                 // one 8-channel group: nt * 64 float4
  const unsigned round_step_b = static_cast<unsigned>(cv_RG(o)) * wstep_b;
  const unsigned wbase_b = o.w[4] + static_cast<unsigned>((active ? ks * cv_gpk(o) : 0) * nt + (active ? ntile : 0)) * 1024u;
  const unsigned lane16 = static_cast<unsigned>(lane) * 16u;

  // ---- first weight chunk and epilogue parameters: fetched by the previous op
  f32x4 wa[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) wa[u] = cy.w[u];
  const f32x4 bias = cy.e[0], gm = cy.e[1], bt = cy.e[2];
  MK_STAMP(0);

  // ---- LDS image of round 0 (skipped when the previous layer handed it over complete)
  const int nstage = cv_merged(o) ? cv_nph(o) : 1;
  if (!cv_staged_by_prev(o)) {
    f32x4 pf[MK_MAXPF];
    image_load(img_src(o), sb, 0, nstage, tid, pf);
    image_store(o, lds_in, 0, nstage, 0, nofw, tid, pf);
    lds_barrier();
  }
  MK_STAMP(1);
  MK_T(2);

  // ---- what this layer owes the next one
  const bool hand = cv_hand_next(o);       // (planned only in front of a conv layer)
  const bool do_fwd = hand && cv_fwd_sel(o);
  const int n_hand = cv_hx_nhand(o);
  MK_STAMP(6);
  MK_T(3);

  f32x16 acc[2];
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[0][r] = 0.f;
  if (tw == 2) {
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[1][r] = 0.f;
  }
  f32x4 acc16 = {0.f, 0.f, 0.f, 0.f};
  // One register set and ONE static load site serve both image prefetches -- the next phase of THIS
  // layer (rounds before the last) and, in the last round, the next layer's image rows this layer does
  // not produce (previous-frame tap, skip-connection channels; HBM, long latency).  Both are issued in
  // the round's last chunk, right after its weight prefetch: the youngest loads (vmcnt retires in
  // order).  Two load sites would meet in phi copies of pending loads, and such a copy costs a full
  // vmcnt(0) wait.  None of the barriers in between drains vmcnt.
  f32x4 pfx[MK_MAXPF];
  const int gpk = active ? cv_gpk(o) : 4;                // idle waves run one empty chunk per round: they stage too
  const int rounds = cv_rounds(o), kf_n = cv_kf(o), gpc = cv_gpc(o), phase_b = cv_phase_b(o);
  const int cc_b = cv_cc_b(o), stride2 = cv_stride2(o);
  const int frag_b = s16 ? 64 : 32;                      // bytes of image channels one K fragment covers
#pragma unroll 1
  for (int rd = 0; rd < rounds; ++rd) {
    const bool last = rd + 1 == rounds;
    KCursor k = kcursor_init(o, ks);
    if (!active) k.gg = 0;
    unsigned wcur = wbase_b + static_cast<unsigned>(rd) * round_step_b;
    auto boff_of = [&](const KCursor& q) {
      const int koff = stride2 ? ((q.kf >> 1) * pitch_b + (q.kf & 1) * cc_b) : q.kf * pitch_b;
      return q.phl * phase_b + koff + frag_b * q.gg;
    };
    auto advance = [&](KCursor& q, int ng) {
      q.gg += ng;
      if (q.gg == gpc) { q.gg = 0; if (++q.kf == kf_n) { q.kf = 0; ++q.phl; } }
    };
    // chunks of 4 fragments = two pairs; only 2-fragment segments (16-position tiles of a 32-channel
    // image) put the second pair into the next segment
    int rem = gpk;
#pragma unroll 1
    while (rem > 0) {
      constexpr int nfr = 4;
      rem -= nfr;
      const int boffA = boff_of(k);
      int boffB = boffA + 2 * frag_b;
      if (gpc == 2) {
        advance(k, 2);
        boffB = boff_of(k);
        advance(k, 2);
      } else {
        advance(k, 4);
      }
      // prefetch the next chunk: next in this slice, else the first chunk of the next round's slice
      // (wave-uniform condition -> scalar branch; nothing is fetched after the layer's last chunk)
      const bool more = active && (rem > 0 || !last);
      f32x4 wn[4];
      if (more) {
        const unsigned nxt = rem > 0 ? wcur + nfr * wstep_b : wbase_b + static_cast<unsigned>(rd + 1) * round_step_b;
#pragma unroll
        for (int u = 0; u < 4; ++u) wn[u] = ldb(wb, nxt + u * wstep_b + lane16);
      }
      // (rounds before the last: with the round's FIRST chunk, so that the rows have a whole round to arrive;
      //  the last round: with the last chunk, behind every weight load of the layer)
      if (last ? (rem == 0 && hand) : rem + nfr == gpk) {
        ImgSrc is;
        is.src0 = last ? o.w[22] : o.w[0]; is.dtap = last ? o.w[17] : o.w[1];
        is.ld = static_cast<unsigned>(last ? cv_hx_ld_b(o) : cv_src_ld_b(o));
        is.cc4_shift = last ? cv_hx_cc4_shift(o) : cv_cc4_shift(o);
        is.n4p_shift = last ? cv_hx_n4p_shift(o) : cv_n4p_shift(o);
        is.nch_shift = last ? cv_hx_nch_shift(o) : cv_nch_shift(o);
        image_load(is, sb, last ? 0 : rd + 1, last ? n_hand : 1, tid, pfx);
        MK_STAMP(7);
        MK_T(4);
      }
      if (active) {
        if (s16) chunk_mfma16(acc16, wa, lds_in, lane_b0 + boffA, lane_b0 + boffB);
        else chunk_mfma32(acc[0], acc[1], tw == 2, wa, lds_in, lane_b0 + boffA, lane_b0 + boffB, lane_b1 + boffA, lane_b1 + boffB);
      }
      if (more) {
#pragma unroll
        for (int u = 0; u < 4; ++u) wa[u] = wn[u];
      }
      wcur += nfr * wstep_b;
    }
    if (!last) {
      lds_barrier();             // every wave is done reading this phase's LDS rows
      image_store(o, lds_in, rd + 1, 1, rd + 1, nofw, tid, pfx);
      lds_barrier();
    }
  }
  MK_T(5);
  // Both descriptors again, by fresh scalar loads: the ~40 words the code below needs are then not live
  // (= not spilled to VGPR lanes and read back one v_readlane at a time) across the MFMA loop above.
  o = reload_op(plan, op_i);
  n = reload_op(plan, nxt_i);
  const bool nconv = nxt_i != op_i && static_cast<int>(n.w[23]) == DEV_OP_CONV;
  // next conv layer's first weight chunk + epilogue parameters (single load site)
  {
    OpWords s;     // (field-wise selects of the words prefetch_conv reads: a layer with no conv successor re-reads its own)
    s.w[4] = nconv ? n.w[4] : o.w[4]; s.w[5] = nconv ? n.w[5] : o.w[5]; s.w[6] = nconv ? n.w[6] : o.w[6]; s.w[7] = nconv ? n.w[7] : o.w[7];
    s.w[10] = nconv ? n.w[10] : o.w[10]; s.w[12] = nconv ? n.w[12] : o.w[12]; s.w[18] = nconv ? n.w[18] : o.w[18];
    s.w[19] = nconv ? n.w[19] : o.w[19];
    prefetch_conv(s, wb, wave, tid, cy);
  }
  MK_STAMP(2);
  MK_T(6);

  // ---- partial tiles -> LDS exchange buffer [ks][pos][32*NT (+4)]
  if (active && s16) {
    // 16x16 tile: lane holds channels 16*ntile + 4*h .. +3 of position pl
    lds4(lds_out, ks * cv_slot_b(o) + pl * cv_opitch_b(o) + ntile * 64 + 16 * h) = acc16;
  } else if (active) {
    const int opitch_b = cv_opitch_b(o);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      if (t < tw) {
        const int ob = ks * cv_slot_b(o) + ((pt + t) * 32 + pl) * opitch_b + ntile * 128 + 16 * h;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 v = {acc[t][4 * q], acc[t][4 * q + 1], acc[t][4 * q + 2], acc[t][4 * q + 3]};
          lds4(lds_out, ob + 32 * q) = v;
        }
      }
    }
  }
  MK_T(7);
  lds_barrier();                 // also: every wave has finished reading lds_in
  MK_STAMP(3);
  MK_T(8);

  const int gcode = cv_gcode(o);
  if (cv_epi_ln(o)) {
    if (gcode == 0) conv_epilogue<8, true>(o, sb, lds_out, tid, bias, gm, bt, do_fwd, n, lds_in);
    else conv_epilogue<16, true>(o, sb, lds_out, tid, bias, gm, bt, do_fwd, n, lds_in);
  } else {
    if (gcode == 1) conv_epilogue<16, false>(o, sb, lds_out, tid, bias, gm, bt, do_fwd, n, lds_in);
    else conv_epilogue<32, false>(o, sb, lds_out, tid, bias, gm, bt, do_fwd, n, lds_in);
  }
  MK_T(9);
  // ---- hand-off: the prefetched part of the next layer's image (the forwarded rows were written above)
  if (hand) {
    FwdWin fw = nofw;
    if (do_fwd) {
      fw.lo4 = cv_fwd_coff4(o); fw.hi4 = fw.lo4 + cv_lpg(o); fw.rmul = 1 + cv_fwd_rmul2(o); fw.radd = cv_fwd_radd(o); fw.on = true;
    }
    image_store(n, lds_in, 0, n_hand, 0, fw, tid, pfx);
  }
  MK_STAMP(4);
  MK_T(10);
  __syncthreads();               // HBM stores visible to the workgroup; next image complete; exchange buffer free
  MK_STAMP(5);
  MK_T(11);
  if (PROF && dbg) {
    __syncthreads();
    if (tid < 128 && (tid & 15) < 12) dbg[tid] = dbg_lds[tid];
  }
}

// Fast transcendental forms (v_exp_f32 / v_rcp_f32, ~1 ulp): the IEEE expf / division sequences are
// 10-30 dependent instructions each, on the critical path of the 21..64 threads that evaluate gates.
__device__ __forceinline__ float fast_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float fast_tanh(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * x)); }


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
__device__ __forceinline__ void lstm_layer(const LstmParams& p, int stream, gcb_t sbb, float* lds, float* lds_in, int tid, bool hand,
                                           int fwd_coff, const OpWords& nx) {
  float* part = lds;               // [16][84]
  float* z = lds + 16 * 84;        // [96]
  float* hn = z + 96;              // [32]
  const float* sb = p.x + static_cast<size_t>(stream) * p.sstride;         // stream slice of the input tensor
  const int n4 = tid % 21, sl = tid / 21;                                    // sl < 16 for tid < 336
  // x_cols / dst_cols are 32 or 64: shifts, not run-time integer divisions (~40 instructions each, one per K element)
  const int xshift = p.x_cols == 64 ? 6 : 5, dshift = p.dst_cols == 64 ? 6 : 5;
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
        const int f = k >> xshift, c = k & (p.x_cols - 1);
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
  if (hand) image_load(img_src(nx), sbb, 0, cv_nph(nx), tid, pfx);

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
    const float gi = fast_sigmoid(z[tid]), gf = fast_sigmoid(z[21 + tid]);
    const float gg = fast_tanh(z[42 + tid]), go = fast_sigmoid(z[63 + tid]);
    const float c_new = gf * c_old + gi * gg;
    const float h_new = go * fast_tanh(c_new);
    st1(p.c_out + static_cast<size_t>(stream) * p.sstride, static_cast<unsigned>(tid), c_new);
    st1(p.h_out + static_cast<size_t>(stream) * p.sstride, static_cast<unsigned>(tid), h_new);
    hn[tid] = h_new;
  }
  lds_barrier();
  if (tid < p.Dout) {
    float a = bd;
#pragma unroll
    for (int u = 0; u < 21; ++u) a = fmaf(wd[u], hn[u], a);
    const int f = tid >> dshift, c = tid & (p.dst_cols - 1);
    st1(p.dst + static_cast<size_t>(stream) * p.sstride, static_cast<unsigned>(f * p.dst_ld + c), a);
    if (hand) img_put1(nx, lds_in, f, fwd_coff + c, a);
  }
  if (hand) {
    const FwdWin fw = {fwd_coff >> 2, (fwd_coff + p.dst_cols) >> 2, 1, 0, true};
    image_store(nx, lds_in, 0, cv_nph(nx), 0, fw, tid, pfx);
  }
  __syncthreads();
}

// 64 -> 16 (ReLU) -> 64 perceptron of the CTFA gates, evaluated by ONE wave without barriers: lane c
// holds input channel c.  The 16 hidden sums over the 64 lanes go through a [64][20] LDS scratch (lane
// (q, u) adds 16 of the 64 products of unit u, then the four quarters meet in two lane exchanges); the
// hidden vector is broadcast from lanes 0..15 with v_readlane.  Returns the pre-activation of channel c.
__device__ __forceinline__ float gate_mlp(float in, const f32x4 (&w1)[4], float b1_u, const f32x4 (&w2)[4], float b2, float* scr, int lane) {
#pragma unroll
  for (int q = 0; q < 4; ++q) *reinterpret_cast<f32x4*>(scr + lane * 20 + 4 * q) = w1[q] * in;
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // same wave wrote it: no barrier needed
  const int u = lane & 15, qtr = lane >> 4;
  float h0 = 0.f, h1 = 0.f;
#pragma unroll
  for (int i = 0; i < 16; i += 2) {
    h0 += scr[(qtr * 16 + i) * 20 + u];
    h1 += scr[(qtr * 16 + i + 1) * 20 + u];
  }
  float hsum = h0 + h1;
  hsum += __shfl_xor(hsum, 16);
  hsum += __shfl_xor(hsum, 32);
  const float hid = fmaxf(hsum + b1_u, 0.f);               // lanes 0..15 (and their copies): hidden unit u
  float a0 = b2, a1 = 0.f;
#pragma unroll
  for (int k = 0; k < 16; k += 2) {
    a0 = fmaf(w2[k >> 2][k & 3], __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, hid), k)), a0);
    a1 = fmaf(w2[(k + 1) >> 2][(k + 1) & 3], __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, hid), k + 1)), a1);
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // scratch reads done before the next call overwrites it
  return a0 + a1;
}

// CTFA gate + residual for one stream (ctfa_rt, models/proposed.py:162-196).
// One memory round trip and three barriers: every thread fetches its rows of x and of the residual (and
// keeps them in registers), the column sums of x meet in LDS, wave 0 alone evaluates both gate
// perceptrons (no barriers inside), then everybody applies the gate to the rows it still holds.
template <bool PROF>
__device__ __forceinline__ void ctfa_layer(const CtfaParams& p, int stream, gcb_t sbb, float* lds, float* lds_in, int tid, bool hand,
                                           int fwd_coff, const OpWords& nx, unsigned long long* dbg) {
  unsigned long long* dbg_lds = reinterpret_cast<unsigned long long*>(lds + MK_LDS_OUT);
  MK_T(0);
  float* part = lds;               // [8][64]   column sums per wave
  float* gate = lds + 512;         // [64]
  float* scr = gate + 64;          // [64][20]  perceptron scratch (wave 0)
  const int c4 = tid & 15, rg = tid >> 4;            // 32 row groups x 16 float4 columns
  const int lane = tid & 63;
  const bool w0 = tid < 64;
  const float* xb = p.x + static_cast<size_t>(stream) * p.sstride;
  const float* eb = p.e0 + static_cast<size_t>(stream) * p.sstride;
  float* yb = p.y + static_cast<size_t>(stream) * p.sstride;
  // ---- every global load of the op, oldest-needed first
  f32x4 xv[8], ev[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int f = rg + 32 * i;
    if (f < p.F) xv[i] = ld4(xb, static_cast<unsigned>(f * p.x_ld + 4 * c4));
  }
  f32x4 w1t[4], w1f[4], w2t[4], w2f[4];
  float b1t = 0.f, b1f = 0.f, b2t = 0.f, b2f = 0.f;
  if (w0) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      w1t[q] = ld4(p.ta_w1T, static_cast<unsigned>(lane * 16 + 4 * q));
      w2t[q] = ld4(p.ta_w2, static_cast<unsigned>(lane * 16 + 4 * q));
      w1f[q] = ld4(p.fa_w1T, static_cast<unsigned>(lane * 16 + 4 * q));
      w2f[q] = ld4(p.fa_w2, static_cast<unsigned>(lane * 16 + 4 * q));
    }
    b1t = ld1(p.ta_b1, static_cast<unsigned>(lane & 15)); b1f = ld1(p.fa_b1, static_cast<unsigned>(lane & 15));
    b2t = ld1(p.ta_b2, static_cast<unsigned>(lane)); b2f = ld1(p.fa_b2, static_cast<unsigned>(lane));
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int f = rg + 32 * i;
    if (f < p.F) ev[i] = ld4(eb, static_cast<unsigned>(f * p.e0_ld + 4 * c4));
  }
  f32x4 pfx[MK_MAXPF];
  if (hand) image_load(img_src(nx), sbb, 0, cv_nph(nx), tid, pfx);
  MK_T(1);

  // ---- column sums of x
  f32x4 s4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 8; ++i)
    if (rg + 32 * i < p.F) s4 += xv[i];
  // the wave's four row groups meet in registers (lanes 16 / 32 apart hold the same float4 column)
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    s4[j] += __shfl_xor(s4[j], 16);
    s4[j] += __shfl_xor(s4[j], 32);
  }
  MK_T(2);
  if (lane < 16) *reinterpret_cast<f32x4*>(part + (tid >> 6) * 64 + 4 * c4) = s4;
  lds_barrier();
  MK_T(3);
  if (w0) {
    float m = 0.f;
#pragma unroll
    for (int r = 0; r < MK_WAVES; ++r) m += part[r * 64 + lane];
    m = m * __builtin_amdgcn_rcpf(static_cast<float>(p.F));      // F is a power of two: exact
    MK_T(4);
    const float ta_c = fast_sigmoid(gate_mlp(m, w1t, b1t, w2t, b2t, scr, lane));
    MK_T(5);
    const float fa_c = fast_sigmoid(gate_mlp(ta_c * (1.0f / 32.0f), w1f, b1f, w2f, b2f, scr, lane));
    gate[lane] = fa_c * ta_c;
    MK_T(6);
  }
  lds_barrier();
  MK_T(7);
  const f32x4 g4 = *reinterpret_cast<const f32x4*>(gate + 4 * c4);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int f = rg + 32 * i;
    if (f < p.F) {
      const f32x4 yv = xv[i] * g4 + ev[i];
      st4(yb, static_cast<unsigned>(f * p.y_ld + 4 * c4), yv);
      if (hand) img_put4(nx, lds_in, f, (fwd_coff >> 2) + c4, yv);
    }
  }
  MK_T(8);
  if (hand) {
    const FwdWin fw = {fwd_coff >> 2, (fwd_coff >> 2) + 16, 1, 0, true};
    image_store(nx, lds_in, 0, cv_nph(nx), 0, fw, tid, pfx);
  }
  MK_T(9);
  __syncthreads();
  MK_T(10);
  if (PROF && dbg) {
    __syncthreads();
    if (tid < 128 && (tid & 15) < 12) dbg[tid] = dbg_lds[tid];
  }
}

__device__ __forceinline__ void input_layer_op(const OpWords& o, const StepArgs& a, int stream, float* lds_in, int tid, bool hand,
                                               const OpWords& nx) {
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
    const float s = group_sum<16>(y[0] + y[1] + y[2] + y[3]);           // DPP lane exchanges (16 lanes = one position)
    y -= s * (1.0f / 64.0f);
    const float q = group_sum<16>(y[0] * y[0] + y[1] * y[1] + y[2] * y[2] + y[3] * y[3]);
    const float rstd = __builtin_amdgcn_rsqf(q * (1.0f / 64.0f) + MK_LN_EPS);
    f32x4 o4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float t = y[i] * rstd * gm[i] + bt[i];
      o4[i] = t >= 0.f ? t : p.alpha * t;
    }
    *G4W(p.y + static_cast<size_t>(stream) * p.sstride + static_cast<size_t>(pos) * 64 + 4 * c4) = o4;
    if (hand) img_put4(nx, lds_in, pos, c4, o4);     // the whole image of msfe6_en_in is this op's output
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
    const float s = group_sum<16>(xv[0] * w[0] + xv[1] * w[1] + xv[2] * w[2] + xv[3] * w[3]);
    if (c4 == 0) GFW(p.y)[static_cast<size_t>(stream) * NUTLS_DEV_BINS + pos] = s + p.bias;
  }
  __syncthreads();
}

// PROF = true: the profiling build of the same kernel (wall-clock stamps at op / phase boundaries of
// workgroup 0); the production build carries none of the stamp code.
// DDB = true: the baseline variant's kernel (dilated-dense bottlenecks, no LSTM code); the two model variants never
// share a plan, and each build stays small enough for the instruction cache.
template <bool PROF, bool DDB>
__global__ __launch_bounds__(MK_THREADS) void nutls_stream_step_kernel(const StepArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* lds_in = lds;
  float* lds_out = lds + MK_LDS_IN;
  const int n_ops = a.n_ops;
  unsigned long long* prof = PROF ? a.prof : nullptr;
  const cplan_t plan = (cplan_t)(unsigned long long)a.plan;
  const gcb_t wb = (gcb_t)(unsigned long long)a.wbase;
  // `fresh_tid()` re-materialises the thread id behind an opaque asm at every layer: without it the
  // compiler hoists dozens of tid-derived per-thread constants out of the layer loop, keeps them live
  // across the whole kernel and spills them -- and a scratch reload is a vmcnt wait, which in this
  // kernel means "wait for every prefetch in flight".
  auto fresh_tid = []() { int t = threadIdx.x; asm volatile("" : "+v"(t)); return t; };
  for (int stream = blockIdx.x; stream < a.B; stream += gridDim.x) {
    if (prof && blockIdx.x == 0 && threadIdx.x == 0) prof[n_ops * 9 + 1] = clock64();
    const gcb_t sb = (gcb_t)(unsigned long long)(a.arena + static_cast<size_t>(stream) * a.sstride);   // this stream's arena slice
    Carry cy;
    OpWords cur = load_op(plan, 0);
#pragma unroll 1
    for (int i = 0; i < n_ops; ++i) {
      const int tid = fresh_tid();
      const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
      if (prof && blockIdx.x == 0 && tid == 0) prof[i] = wall_clock64();
      const int nxt_i = i + 1 < n_ops ? i + 1 : i;
      const int op = static_cast<int>(cur.w[23]);
      unsigned long long* sub = (prof && blockIdx.x == 0) ? prof + (n_ops + 1) + 8 * i : nullptr;
      unsigned long long* dbg = (sub && i == a.dbg_op) ? prof + n_ops * 9 + 3 : nullptr;
      if (dbg && (tid & 63) == 0) dbg[(tid >> 6) * 16 + 15] = clock64();
      if (dbg) __syncthreads();      // (the stamp's store must not sit in vmcnt during the op)
      OpWords nxt;                   // the next op's words: a conv layer reads them late (after its MFMA loop)
      if (op == DEV_OP_CONV) {
        conv_layer<PROF>(cur, nxt, plan, i, nxt_i, sb, wb, lds_in, lds_out, tid, cy, sub, dbg);
      } else {
        nxt = load_op(plan, nxt_i);
        const bool nconv = (i + 1 < n_ops) && static_cast<int>(nxt.w[23]) == DEV_OP_CONV;
        const bool nc_hand = nconv && b0(cur.w[22]) != 0;      // non-conv op -> conv hand-off
        const int nc_coff = b1(cur.w[22]);
        // the next conv layer's first weights / epilogue parameters: requested before the op's own loads, so that
        // they have arrived when the op ends (requested after it, the next layer would start with an L2 round trip)
        if (nconv) prefetch_conv(nxt, wb, wave, tid, cy);
        if (!DDB && op == DEV_OP_LSTM) {
          LstmParams p;
          decode_lstm(cur, a, p);
          lstm_layer(p, stream, sb, lds_out, lds_in, tid, nc_hand, nc_coff, nxt);
        } else if (op == DEV_OP_CTFA) {
          CtfaParams p;
          decode_ctfa(cur, a, p);
          ctfa_layer<PROF>(p, stream, sb, lds_out, lds_in, tid, nc_hand, nc_coff, nxt, dbg);
        } else if (op == DEV_OP_INLAYER) {
          input_layer_op(cur, a, stream, lds_in, tid, nc_hand, nxt);
        } else if (DDB && op == DEV_OP_DDB) {
          ddb_block_wg<MK_THREADS>(a.ddb[cur.w[0]], stream, lds_out, tid,
                       (PROF && dbg) ? reinterpret_cast<unsigned long long*>(lds_out + MK_LDS_OUT) : nullptr);
          if (PROF && dbg) {
            __syncthreads();
            if (tid < 128 && (tid & 15) < 12) dbg[tid] = reinterpret_cast<unsigned long long*>(lds_out + MK_LDS_OUT)[tid];
          }
        } else {
          out_conv_op(cur, a, stream, tid);
        }
      }
      cur = nxt;
    }
    if (prof && blockIdx.x == 0 && threadIdx.x == 0) { prof[n_ops] = wall_clock64(); prof[n_ops * 9 + 2] = clock64(); }
  }
}

// Dynamic-LDS limit of the four builds, on the CURRENT device (function attributes are per device: every handle sets them
// for its own device when it is created).
hipError_t stream_step_set_attributes() {
  const void* fns[4] = {reinterpret_cast<const void*>(nutls_stream_step_kernel<false, false>),
                        reinterpret_cast<const void*>(nutls_stream_step_kernel<true, false>),
                        reinterpret_cast<const void*>(nutls_stream_step_kernel<false, true>),
                        reinterpret_cast<const void*>(nutls_stream_step_kernel<true, true>)};
  for (const void* f : fns) {
    hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(MK_LDS_BYTES));
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

hipError_t launch_stream_step(const StepArgs& a, int grid, hipStream_t s) {
  const bool prof = a.prof != nullptr, ddb = a.ddb != nullptr;
  if (!prof && !ddb) hipLaunchKernelGGL((nutls_stream_step_kernel<false, false>), dim3(grid), dim3(MK_THREADS), MK_LDS_BYTES, s, a);
  else if (prof && !ddb) hipLaunchKernelGGL((nutls_stream_step_kernel<true, false>), dim3(grid), dim3(MK_THREADS), MK_LDS_BYTES, s, a);
  else if (!prof && ddb) hipLaunchKernelGGL((nutls_stream_step_kernel<false, true>), dim3(grid), dim3(MK_THREADS), MK_LDS_BYTES, s, a);
  else hipLaunchKernelGGL((nutls_stream_step_kernel<true, true>), dim3(grid), dim3(MK_THREADS), MK_LDS_BYTES, s, a);
  return hipGetLastError();
}

}  // namespace nutls
