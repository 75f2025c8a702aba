// Fused frame-step kernel: the whole NUNet-TLS-LSTM step of one stream as ONE specialised instruction
// stream per op.  One 512-thread workgroup (8 waves) owns a stream (streams are independent);
// the 154 ops of the step (fused_plan_lstm.inc, written by tools/gen_fused_plan.py) are instantiated one
// after the other from templates, so every shape, LDS address, arena offset and weight offset is an
// immediate -- there is no plan to decode at run time.  Straight-line code of this size costs nothing on
// gfx950 (tools/ubench/icache.hip: 256 KB unrolled = 5.1 ticks / instruction vs 4.7 in a loop).
//
// Reference semantics: TFL_SIGNITURE.nutls_lstm, dnn_model/converter_proposed.py:188-867; blocks
// dnn_model/models/proposed.py:162-282.
//
// Arithmetic of the convs: fp32 results on the bf16 matrix pipe.  The weights are int8 (exact in bf16); every fp32
// activation is split ERROR-FREE into three bf16 pieces, x = hi + mid + lo (hi = the top 16 bits of x, mid = the top 16
// bits of x - hi, lo = the rest: 8 + 8 + 8 significand bits), so a conv is three bf16 MFMAs per tile and K step whose
// products w * piece are exact in fp32 and are accumulated in fp32 -- the same quantity an fp32 FMA chain rounds, with
// fewer roundings (tools/ubench/bf16x3.hip: 2.2e-7 relative rms against 3.7e-7 for v_mfma_f32_32x32x2_f32), at 16/3
// of the rate of the fp32 MFMA.  Everything else (scale, bias, LayerNorm, PReLU, LSTM, CTFA) is fp32 on the VALU.
//
// Data flow of a conv op I (inconv / strided conv / sub-pixel conv / down / up, proposed.py:198-265):
//   * its B operand is an LDS image [time tap][row][plane hi | mid | lo][channel] of bf16 with zero halo rows (the two
//     largest images stay fp32 and are split when they are read); the op BEFORE it completes that image: the rows it
//     produces go there straight from registers (split in its epilogue), everything else (previous-frame tap = the other
//     parity of the state tensor in HBM, skip-connection channels) is loaded from HBM into registers one or two ops
//     ahead and split + stored into the image between the two barriers of the op before;
//   * the strided convs (OpD::ys) keep only the CURRENT frame in their image: y_t = W[tap 1] x_t + S_{t-1}, where S_t = W[tap 0] x_t is
//     computed by the same op from the same B fragments and handed to the next frame through HBM as P x 32 fp32 sums (2-8x
//     fewer bytes than the 2P x cin input rows, no staging of them).  Their conv-input state tensors are still written -- they are
//     the ABI's states -- but never read here; the host rebuilds S after nutls_state_set / a step of another mode (engine.cpp
//     ysum_refresh).  The plans of this round have no two-round image left (the machinery for one -- Part::round2 -- stays);
//   * its A operand (int8 weights in MFMA fragment order, one blob in plan order) streams from L2 through a register
//     ring whose first fill is issued two ops ahead, converted to bf16 pairs in the MFMA shadow;
//   * large layers: 32x32x16 bf16 MFMA tiles, each wave owns whole LayerNorm groups, epilogue in registers;
//     small layers (<= 64 positions): 16x16x32 tiles, K split over the waves, partial tiles meet in an LDS
//     exchange buffer and a row-wise epilogue finishes them (LayerNorm over channels with DPP exchanges);
//   * the epilogue writes the state tensor (HBM, `cur` parity, fp32) and the next image (LDS).
// Barriers order LDS only; the memory counter is never drained except at the plan's drain points (at every LSTM and
// CTFA), which is what makes same-frame HBM hand-offs of skip connections safe.
#include <hip/hip_runtime.h>

#include <type_traits>
#include <utility>

#include "nutls_internal.hpp"
#include "fused_plan.hpp"

#ifndef FZ_PROF
#define FZ_PROF 0
#endif
// FZ_STOPAT (fused_step_stop.hip, the library's "stop twin" of the one-stream kernel; experiment builds of tools/exp): FzTa::skew is read as "end the
// launch in front of op <skew>" -- the production instruction stream plus one scalar compare per op, no stamps: the launch time as a function of that op
// index, differenced, is what each op costs the UN-instrumented kernel (nutls_profile_production; DESIGN.md section 4 "Round 6").
#ifndef FZ_STOPAT
#define FZ_STOPAT 0
#endif
// FZ_BASE = 1: the baseline variant's build (fused_base.hip): same ops, the LSTM + Dense bottlenecks replaced by the
// dilated-dense blocks of models/nunet_tls.py:277-359 (plan fused_plan_base.inc)
#ifndef FZ_BASE
#define FZ_BASE 0
#endif
// FZ_STREAMS = 2 / 4: the packed builds (fused_step_g2.hip / _g4.hip, plans fused_plan_lstm_g2.inc / _g4.inc): a workgroup owns that many
// consecutive streams -- the layers whose images fit LDS that often run them side by side on one virtual position axis (one weight
// fetch and conversion for all of them, full 16-position tiles in the deep layers), the others once per stream (fused_plan.hpp OpD::gs)
#ifndef FZ_STREAMS
#define FZ_STREAMS 1
#endif
#if FZ_BASE && FZ_STREAMS != 1
#error "packed plans exist for the LSTM variant only"
#endif

namespace nutls {
namespace fz {

#if FZ_BASE
#include "fused_plan_base.inc"
#elif FZ_STREAMS == 2
#include "fused_plan_lstm_g2.inc"
#elif FZ_STREAMS == 4
#include "fused_plan_lstm_g4.inc"
#else
#include "fused_plan_lstm.inc"
#endif
static_assert(kStreams == FZ_STREAMS, "plan and build disagree on the streams per workgroup");

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef const char __attribute__((address_space(1))) * gcb_t;
typedef char __attribute__((address_space(1))) * gb_t;
typedef const f32x4 __attribute__((address_space(1))) * gc4_t;
typedef f32x4 __attribute__((address_space(1))) * g4_t;
typedef const float __attribute__((address_space(1))) * gcf_t;
typedef float __attribute__((address_space(1))) * gf_t;

constexpr int THREADS = 512;
#define FZ_LN_EPS 1e-8f
// Packed plans (kStreams > 1, fused_plan.hpp OpD::gs / g0): a workgroup owns kStreams consecutive streams.  Everything below is written
// over "stream slot g of the workgroup"; in a one-stream plan every such term is the constant 0 and folds away.
constexpr int NSTREAMS = kStreams;

struct Ctx {
  gcb_t sbp, sbc, sbs, wb;     // `prev` / `cur` state bases and arena slice base of the workgroup's FIRST stream; weight blob
  gcb_t ysr, ysw;              // carried partial sums of the two-tap convs: last frame's block (read), this frame's (written)
  gcf_t io_in;                 // the first stream's 256 input magnitudes (stream g: + 256 g floats)
  gf_t io_out;
  unsigned sstride_b;          // packed plans: bytes between the arena slices of consecutive streams
  // CTFA frequency branch (FzTa): what is added to the frame's time attention before the / 32, and where the time attention goes
  gcb_t ta_sum, ta_ring;
  int ta_sum_sstride, ta_sum_gstride, ta_ring_sstride, ta_ring_gstride;
  unsigned long long* prof;
  const DdbParams* ddb;        // baseline variant: the 13 dilated-dense blocks of this step's parity
  int stream, step;            // step: frame counter (position in the dilated-dense history rings)
  int eager;                   // write the state tensors nothing here reads too (OpD::d0_on = 2; FzTa::eager)
  gcb_t dbg;                   // profiling builds: activation trace of this workgroup's first stream (FzTa::dbg), null = off
  unsigned dbg_sstride_b;
  int stop_at;                 // FZ_STOPAT builds (the stop twin, tools/exp/prod_timeline.py): the launch ends in front of this op
};
// Activation trace (profiling builds, nutls_debug_trace): the tensors the fused kernel keeps in LDS -- input layer, CTFA outputs, up-sampling
// outputs -- are also copied to the trace buffer, slot `slot`, row-major [row][ld floats].  Compiled out of the production kernels.
#define FZ_TRACE4(cx, g, slot, row, ld, c, v) do { if constexpr (FZ_PROF != 0) { if ((cx).dbg) \
    stb((cx).dbg, static_cast<unsigned>((g) * (cx).dbg_sstride_b) + static_cast<unsigned>(((slot) * kDbgSlotFloats + (row) * (ld) + (c)) * 4), (v)); } } while (0)
// does the launch write destination 0 of op d?  (d0_on 2: a conv-input state the kernel never reads -- only when the handle wants eager states)
#define FZ_D0(d, cx) ((d).d0_on == 1 || ((d).d0_on == 2 && (cx).eager != 0))
#define FZ_D1(d, cx) ((d).d1_on == 1 || ((d).d1_on == 2 && (cx).eager != 0))
// byte offset of stream slot g's arena slice relative to the workgroup's first stream (added to the 32-bit offset of a load / store)
__device__ __forceinline__ unsigned gofs(const Ctx& cx, int g) {
  if constexpr (NSTREAMS == 1) return 0u;
  else return static_cast<unsigned>(g) * cx.sstride_b;
}

// profiling build: phase stamps inside an op (wave 0 of workgroup 0), slot k of op I at prof[kNumOps + 1 + 8 I + k]
#define FZ_STAMP(I, k) do { if (FZ_PROF && cx.prof && tid == 0) cx.prof[kNumOps + 1 + 8 * (I) + (k)] = wall_clock64(); } while (0)
// FZ_WTRACE (profiling twin only, tools/gpu_wave_trace.py): EVERY wave of workgroup 0 stamps the shader clock (s_memtime, one tick per
// cycle) at the phase boundaries of every op: slot k of op I of wave w at prof[9 kNumOps + 1 + (w kNumOps + I) 12 + k].  The stamps cost
// a scalar memory round trip each -- the trace shows who waits for whom, not the un-instrumented step time.
#ifndef FZ_WTRACE
#define FZ_WTRACE 0
#endif
#define FZ_WSTAMP(I, k) do { if (FZ_PROF && FZ_WTRACE && cx.prof && (tid & 63) == 0) \
    cx.prof[9 * kNumOps + 1 + ((tid >> 6) * kNumOps + (I)) * 12 + (k)] = __builtin_readcyclecounter(); } while (0)

// ---- memory helpers (byte offsets) ---------------------------------------------------------------
__device__ __forceinline__ f32x4 ldb(gcb_t base, unsigned boff) { return *(gc4_t)(base + static_cast<unsigned long long>(boff)); }
__device__ __forceinline__ float ldb1(gcb_t base, unsigned boff) { return *(gcf_t)(base + static_cast<unsigned long long>(boff)); }
__device__ __forceinline__ void stb(gcb_t base, unsigned boff, f32x4 v) { *(g4_t)((gb_t)(unsigned long long)base + static_cast<unsigned long long>(boff)) = v; }
__device__ __forceinline__ void stb1(gcb_t base, unsigned boff, float v) { *(gf_t)((gb_t)(unsigned long long)base + static_cast<unsigned long long>(boff)) = v; }
// Cache policy (round 6; profiles/r06_dev_log.txt "nt"): every ACTIVATION is read exactly once per launch, by the one workgroup that owns the stream -- the
// previous frame's state rows and carried sums (written a whole launch ago), this frame's skip / residual / up-sampling rows -- while the WEIGHTS are read by
// all 32 workgroups of an XCD within microseconds of each other and again next launch.  With plain loads the 2 MB of activations a stream pulls through the
// 4 MB L2 of its XCD per frame (x 32 streams) evict the 3.4 MB of weights and each other; with the non-temporal hint on the activation loads (and on the
// stores whose rows only the NEXT launch reads: the carried sums, OpD::nt0 / nt1 destinations) the step at 256 streams fell from 0.348 to 0.301 ms on one box
// -- most of what the memory system cost between 160 and 256 streams.  NOT on the stores that are re-read in the same frame (skip copies, residual rows):
// with those non-temporal too the step is 0.330.
__device__ __forceinline__ f32x4 ld_once(gcb_t base, unsigned boff) { return __builtin_nontemporal_load((gc4_t)(base + static_cast<unsigned long long>(boff))); }
__device__ __forceinline__ float ld_once1(gcb_t base, unsigned boff) { return __builtin_nontemporal_load((gcf_t)(base + static_cast<unsigned long long>(boff))); }
__device__ __forceinline__ void st_next(gcb_t base, unsigned boff, f32x4 v) { __builtin_nontemporal_store(v, (g4_t)((gb_t)(unsigned long long)base + static_cast<unsigned long long>(boff))); }
__device__ __forceinline__ void st_next1(gcb_t base, unsigned boff, float v) { __builtin_nontemporal_store(v, (gf_t)((gb_t)(unsigned long long)base + static_cast<unsigned long long>(boff))); }
// a state store: non-temporal when nothing in this launch reads the rows again (OpD::nt0 / nt1)
template <bool NT> __device__ __forceinline__ void st_state(gcb_t base, unsigned boff, f32x4 v) { if constexpr (NT) st_next(base, boff, v); else stb(base, boff, v); }
template <bool NT> __device__ __forceinline__ void st_state1(gcb_t base, unsigned boff, float v) { if constexpr (NT) st_next1(base, boff, v); else stb1(base, boff, v); }
extern __shared__ __attribute__((aligned(16))) float lds[];
__device__ __forceinline__ f32x4& lds4(int boff) { return *reinterpret_cast<f32x4*>(reinterpret_cast<char*>(lds) + boff); }
__device__ __forceinline__ float& lds1(int boff) { return *reinterpret_cast<float*>(reinterpret_cast<char*>(lds) + boff); }
__device__ __forceinline__ u32x2& lds2u(int boff) { return *reinterpret_cast<u32x2*>(reinterpret_cast<char*>(lds) + boff); }
__device__ __forceinline__ unsigned short& lds_h(int boff) { return *reinterpret_cast<unsigned short*>(reinterpret_cast<char*>(lds) + boff); }
// workgroup barrier that orders LDS traffic only (global loads / stores stay in flight across it)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ void drain_vm() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

template <class F, int... Is>
__device__ __forceinline__ void sfor_impl(F&& f, std::integer_sequence<int, Is...>) { (f(std::integral_constant<int, Is>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void sfor(F&& f) { sfor_impl(f, std::make_integer_sequence<int, N>{}); }

#define FZ_LIKELY(x) __builtin_expect(!!(x), 1)
// nothing is scheduled across this point: keeps asynchronous loads where they were written (the machine scheduler
// otherwise sinks a load down to its first use to shorten the live range -- and the wave then waits for it on the spot)
__device__ __forceinline__ void sched_pin() { __builtin_amdgcn_sched_barrier(0); }

constexpr int cmax(int a, int b) { return a > b ? a : b; }
constexpr int cmin(int a, int b) { return a < b ? a : b; }
constexpr int clog2(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }
constexpr int img_row_b(int pitch_b, int pair, int half_b, int lr) { return pair ? (lr >> 1) * pitch_b + (lr & 1) * half_b : lr * pitch_b; }
__device__ __forceinline__ int img_row_rt(int pitch_b, int pair, int half_b, int lr) { return pair ? (lr >> 1) * pitch_b + (lr & 1) * half_b : lr * pitch_b; }

// ---- DPP lane exchanges -----------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
// s + (the value 16 / 32 lanes away): gfx950's v_permlane16_swap / v_permlane32_swap give both halves of the exchange in two registers
// (one VALU instruction; __shfl_xor is a ds_bpermute: an LDS-crossbar round trip in the middle of a dependent chain)
__device__ __forceinline__ float xor16_sum(float s) {
  const int v = __builtin_bit_cast(int, s);
  const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);      // r[0]: rows 0 0 2 2, r[1]: rows 1 1 3 3
  return __builtin_bit_cast(float, static_cast<int>(r[0])) + __builtin_bit_cast(float, static_cast<int>(r[1]));
}
__device__ __forceinline__ float xor32_sum(float s) {
  const int v = __builtin_bit_cast(int, s);
  const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);      // r[0]: lanes 0..31 twice, r[1]: lanes 32..63 twice
  return __builtin_bit_cast(float, static_cast<int>(r[0])) + __builtin_bit_cast(float, static_cast<int>(r[1]));
}
template <int LPG>
__device__ __forceinline__ float group_sum(float s) {      // sum over LPG consecutive lanes (aligned), result in all of them
  s += dpp_mov<0xB1>(s);                  // quad_perm [1,0,3,2]
  s += dpp_mov<0x4E>(s);                  // quad_perm [2,3,0,1]
  if (LPG >= 8) s += dpp_mov<0x141>(s);   // row_half_mirror
  if (LPG >= 16) s += dpp_mov<0x140>(s);  // row_mirror
  if (LPG >= 32) s = xor16_sum(s);
  if (LPG >= 64) s = xor32_sum(s);
  return s;
}
__device__ __forceinline__ float fast_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float fast_tanh(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * x)); }

// ---- error-free split of fp32 into three bf16 pieces, and the image accessors built on it -------------------------------
// x = hi + mid + lo exactly: hi = the top 16 bits of x (a bf16: 8 significand bits), mid = the top 16 bits of the exact
// remainder x - hi (<= 16 significand bits), lo = what is left (<= 8 bits, so its fp32 pattern has 16 zero low bits).
__device__ __forceinline__ unsigned fbits(float x) { return __builtin_bit_cast(unsigned, x); }
__device__ __forceinline__ float bitsf(unsigned b) { return __builtin_bit_cast(float, b); }
// dword of two bf16: low half = the top 16 bits of a, high half = the top 16 bits of b  (v_perm_b32)
__device__ __forceinline__ unsigned pk_top16(unsigned a, unsigned b) { return __builtin_amdgcn_perm(b, a, 0x07060302u); }
__device__ __forceinline__ void split3(float x, unsigned& xb, unsigned& rb, unsigned& lb) {
  xb = fbits(x);
  const float r = x - bitsf(xb & 0xffff0000u);
  rb = fbits(r);
  lb = fbits(r - bitsf(rb & 0xffff0000u));
}
struct Pieces4 { u32x2 hi, mid, lo; };       // 4 consecutive channels as 4 bf16 per plane
__device__ __forceinline__ Pieces4 split4(const f32x4& v) {
  unsigned xb[4], rb[4], lb[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) { const float x = v[e]; split3(x, xb[e], rb[e], lb[e]); }
  Pieces4 o;
  o.hi = u32x2{pk_top16(xb[0], xb[1]), pk_top16(xb[2], xb[3])};
  o.mid = u32x2{pk_top16(rb[0], rb[1]), pk_top16(rb[2], rb[3])};
  o.lo = u32x2{pk_top16(lb[0], lb[1]), pk_top16(lb[2], lb[3])};
  return o;
}
__device__ __forceinline__ float join3(unsigned h16, unsigned m16, unsigned l16) {      // pieces in the TOP halves of the arguments
  return (bitsf(h16) + bitsf(m16)) + bitsf(l16);      // (exact: 16 bits, then 24 bits)
}
// Image accessors: `a` = LDS byte address of (row, first channel) -- in the hi plane of a three-plane image (FMT 1)
template <int FMT>
__device__ __forceinline__ void img_st4(int a, int plane_b, const f32x4& v) {
  if constexpr (FMT == 0) {
    lds4(a) = v;
  } else {
    const Pieces4 q = split4(v);
    lds2u(a) = q.hi;
    lds2u(a + plane_b) = q.mid;
    lds2u(a + 2 * plane_b) = q.lo;
  }
}
template <int FMT>
__device__ __forceinline__ f32x4 img_ld4(int a, int plane_b) {
  if constexpr (FMT == 0) {
    return lds4(a);
  } else {
    const u32x2 h = lds2u(a), m = lds2u(a + plane_b), l = lds2u(a + 2 * plane_b);
    f32x4 v;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const unsigned hh = h[e], mm = m[e], ll = l[e];
      v[2 * e] = join3(hh << 16, mm << 16, ll << 16);
      v[2 * e + 1] = join3(hh & 0xffff0000u, mm & 0xffff0000u, ll & 0xffff0000u);
    }
    return v;
  }
}
template <int FMT>
__device__ __forceinline__ void img_st1(int a, int plane_b, float x) {
  if constexpr (FMT == 0) {
    lds1(a) = x;
  } else {
    unsigned xb, rb, lb;
    split3(x, xb, rb, lb);
    lds_h(a) = static_cast<unsigned short>(xb >> 16);
    lds_h(a + plane_b) = static_cast<unsigned short>(rb >> 16);
    lds_h(a + 2 * plane_b) = static_cast<unsigned short>(lb >> 16);
  }
}
constexpr int esz_of(int fmt) { return fmt ? 2 : 4; }      // bytes per channel inside a row (plane)

// ---- static facts about an op ------------------------------------------------------------------------
constexpr bool is_up(const OpD& d) { return d.kind == K_UP; }
constexpr int ntot(const OpD& d) { return d.N * (is_up(d) ? 2 : 1); }
constexpr int ksl(const OpD& d) { return d.KSt * d.KSg; }                                        // K slices (X16B)
constexpr int kgroups(const OpD& d) { return d.cin / (d.path == P_R32B ? 16 : 32); }            // K steps (one MFMA deep) per segment
constexpr int gw(const OpD& d) { return kgroups(d) / d.KSg; }                                     // ... per wave
// two-tap convs (d.ys): the image holds the current frame only; K segments 0..2 carry the weights of time tap 0 (their sums go
// to the NEXT frame), 3..5 those of tap 1 (this frame).  16x16 tiles: the K slice's ks_t selects the tap (KSt 2), or a wave owns
// both (KSt 1, two accumulator sets); 32x32 tiles: waves 0..3 own tap 0, waves 4..7 tap 1 (r32_two).
constexpr bool r32_two(const OpD& d) { return d.path == P_R32B && d.ys != 0; }
constexpr bool x16_both(const OpD& d) { return d.path == P_X16B && d.ys != 0 && d.KSt == 1; }
constexpr int segw(const OpD& d) { return is_up(d) ? 3 : (r32_two(d) ? 3 : d.nseg / d.KSt); }     // segments per wave
// weight fragments (one A operand: the 8 K values of a lane) per wave
constexpr int conv_nf(const OpD& d) { return segw(d) * gw(d) * d.NT; }
constexpr int ntask(const OpD& d) { return d.PG * d.CG * d.KSt * d.KSg * (r32_two(d) ? 2 : 1); }
constexpr int ex_slices(const OpD& d) { return ksl(d) * (x16_both(d) ? 2 : 1); }                  // slices of the exchange buffer (X16B)
constexpr int ex_group(const OpD& d) { return d.ys ? ex_slices(d) / 2 : ex_slices(d); }           // ... that add up to one result
constexpr int nparams(const OpD& d) { return 2 * ntot(d) + 2 * d.gc + 1; }      // bias | weight scale | gamma | beta | alpha
constexpr int conv_nsf(const OpD& d) { return (conv_nf(d) + 1) / 2; }             // "super-fragments": 2 int8 fragments = one dwordx4 per lane
// The epilogue parameter block travels from L2 to LDS through one register per thread.  A vector load costs the CU's memory pipe by its WIDTH,
// whatever its lanes fetch (tools/ubench/ta_cost.hip, profiles/r06_ubench_ta_cost.txt: a dwordx4 wave-load 14-18 cycles also when 63 lanes
// re-fetch the last item, a dword wave-load 5.7), and every wave issues every load of a prefetch (no branch around a load): with at most 512
// parameters each thread loads ONE float -- 8 dword wave-loads instead of 8 dwordx4 of which one or two carry anything.
constexpr bool prm_dword(const OpD& d) { return nparams(d) <= THREADS; }
constexpr int ring_sf(const OpD& d) { return cmin(conv_nsf(d), d.path == P_R32B ? RING_SF_R32 : RING_SF); }
// staging classes of a part: 1 loaded and stored by the op that builds the image, 2 loaded one op earlier (carried); second-round parts
// of a two-round image (stored in the middle of the op that OWNS the image): 3 loaded at the start of that op, 4 loaded one op earlier
constexpr int part_cls(const Part& p) { return p.round2 ? (p.la == 2 ? 4 : 3) : p.la; }
// Who stages: the parts an op loads AND stores itself (classes 1 and 3) are handled by its "stager" threads -- all 512, or, in the
// large-layer ops whose tiling leaves waves 4..7 without MFMA work, only those 256: the MFMA waves' weight refills then do not
// queue behind the staging loads on the (in-order) memory counter, and the staging waves' waits cost the MFMA waves nothing.
// Class-2 parts (loaded one op before they are stored) stay with all threads: both ops must agree on who holds what.
constexpr int stg_threads(int i) {
  return (i >= 0 && i < kNumOps && kOps[i].type == T_CONV && kOps[i].path == P_R32B && kOps[i].PG * kOps[i].CG == 4 && !kOps[i].ys) ? 256 : THREADS;
}
constexpr int part_items(const Part& p) { return p.ng * p.rows * p.c4s; }      // (packed plans: the block once per stream)
constexpr int part_n(const Part& p, int nthr) { return (part_items(p) + nthr - 1) / nthr; }
constexpr int parts_regs(const Img& g, int cls, int nthr) {
  int n = 0;
  for (int k = 0; k < g.nparts; ++k) if (part_cls(g.parts[k]) == cls) n += part_n(g.parts[k], nthr);
  return n;
}
constexpr int part_base(const Img& g, int cls, int k, int nthr) {
  int n = 0;
  for (int kk = 0; kk < k; ++kk) if (part_cls(g.parts[kk]) == cls) n += part_n(g.parts[kk], nthr);
  return n;
}
constexpr int nxt_of(int i) { return (i >= 0 && i < kNumOps) ? kOps[i].nxt : -1; }
constexpr int nxt_regs(int i, int cls) { return nxt_of(i) >= 0 ? parts_regs(kOps[nxt_of(i)].img, cls, cls == 1 ? stg_threads(i) : THREADS) : 0; }
constexpr int own_regs(int i, int cls) { return (i < kNumOps && kOps[i].type == T_CONV) ? parts_regs(kOps[i].img, cls, cls == 4 ? THREADS : stg_threads(i)) : 0; }
constexpr int ctfa_ni(const OpD& d) { return (d.F + 31) / 32; }
// registers of last frame's partial sums an op adds in its epilogue: row-wise epilogue: one float4 per item and pass; 32x32 tiles: the
// accumulator layout (4 float4 per tile)
constexpr int yp_regs(int i) {
  if (i < 0 || i >= kNumOps || kOps[i].type != T_CONV || !kOps[i].ys) return 0;
  const OpD& d = kOps[i];
  return d.path == P_R32B ? d.PT * d.NT * 4 : (d.epl == 1 ? 1 : (d.gs * d.P * ntot(d) / 4 + THREADS - 1) / THREADS);      // (epl 1: one float per thread)
}
constexpr int lstm_s0(const OpD& d) { return lstm_nrp(d.din) / 4; }      // carry slots of the gate weights: NRP int8x4 rows = NRP / 4 float4 (see lstm_op)
constexpr int carry_w(int i) {
  if (i >= kNumOps) return 0;
  const OpD& d = kOps[i];
  if (d.type == T_CONV) return ring_sf(d);
  if (d.type == T_LSTM) return lstm_s0(d) + 10 + 2 * (d.gs - 1);      // (packed plans: the h values (2 slots) of every further stream)
  if (d.type == T_CTFA) return d.gs * ctfa_ni(d) + 2;
#if FZ_BASE
  if (d.type == T_DDB) {
    if (d.x_cols == 64) return DdbzCarry<THREADS, 32, 4>::N;
    const int F = d.din / d.x_cols;
    return F == 4 ? DdbzCarry<THREADS, 16, 4>::N : (F == 2 ? DdbzCarry<THREADS, 16, 2>::N : DdbzCarry<THREADS, 16, 1>::N);
  }
#endif
  return 0;
}

// Large-layer ops with a few more weight fragments than the carried ring holds: the rest is requested at the very start of the
// op, ahead of its staging loads, and the MFMA loop runs without refills (a refill would be YOUNGER than the staging loads on
// the in-order memory counter: waiting for it means waiting for the whole staging burst -- seen as `vmcnt(0)` in mid-loop).
constexpr int EXT_MAX = 5;
constexpr int ext_sf(int i) {
  if (i < 0 || i >= kNumOps || kOps[i].type != T_CONV || kOps[i].path != P_R32B) return 0;
  const int extra = conv_nsf(kOps[i]) - ring_sf(kOps[i]);
  return (extra > 0 && extra <= EXT_MAX) ? extra : 0;
}

// CTFA column sums in the epilogue of the stage's last sub-pixel conv (converter_proposed.py:257-263: the CTFA averages that conv's output
// over frequency): the row-wise epilogue has every output row in registers, so each thread adds its rows up, the four lanes of a wave that
// hold the same channel quad meet through two lane swaps, and every wave leaves one partial row in the CTFA's scratch (above the
// epilogue's own parameter block).  The CTFA op then starts at its gate perceptrons: no pass over the image to form the sums, one
// barrier less.  One-stream plans, 16x16-tile producers (the two 32x32-tile ones would need 160 lane-exchange steps per wave).
constexpr int CSUM_OFF_B = 1792;       // inside the CTFA's scratch: 8 waves x 64 floats
constexpr bool feeds_ctfa_sums(int i) {
  return NSTREAMS == 1 && i >= 0 && i + 1 < kNumOps && kOps[i + 1].type == T_CTFA && kOps[i].type == T_CONV && kOps[i].path == P_X16B &&
         kOps[i].gc == 64 && nparams(kOps[i]) * 4 <= CSUM_OFF_B && (kOps[i].epl != 1 || kOps[i].P * ntot(kOps[i]) == THREADS);
}
// Such a CTFA op has nothing left to hide the fetch of its gate perceptrons behind (the column-sum pass did): they are requested at the
// END of the conv op before it -- late enough to cost that op's MFMA loop no registers -- and travel in the Carry.
constexpr bool early_gates(int i) { return i >= 1 && i < kNumOps && kOps[i].type == T_CTFA && feeds_ctfa_sums(i - 1); }
// What travels in registers from op I-1 to op I: the first weight fragments of op I (or the LSTM's input
// weights / the CTFA's residual rows and gate matrices) and the far-ahead staged parts of the image op I completes.
template <int I>
struct Carry {
  f32x4 w[cmax(1, carry_w(I))];
  f32x4 cg[early_gates(I) ? 17 : 1];      // CTFA whose column sums come with the rows: its gate perceptrons (16 float4 of the wave that evaluates them + biases), requested at the END of the op before
  f32x4 p[cmax(1, nxt_regs(I, 2))];
  f32x4 p4[cmax(1, own_regs(I, 4))];      // the previous-frame tap of op I's own two-round image, requested by op I-1 (all threads hold it)
  f32x4 yp[cmax(1, yp_regs(I))];          // last frame's partial sums of op I (two-tap convs), requested two ops ahead like the weights
  f32x4 yp2[cmax(1, yp_regs(I + 1))];
  f32x4 prm;                   // conv ops: this thread's float4 of the epilogue parameter block (bias | scale | gamma | beta | alpha)
  // the same for op I+1, already requested by op I-1: weights are fetched TWO ops ahead (a fetch that misses L2 -- the
  // state tensors stream through it all the time -- takes longer than one small op)
  f32x4 w2[cmax(1, carry_w(I + 1))];
  f32x4 prm2;
};

// ---- staging: HBM tensor blocks -> registers -> LDS image -------------------------------------------------
// (tid: index among the NTHR staging threads)
// Who holds which item: the items of a part are dealt to the threads from the TOP of the workgroup downwards, and every
// part starts one wavefront below the end of the part before it.  Between the two barriers of a small conv op the row-wise
// epilogue keeps the LOWEST waves busy (a few dozen rows); the stores of the staged parts and the halo zeroing (middle
// waves) then run beside it on other SIMDs instead of queueing behind it on wave 0.
constexpr int part_shift(const Img& g, int cls, int k, int nthr) {
  int sh = 0;
  for (int kk = 0; kk < k; ++kk)
    if (part_cls(g.parts[kk]) == cls) sh += (cmin(part_items(g.parts[kk]), nthr) + 63) / 64 * 64;
  return sh % nthr;
}
template <int NTHR>
__device__ __forceinline__ int stage_slot(int tid, int shift) { return (2 * NTHR - 1 - tid - shift) & (NTHR - 1); }
template <int J, int CLS, int NTHR, int NR>
__device__ __forceinline__ void stage_load(const Ctx& cx, int tid, f32x4 (&r)[NR]) {
  if constexpr (J >= 0 && J < kNumOps) {
    constexpr Img g = kOps[J].img;
    sfor<g.nparts>([&](auto kk) {
      constexpr int K = decltype(kk)::value;
      constexpr Part p = g.parts[K];
      if constexpr (part_cls(p) == CLS) {
        constexpr int items = part_items(p), per = p.rows * p.c4s, base = part_base(g, CLS, K, NTHR), cs = clog2(p.c4s), NG = p.ng, PG0 = p.g0;
        const gcb_t src = p.src == S_PREV ? cx.sbp : (p.src == S_CUR ? cx.sbc : cx.sbs);
        const int slot = stage_slot<NTHR>(tid, part_shift(g, CLS, K, NTHR));
        sfor<part_n(p, NTHR)>([&](auto ii) {
          constexpr int i = decltype(ii)::value;
          int q = slot + NTHR * i;
          if ((i + 1) * NTHR > items) q = q < items ? q : items - 1;      // lanes past the end re-load the last item
          const int gi = NG > 1 ? q >> clog2(per) : 0;                     // stream of the item (packed plans)
          if constexpr (NG > 1) q &= per - 1;
          const int row = q >> cs, c4 = q & (p.c4s - 1);
          r[base + i] = ld_once(src, static_cast<unsigned>(p.off * 4 + row * (p.ld * 4) + c4 * 16) + gofs(cx, PG0 + gi));
        });
      }
    });
  }
}
template <int J, int CLS, int NTHR, int NR>
__device__ __forceinline__ void stage_store(int tid, const f32x4 (&r)[NR]) {
  if constexpr (J >= 0 && J < kNumOps) {
    constexpr Img g = kOps[J].img;
    sfor<g.nparts>([&](auto kk) {
      constexpr int K = decltype(kk)::value;
      constexpr Part p = g.parts[K];
      if constexpr (part_cls(p) == CLS) {
        constexpr int items = part_items(p), per = p.rows * p.c4s, base = part_base(g, CLS, K, NTHR), cs = clog2(p.c4s), NG = p.ng;
        const int slot = stage_slot<NTHR>(tid, part_shift(g, CLS, K, NTHR));
        sfor<part_n(p, NTHR)>([&](auto ii) {
          constexpr int i = decltype(ii)::value;
          const int q = slot + NTHR * i;
          const int gi = NG > 1 ? q >> clog2(per) : 0, ql = NG > 1 ? q & (per - 1) : q;
          const int row = ql >> cs, c4 = ql & (p.c4s - 1);
          const int a = p.lds_b + gi * p.gstride_b + img_row_rt(g.pitch_b, g.pair, g.half_b, p.row0 + row) + c4 * (4 * esz_of(g.fmt));
          if ((i + 1) * NTHR <= items || FZ_LIKELY(q < items)) img_st4<g.fmt>(a, g.plane_b, r[base + i]);
        });
      }
    });
  }
}
// halo blocks (float4 units) of image J: block K is zeroed by threads [t0_K, t0_K + n4_K), one float4 per thread, ONE masked
// store for all blocks (the block a thread belongs to is found with a compare + select per block: a branch per block cost
// 0.4 us per op).  The halo threads start at wave 4: below them the row-wise epilogue, above them the staged parts.
constexpr int zero_total(const Img& g) { int n = 0; for (int k = 0; k < g.nzero; ++k) n += g.zero[k].n4; return n; }
constexpr int zero_first(const Img& g) { return cmin(THREADS / 2, THREADS - zero_total(g)); }
constexpr int zero_start(const Img& g, int k) { int n = zero_first(g); for (int kk = 0; kk < k; ++kk) n += g.zero[kk].n4; return n; }
template <int J>
__device__ __forceinline__ void zero_halos(int tid) {
  if constexpr (J >= 0 && J < kNumOps) {
    constexpr Img g = kOps[J].img;
    static_assert(zero_total(g) <= THREADS, "halo blocks: one float4 per thread");
    if constexpr (g.nzero > 0) {
      int delta = g.zero[0].lds_b - zero_start(g, 0) * 16;        // LDS address of thread t's float4 = delta + 16 t
      sfor<g.nzero - 1>([&](auto kk) {
        constexpr int K = decltype(kk)::value + 1;
        delta = tid >= zero_start(g, K) ? g.zero[K].lds_b - zero_start(g, K) * 16 : delta;
      });
      float zf = 0.f;
      asm volatile("" : "+v"(zf));
      const f32x4 z = {zf, zf, zf, zf};
      if (static_cast<unsigned>(tid - zero_first(g)) < static_cast<unsigned>(zero_total(g))) {
        lds4(delta + tid * 16) = z;
        sfor<kOps[J].gs - 1>([&](auto gg) { lds4(delta + tid * 16 + (decltype(gg)::value + 1) * g.gstride_b) = z; });      // (packed plans: every stream's sub-image)
      }
    }
  }
}

// A operand of the bf16 MFMAs: fragment `fr` (0 / 1) of a super-fragment = 8 int8 weights of this lane (exact in bf16) -> four
// dwords of two bf16 each; the per-channel scale of the quantisation is applied to the accumulators in the epilogue.
// (The blob stays int8: with the weights widened to bf16 on the host the conversion disappears from the loops -- 12 VALU per
//  fragment -- but twice the bytes per op through L2 cost more than that: 0.385 -> 0.418 ms/step, DESIGN.md "tried and dropped".)
__device__ __forceinline__ bf16x8 wfrag(const f32x4& sfrag, int fr) {
  const float x0 = sfrag[2 * fr], x1 = sfrag[2 * fr + 1];     // (scalar copies first: __builtin_bit_cast applied to `vec[j]` directly reads element 0, ROCm 7.2 clang)
  const int v0 = __builtin_bit_cast(int, x0), v1 = __builtin_bit_cast(int, x1);
  u32x4 o;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int v = q < 2 ? v0 : v1;
    const float f0 = static_cast<float>(static_cast<signed char>(v >> (16 * (q & 1))));
    const float f1 = static_cast<float>(static_cast<signed char>(v >> (16 * (q & 1) + 8)));
    o[q] = pk_top16(fbits(f0), fbits(f1));
  }
  return __builtin_bit_cast(bf16x8, o);
}
__device__ __forceinline__ bf16x8 as_bf(const f32x4& v) { return __builtin_bit_cast(bf16x8, v); }
// B operand from an fp32 image: 8 consecutive channels -> the three bf16 fragments (split when read)
__device__ __forceinline__ void split8(const f32x4& x0, const f32x4& x1, bf16x8 (&b)[3]) {
  unsigned xb[8], rb[8], lb[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { const float x = e < 4 ? x0[e & 3] : x1[e & 3]; split3(x, xb[e], rb[e], lb[e]); }
  u32x4 h, m, l;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    h[q] = pk_top16(xb[2 * q], xb[2 * q + 1]);
    m[q] = pk_top16(rb[2 * q], rb[2 * q + 1]);
    l[q] = pk_top16(lb[2 * q], lb[2 * q + 1]);
  }
  b[0] = __builtin_bit_cast(bf16x8, h);
  b[1] = __builtin_bit_cast(bf16x8, m);
  b[2] = __builtin_bit_cast(bf16x8, l);
}
// The carried weights become visible to the optimiser only here: without this it hoists the byte extraction up to the
// loads in the previous op -- and waits for them there, which turns the prefetch into a synchronous load.
template <int N>
__device__ __forceinline__ void pin_regs(f32x4 (&r)[N]) {
#pragma unroll
  for (int i = 0; i < N; ++i) asm volatile("" : "+v"(r[i]));
}

// "Defined, contents unknown" without an instruction: registers only one wave-branch loads (gates_load) get an opaque definition on the other path,
// so that what is computed from them later stays a plain register phi (LLVM folds anything computed from phi(value, undef) into the defining block).
template <int N>
__device__ __forceinline__ void opaque_regs(f32x4 (&r)[N]) {
#pragma unroll
  for (int i = 0; i < N; ++i) asm volatile("" : "=v"(r[i]));
}

// ---- wave task of a conv op -----------------------------------------------------------------------------------
struct Task { int active, wbase_f, a, b, ks; };   // a: position group, b: channel group / tile, ks: K slice (X16B)
template <int I>
__device__ __forceinline__ Task conv_task(int wave) {
  constexpr OpD d = kOps[I];
  Task t;
  t.active = ntask(d) >= 8 ? 1 : wave < ntask(d);      // (a compile-time fact where all 8 waves have a task: no branch around their loads)
  if constexpr (d.path == P_R32B) {
    t.a = wave & (d.PG - 1);                    // position group
    t.b = (wave >> clog2(d.PG)) & (d.CG - 1);   // channel group
    t.ks = r32_two(d) ? ((wave >> clog2(d.PG * d.CG)) & 1) : 0;      // two-tap conv: 1 = this frame's sums (tap 1), 0 = the next frame's
    t.wbase_f = (t.ks * d.CG + t.b) * (conv_nsf(d) * 256);
  } else {
    // wave = ct + CG (ks + KS pg): the waves of one position group share the weights of (ct, ks)
    t.b = wave & (d.CG - 1);
    t.ks = (wave >> clog2(d.CG)) & (ksl(d) - 1);
    t.a = (wave >> clog2(d.CG * ksl(d))) & (d.PG - 1);
    t.wbase_f = (t.ks * d.CG + t.b) * (conv_nsf(d) * 256);
  }
  return t;
}

// ---- prefetch of what op I needs first (issued by op I-1) ---------------------------------------------------------
template <int I, int NW>
__device__ __forceinline__ void prefetch_w(const Ctx& cx, int tid, f32x4 (&w)[NW], f32x4& prm) {
  if constexpr (I < kNumOps) {
    constexpr OpD d = kOps[I];
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if constexpr (d.type == T_CONV) {
      // Every load of a prefetch is issued by EVERY thread (indices clamped, no branch around a load): the compiler counts
      // outstanding loads per control-flow path, and after a branch that holds loads it must assume the path without them --
      // a later wait for an OLDER load then also drains the ones just issued (seen in the disassembly as `vmcnt(1)` in front
      // of the first MFMA of a small op: one full memory round trip per op).
      const Task t = conv_task<I>(wave);
      // wave-uniform base (SGPR pair) + lane offset (VGPR) + immediate: no 64-bit vector address arithmetic
      // (a wave without a task fetches the fragments of task wave mod ntask and never uses them)
      const gcb_t wbase = cx.wb + static_cast<unsigned long long>(static_cast<unsigned>((d.w_off + t.wbase_f) * 4));
      sfor<carry_w(I)>([&](auto ff) {
        constexpr int sf = decltype(ff)::value;
        w[sf] = ldb(wbase + static_cast<unsigned long long>(sf * 1024), static_cast<unsigned>(lane * 16));
      });
      if constexpr (prm_dword(d)) {
        prm[0] = ldb1(cx.wb + static_cast<unsigned long long>(d.p_off * 4), static_cast<unsigned>((tid < nparams(d) ? tid : nparams(d) - 1) * 4));
      } else {
        constexpr int NP4 = (nparams(d) + 3) / 4;
        prm = ldb(cx.wb + static_cast<unsigned long long>(d.p_off * 4), static_cast<unsigned>((tid < NP4 ? tid : NP4 - 1) * 16));
      }
    } else if constexpr (d.type == T_LSTM) {
      // everything lstm_op needs from memory (slot map there), one op ahead; branch-free (see above): thread (u, sl) loads the NRP int8x4 rows
      // of its K slice as NRP / 4 float4 (fused_plan.hpp: blob layout of an LSTM op; sl >= 20 re-loads slice 19), the 6 values of h of an h
      // slice, every thread the Dense row of its output (clamped: 2 float4 int8 / 6 float4 fp32) and the record of its unit (clamped)
      constexpr int NRP = lstm_nrp(d.din), S0 = lstm_s0(d), WB = d.lw_off, REC = WB + lstm_gates_f(d.din), WD = REC + lstm_rec_f();
      constexpr int DROW = lstm_dense_row_f(d.dout);
      const int u = tid % 21, sl = tid / 21 < 20 ? tid / 21 : 19;
      const bool xs = sl < 16;
      const int hs = sl - 16;
      sfor<NRP / 4>([&](auto jj) {
        constexpr int j = decltype(jj)::value;
        w[j] = ldb(cx.wb, static_cast<unsigned>((WB + (sl * 21 + u) * NRP + 4 * j) * 4));
      });
      const int h0 = xs ? 0 : 6 * hs;
      sfor<6 * d.gs>([&](auto jj) {
        constexpr int j = decltype(jj)::value % 6, gi = decltype(jj)::value / 6, SH = gi == 0 ? S0 : S0 + 10 + 2 * (gi - 1);
        w[SH + j / 4][j % 4] = ld_once1(cx.sbp, static_cast<unsigned>((d.h_off + h0 + j) * 4) + gofs(cx, d.g0 + gi));      // (h[21..23]: slot padding, zero)
      });
      // (packed plans: gates and Dense outputs are dealt to threads as (stream, unit) / (stream, output): thread tid owns unit tid % 21 of
      //  stream tid / 21 and output tid % dout -- of stream (tid + 512 pass) / dout)
      const int drow = d.gs > 1 ? (tid & (d.dout - 1)) : (tid < d.dout ? tid : d.dout - 1);
      sfor<DROW / 4>([&](auto jj) {
        constexpr int j = decltype(jj)::value;
        w[S0 + 2 + j] = ldb(cx.wb, static_cast<unsigned>((WD + drow * DROW + 4 * j) * 4));
      });
      const int u21 = d.gs > 1 ? (tid < 21 * d.gs ? tid % 21 : 20) : (tid < 21 ? tid : 20);
      const int g21 = d.gs > 1 ? (tid < 21 * d.gs ? tid / 21 : d.gs - 1) : 0;          // the stream slot whose cell state this thread updates
      w[S0 + 8] = ldb(cx.wb, static_cast<unsigned>((REC + 4 * u21) * 4));
      w[S0 + 9][0] = ld_once1(cx.sbp, static_cast<unsigned>((d.c_off + u21) * 4) + gofs(cx, d.g0 + g21));
      w[S0 + 9][1] = ldb1(cx.wb, static_cast<unsigned>((REC + 84) * 4));      // s_x
      w[S0 + 9][2] = ldb1(cx.wb, static_cast<unsigned>((REC + 85) * 4));      // s_h
#if FZ_BASE
    } else if constexpr (d.type == T_DDB) {
      ddbz_prefetch<THREADS, d.x_cols / 2, d.din / d.x_cols>(ddbz_load_rec(cx.ddb + d.bidx), cx.stream, cx.step, tid, w);
#endif
    } else if constexpr (d.type == T_CTFA) {
      constexpr int NI = ctfa_ni(d);
      const int c4 = tid & 15, rg = tid >> 4;
      sfor<NI * d.gs>([&](auto ii) {
        constexpr int i = decltype(ii)::value % NI, gi = decltype(ii)::value / NI;
        int f = rg + 32 * i;
        if (f > d.F - 1) f = d.F - 1;
        w[gi * NI + i] = ld_once(cx.sbc, static_cast<unsigned>((d.e0_off + f * d.e0_ld + 4 * c4) * 4) + gofs(cx, d.g0 + gi));
      });
      if constexpr (d.last) {
        w[NI * d.gs] = ldb(cx.wb, static_cast<unsigned>((d.cw_off + 4256 + 4 * c4) * 4));                     // output conv weights
        w[NI * d.gs + 1][0] = ldb1(cx.wb, static_cast<unsigned>((d.cw_off + 4320) * 4));
      }
    }
  }
}

// HBM byte offset (relative to a block of partial sums of the workgroup's first stream) of the float4 a lane of a 32x32-tile two-tap
// conv keeps for (position tile pt of its task, channel tile n, accumulator quarter q) -- without the lane's own 16 bytes.  Packed
// plans: the task's virtual position tile belongs to one stream; inside a stream the layout is that of the one-stream plan.
template <int I>
__device__ __forceinline__ unsigned r32_ys_off(const Ctx& cx, const Task& t, int pt, int n, int q) {
  constexpr OpD d = kOps[I];
  if constexpr (d.gs == 1) {
    return static_cast<unsigned>(d.ys_off * 4 + ((((t.a + d.PG * t.b) * d.PT + pt) * d.NT + n) * 4 + q) * 1024) + gofs(cx, d.g0);
  } else {
    static_assert(d.CG == 1 && d.P % 32 == 0, "packed two-tap conv on 32x32 tiles: one channel group, whole tiles per stream");
    const int vt = t.a * d.PT + pt, gi = vt >> clog2(d.P / 32), tp = vt & (d.P / 32 - 1);
    return static_cast<unsigned>(d.ys_off * 4 + ((tp * d.NT + n) * 4 + q) * 1024) + gofs(cx, d.g0 + gi);
  }
}
// the same for the row-wise epilogue's items ([virtual position][packed channel] float4)
template <int I>
__device__ __forceinline__ unsigned x16_ys_off(const Ctx& cx, int item) {
  constexpr OpD d = kOps[I];
  constexpr int per = d.P * ntot(d) / 4;
  if constexpr (d.gs == 1) return static_cast<unsigned>(d.ys_off * 4 + item * 16) + gofs(cx, d.g0);
  else return static_cast<unsigned>(d.ys_off * 4 + (item & (per - 1)) * 16) + gofs(cx, d.g0 + (item >> clog2(per)));
}

// last frame's partial sums of op I ([pos][packed channel] fp32), in the layout its epilogue wants them
template <int I, int NY>
__device__ __forceinline__ void prefetch_y(const Ctx& cx, int tid, f32x4 (&yp)[NY]) {
  if constexpr (yp_regs(I) > 0) {
    constexpr OpD d = kOps[I];
    if constexpr (d.path == P_R32B) {
      const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
      const Task t = conv_task<I>(wave);
      sfor<yp_regs(I)>([&](auto ii) {
        constexpr int i = decltype(ii)::value, q = i & 3, n = (i >> 2) % d.NT, pt = (i >> 2) / d.NT;
        // (32x32 tiles: the sums live in HBM in accumulator order -- [wave task][pt][n][q][lane] float4 -- so that a wave's load /
        //  store is 1 KB contiguous; a [pos][channel] layout costs 32 partial lines per instruction)
        yp[i] = ld_once(cx.ysr, r32_ys_off<I>(cx, t, pt, n, q) + static_cast<unsigned>(lane * 16));
      });
    } else if constexpr (d.epl == 1) {
      // one output element per lane (x_epilogue1): element e = [virtual position][packed channel], the thread's own
      constexpr int total = d.gs * d.P * ntot(d);
      const int e = tid < total ? tid : total - 1;
      yp[0][0] = ld_once1(cx.ysr, x16_ys_off<I>(cx, e >> 2) + static_cast<unsigned>((e & 3) * 4));
    } else {
      constexpr int per = d.P * ntot(d) / 4, total = d.gs * per;
      sfor<yp_regs(I)>([&](auto ii) {
        constexpr int i = decltype(ii)::value;
        int item = tid + THREADS * i;
        if ((i + 1) * THREADS > total) item = item < total ? item : total - 1;
        yp[i] = ld_once(cx.ysr, x16_ys_off<I>(cx, item));
      });
    }
  }
}

template <int I, int NX>
__device__ __forceinline__ void ext_load(const Ctx& cx, int tid, f32x4 (&wx)[NX]) {
  if constexpr (ext_sf(I) > 0) {
    constexpr OpD d = kOps[I];
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const Task t = conv_task<I>(wave);
    const gcb_t wbase = cx.wb + static_cast<unsigned long long>(static_cast<unsigned>((d.w_off + t.wbase_f) * 4));
    sfor<ext_sf(I)>([&](auto ff) {
      constexpr int sf = carry_w(I) + decltype(ff)::value;
      wx[decltype(ff)::value] = ldb(wbase + static_cast<unsigned long long>(sf * 1024), static_cast<unsigned>(lane * 16));
    });
  }
}

// ---- completing the next image: staged parts + halos (between the two barriers of op I) --------------------------
template <int I, int N1, int N2>
__device__ __forceinline__ void build_next(int tid, const f32x4 (&p1)[N1], const f32x4 (&p2)[N2]) {
  constexpr int J = nxt_of(I), ST = stg_threads(I);
  if constexpr (ST == THREADS) {
    stage_store<J, 1, THREADS>(tid, p1);
  } else {
    if (__builtin_amdgcn_readfirstlane(tid >> 6) >= (THREADS - ST) / 64) stage_store<J, 1, ST>(tid - (THREADS - ST), p1);
  }
  stage_store<J, 2, THREADS>(tid, p2);
  zero_halos<J>(tid);
}
// the second-round part of an op's own two-round image (between the two barriers in the middle of the op)
template <int I, int N3, int N4>
__device__ __forceinline__ void store_round2(int tid, const f32x4 (&p3)[N3], const f32x4 (&p4)[N4]) {
  constexpr int ST = stg_threads(I);
  if constexpr (ST == THREADS) {
    stage_store<I, 3, THREADS>(tid, p3);
  } else {
    if (__builtin_amdgcn_readfirstlane(tid >> 6) >= (THREADS - ST) / 64) stage_store<I, 3, ST>(tid - (THREADS - ST), p3);
  }
  stage_store<I, 4, THREADS>(tid, p4);
}

// LDS address of (output row `row`, channel c) inside the forward target of op I (hi plane of a three-plane image)
template <int I>
__device__ __forceinline__ int fwd_addr(int row, int c) {
  constexpr Fwd f = kOps[I].fwd;
  return f.base_b + img_row_rt(f.pitch_b, f.pair, f.half_b, f.row0 + row) + c * esz_of(f.fmt);
}
template <int I>
__device__ __forceinline__ void fwd_st4(int row, int c, const f32x4& v) { img_st4<kOps[I].fwd.fmt>(fwd_addr<I>(row, c), kOps[I].fwd.plane_b, v); }
template <int I>
__device__ __forceinline__ f32x4 fwd_ld4(int row, int c) { return img_ld4<kOps[I].fwd.fmt>(fwd_addr<I>(row, c), kOps[I].fwd.plane_b); }
// packed plans: the same for the op's stream slot gi (its sub-image of the target is fwd.gstride_b further); fwd_has: do this slot's rows go to LDS at all?
template <int I>
__device__ __forceinline__ void fwd_st4g(int gi, int row, int c, const f32x4& v) {
  img_st4<kOps[I].fwd.fmt>(fwd_addr<I>(row, c) + (kOps[I].gs > 1 ? gi * kOps[I].fwd.gstride_b : 0), kOps[I].fwd.plane_b, v);
}
template <int I>
__device__ __forceinline__ f32x4 fwd_ld4g(int gi, int row, int c) {
  return img_ld4<kOps[I].fwd.fmt>(fwd_addr<I>(row, c) + (kOps[I].gs > 1 ? gi * kOps[I].fwd.gstride_b : 0), kOps[I].fwd.plane_b);
}
constexpr bool fwd_all(const OpD& d) { return d.fwd.on && d.fwd.mask == (1 << d.gs) - 1; }
template <int I>
__device__ __forceinline__ bool fwd_has(int gi) {
  constexpr OpD d = kOps[I];
  if constexpr (!d.fwd.on || d.fwd.mask == 0) return false;
  else if constexpr (fwd_all(d)) return true;
  else return ((d.fwd.mask >> gi) & 1) != 0;
}
// does op I feed an LSTM / dilated-dense op (which reads its rows as fp32 from XCOPY_B)?
constexpr bool feeds_x(int i) { return i + 1 < kNumOps && (kOps[i + 1].type == T_LSTM || kOps[i + 1].type == T_DDB); }

// ---- row-wise epilogue of the X16B path ---------------------------------------------------------------------------
// LPG lanes per output row (float4 each): K-slice sum + bias, LayerNorm over the row's channels, PReLU, stores.
template <int I, int NTHR = THREADS, int NY>
__device__ __forceinline__ void x_epilogue(const Ctx& cx, int tid, const f32x4 (&yp)[NY]) {
  constexpr OpD d = kOps[I];
  // (two-tap convs: slices [0, KS) of the exchange buffer hold the NEXT frame's partial sums, [KS, 2 KS) this frame's)
  constexpr int GC = d.gc, LPG = GC / 4, R = d.R, NTOT = ntot(d), KS = ex_group(d), K0 = d.ys ? KS : 0, OPB = (NTOT + 4) * 4;
  constexpr int VP = d.gs * d.P;      // (packed plans: the positions of the op's streams side by side)
  constexpr int total = VP * NTOT / 4, passes = (total + NTHR - 1) / NTHR;
  static_assert(NTHR == THREADS || passes == 1, "the carried sums are dealt to 512 threads: a pass of 256 sees the same items only if it is the only one");
  const int li = tid & (LPG - 1);
  const int u0 = tid >> clog2(LPG);
  const int r = u0 & (R - 1);
  const f32x4 bias = lds4(SCR_B + (r * GC + 4 * li) * 4);
  const f32x4 wsc = lds4(SCR_B + (NTOT + r * GC + 4 * li) * 4);
  f32x4 gm = bias, bt = bias;
  float alpha = 0.f;
  if constexpr (d.ln) {
    gm = lds4(SCR_B + (2 * NTOT + 4 * li) * 4);
    bt = lds4(SCR_B + (2 * NTOT + GC + 4 * li) * 4);
    alpha = lds1(SCR_B + (2 * NTOT + 2 * GC) * 4);
  }
  constexpr bool CSUM = feeds_ctfa_sums(I);
  f32x4 cs = {0.f, 0.f, 0.f, 0.f};      // (CSUM: this thread's rows of the output, added up -- every row of its passes has the thread's channel quad)
  sfor<passes>([&](auto pp) {
    constexpr int ps = decltype(pp)::value;
    const int u = u0 + ps * (NTHR / LPG);
    if ((ps + 1) * NTHR <= total || FZ_LIKELY(u * LPG < total)) {
      const int vpos = u >> clog2(R);
      const int gi = d.gs > 1 ? vpos >> clog2(d.P) : 0, pos = d.gs > 1 ? vpos & (d.P - 1) : vpos;      // stream slot, position inside the stream
      const int eb = d.ex_b + vpos * OPB + (r * GC + 4 * li) * 4;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      sfor<KS>([&](auto kk) { v += lds4(eb + (K0 + decltype(kk)::value) * (VP * OPB)); });
      if constexpr (d.ys != 0) v += yp[ps];      // W[tap 0] x_{t-1}, computed by this op one frame ago
      v = v * wsc + bias;
      if constexpr (d.ln) {
        const float mean = group_sum<LPG>(v[0] + v[1] + v[2] + v[3]) * (1.0f / GC);
        v -= mean;
        const float q = group_sum<LPG>(v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3]);
        const float rstd = __builtin_amdgcn_rsqf(q * (1.0f / GC) + FZ_LN_EPS);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float y = v[i] * rstd * gm[i] + bt[i];
          v[i] = y >= 0.f ? y : alpha * y;
        }
      }
      const int row = pos * d.row_mul + d.row_add + r;
      const unsigned go = gofs(cx, d.g0 + gi);
      if constexpr (is_up(d)) FZ_TRACE4(cx, d.g0 + gi, 13 + d.bidx, row, 128, 4 * li, v);
      if constexpr (CSUM) cs += v;
      // (the rows of the next image first: the next op's MFMAs wait for them behind the barrier; the HBM copies are nobody's critical path,
      //  and a store that has to queue behind the op's prefetches at the memory pipe would hold the LDS writes back with it)
      if constexpr (d.fwd.on) { if (fwd_has<I>(gi)) fwd_st4g<I>(gi, row, 4 * li, v); }
      if constexpr (feeds_x(I)) lds4(d.xcopy_b + gi * 1024 + (row * GC + 4 * li) * 4) = v;
      sched_pin();
      if constexpr (d.d0_on) { if (FZ_D0(d, cx)) st_state<d.nt0 != 0>(d.d0_src == S_CUR ? cx.sbc : cx.sbs, static_cast<unsigned>((d.d0_off + row * d.d0_ld + 4 * li) * 4) + go, v); }
      if constexpr (d.d1_on) { if (FZ_D1(d, cx)) st_state<d.nt1 != 0>(d.d1_src == S_CUR ? cx.sbc : cx.sbs, static_cast<unsigned>((d.d1_off + row * d.d1_ld + 4 * li) * 4) + go, v); }
    }
  });
  if constexpr (CSUM) {
    static_assert(LPG == 16 && NTHR == THREADS, "column sums: 16 lanes per row of 64 channels, all eight waves leave a partial row");
#pragma unroll
    for (int e = 0; e < 4; ++e) cs[e] = xor32_sum(xor16_sum(cs[e]));
    if ((tid & 63) < 16) lds4(kOps[I + 1].scr_b + CSUM_OFF_B + ((tid >> 6) * 64 + 4 * li) * 4) = cs;
  }
  if constexpr (d.ys != 0) {
    // next frame's partial sums W[tap 0] x_t: K slices summed, stored raw ([pos][packed channel] = item order); the items are dealt
    // from the top of the workgroup, beside the epilogue rows of the low waves
    sfor<passes>([&](auto pp) {
      constexpr int ps = decltype(pp)::value;
      const int item = (NTHR - 1 - tid) + ps * NTHR;
      if ((ps + 1) * NTHR <= total || FZ_LIKELY(item < total)) {
        const int u = item >> clog2(LPG), l2 = item & (LPG - 1);
        const int eb = d.ex_b + (u >> clog2(R)) * OPB + ((u & (R - 1)) * GC + 4 * l2) * 4;
        f32x4 y = {0.f, 0.f, 0.f, 0.f};
        sfor<KS>([&](auto kk) { y += lds4(eb + decltype(kk)::value * (VP * OPB)); });
        st_next(cx.ysw, x16_ys_off<I>(cx, item), y);
      }
    });
  }
}

// The same with ONE output element per lane (OpD::epl 1; layers with at most 512 outputs): an output row of GC channels is GC consecutive
// lanes, so the K-slice sums, scale + bias, the two LayerNorm sums (DPP inside a row of 16 lanes, lane swaps across rows), PReLU and the
// three-plane split are ~50 dependent VALU instructions on EVERY wave that holds a row instead of ~110 on the one or two waves that hold
// all rows as float4 -- this chain is the critical path of a small op (profiles/r05_v0_wave_trace.txt: 1 000 of its 2 800 cycles).
template <int I, int NY>
__device__ __forceinline__ void x_epilogue1(const Ctx& cx, int tid, const f32x4 (&yp)[NY]) {
  constexpr OpD d = kOps[I];
  constexpr int GC = d.gc, R = d.R, NTOT = ntot(d), KS = ex_group(d), K0 = d.ys ? KS : 0, OPB = (NTOT + 4) * 4;
  constexpr int VP = d.gs * d.P, total = VP * NTOT;
  static_assert(total <= THREADS && (GC == 32 || GC == 64) && !is_up(d), "epl 1: at most one output element per thread, rows of 32 / 64 lanes");
  static_assert(total % GC == 0 && (total % 64 == 0 || total == 32), "epl 1: whole rows per wave (the LayerNorm sums are lane exchanges)");
  const int cch = tid & (GC - 1);                 // channel inside the output row
  const int u = tid >> clog2(GC);                 // output row: virtual position x sub-row
  const int r = u & (R - 1);
  constexpr bool CSUM = feeds_ctfa_sums(I);
  if (total == THREADS || FZ_LIKELY(tid < total)) {
    const float bias = lds1(SCR_B + (r * GC + cch) * 4), wsc = lds1(SCR_B + (NTOT + r * GC + cch) * 4);
    float gm = 0.f, bt = 0.f, alpha = 0.f;
    if constexpr (d.ln) {
      gm = lds1(SCR_B + (2 * NTOT + cch) * 4);
      bt = lds1(SCR_B + (2 * NTOT + GC + cch) * 4);
      alpha = lds1(SCR_B + (2 * NTOT + 2 * GC) * 4);
    }
    const int vpos = u >> clog2(R);
    const int gi = d.gs > 1 ? vpos >> clog2(d.P) : 0, pos = d.gs > 1 ? vpos & (d.P - 1) : vpos;
    const int eb = d.ex_b + vpos * OPB + (r * GC + cch) * 4;
    float v = 0.f;
    sfor<KS>([&](auto kk) { v += lds1(eb + (K0 + decltype(kk)::value) * (VP * OPB)); });
    if constexpr (d.ys != 0) v += yp[0][0];      // W[tap 0] x_{t-1}, computed by this op one frame ago
    v = v * wsc + bias;
    if constexpr (d.ln) {
      const float mean = group_sum<GC>(v) * (1.0f / GC);
      v -= mean;
      const float rstd = __builtin_amdgcn_rsqf(group_sum<GC>(v * v) * (1.0f / GC) + FZ_LN_EPS);
      const float y = v * rstd * gm + bt;
      v = y >= 0.f ? y : alpha * y;
    }
    const int row = pos * d.row_mul + d.row_add + r;
    const unsigned go = gofs(cx, d.g0 + gi);
    if constexpr (d.fwd.on) {
      if (fwd_has<I>(gi)) img_st1<d.fwd.fmt>(fwd_addr<I>(row, cch) + (d.gs > 1 ? gi * d.fwd.gstride_b : 0), d.fwd.plane_b, v);
    }
    if constexpr (feeds_x(I)) lds1(d.xcopy_b + gi * 1024 + (row * GC + cch) * 4) = v;
    if constexpr (CSUM) lds1(kOps[I + 1].scr_b + CSUM_OFF_B + tid * 4) = v;      // (every wave holds one output row of 64 channels: the CTFA adds the eight up)
    sched_pin();
    if constexpr (d.d0_on) { if (FZ_D0(d, cx)) st_state1<d.nt0 != 0>(d.d0_src == S_CUR ? cx.sbc : cx.sbs, static_cast<unsigned>((d.d0_off + row * d.d0_ld + cch) * 4) + go, v); }
    if constexpr (d.d1_on) { if (FZ_D1(d, cx)) st_state1<d.nt1 != 0>(d.d1_src == S_CUR ? cx.sbc : cx.sbs, static_cast<unsigned>((d.d1_off + row * d.d1_ld + cch) * 4) + go, v); }
  }
  if constexpr (d.ys != 0) {
    // next frame's partial sums W[tap 0] x_t: K slices summed, stored raw ([pos][packed channel]) as float4 items dealt from the top of the workgroup
    constexpr int items = total / 4, LPG4 = GC / 4;
    const int item = THREADS - 1 - tid;
    if (FZ_LIKELY(item < items)) {
      const int uu = item >> clog2(LPG4), l2 = item & (LPG4 - 1);
      const int eb = d.ex_b + (uu >> clog2(R)) * OPB + ((uu & (R - 1)) * GC + 4 * l2) * 4;
      f32x4 y = {0.f, 0.f, 0.f, 0.f};
      sfor<KS>([&](auto kk) { y += lds4(eb + decltype(kk)::value * (VP * OPB)); });
      st_next(cx.ysw, x16_ys_off<I>(cx, item), y);
    }
  }
}

// ---- conv op, small layers: 16x16x32 bf16 tiles, K split over waves, LDS exchange -------------------------------------
// Wave task = (position group pg, 16-channel tile ct, K slice (time tap ks_t, channel-group range ks_g)); per K step (32
// channels of one (time tap, frequency tap) segment) and position tile: three MFMAs, one per plane of the image, each into
// its own accumulator (summed hi + (mid + lo) at the end).  Lane (j = lane & 15, h = lane >> 4): A = weights of channel
// 16 ct + j, B = position j, both for the 8 channels 8 h .. 8 h + 7 of the step; D = channels 4 h .. 4 h + 3 of position j.
template <int I, int N1, int N3>
__device__ __forceinline__ void conv_x16b(const Ctx& cx, int tid, Carry<I>& c, const f32x4 (&p1)[N1], const f32x4 (&p3)[N3]) {
  constexpr OpD d = kOps[I];
  constexpr bool UP = is_up(d);
  constexpr int PT = d.PT, GW = gw(d), NF = conv_nf(d), NSF = conv_nsf(d), CW = carry_w(I), NTOT = ntot(d), OPB = (NTOT + 4) * 4;
  static_assert(d.img.fmt == 1, "X16B: three-plane image");
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const Task t = conv_task<I>(wave);
  const int ks_t = t.ks >> clog2(d.KSg), ks_g = t.ks & (d.KSg - 1);
  const int j = lane & 15, h = lane >> 4;
  int lane_b[PT];
  constexpr int VP = d.gs * d.P;      // packed plans: virtual position = stream slot * P + position
#pragma unroll
  for (int pt = 0; pt < PT; ++pt) {
    int pos = 16 * (t.a * PT + pt) + j;
    if (pos > VP - 1) pos = VP - 1;
    const int rowb = d.gs > 1 ? (pos >> clog2(d.P)) * d.img.gstride_b + (pos & (d.P - 1)) * d.img.pitch_b : pos * d.img.pitch_b;
    lane_b[pt] = rowb + 16 * h + ks_t * d.img.tap_b + ks_g * (GW * 64);
  }
  constexpr bool BOTH = x16_both(d);            // two-tap conv, the wave owns both taps: segments 0..2 (tap 0) go to the second set
  constexpr bool TWO = UP || BOTH;
  // NT channel tiles per wave task (the planner splits K before it splits the channels: the waves that would differ only in their
  // channel tile read the same B fragments).  Fragment f = (K step f / NT, tile f % NT): a B fragment is read once per K step.
  constexpr int NT = d.NT;
  static_assert(d.CG * NT * 16 == d.N && NF % NT == 0, "channel groups x tiles per task");
  f32x4 acc[PT][NT][3], acco[TWO ? PT : 1][TWO ? NT : 1][3];
#pragma unroll
  for (int pt = 0; pt < PT; ++pt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) { acc[pt][nt][pl] = f32x4{0.f, 0.f, 0.f, 0.f}; if constexpr (TWO) acco[pt][nt][pl] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  const gcb_t wbase = cx.wb + static_cast<unsigned long long>(static_cast<unsigned>((d.w_off + t.wbase_f) * 4));
  const unsigned lane16 = static_cast<unsigned>(lane * 16);
  if (t.active) pin_regs(c.w);
  FZ_STAMP(I, 5);
  bf16x8 bfr[PT][3];
  auto half = [&](auto lo_, auto hi_) {
    constexpr int lo = decltype(lo_)::value, hi = decltype(hi_)::value;
    if (FZ_LIKELY(t.active)) {
      sfor<hi - lo>([&](auto ff) {
        constexpr int f = lo + decltype(ff)::value;
        constexpr int nt = f % NT, r = f / NT, s = r / GW, g = r % GW;
        constexpr int sf = f / 2;
        const bf16x8 a = wfrag(c.w[sf % CW], f % 2);
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) {
#pragma unroll
          for (int pl = 0; pl < 3; ++pl) {
            if constexpr (nt == 0) bfr[pt][pl] = as_bf(lds4(lane_b[pt] + d.seg_b[s] + g * 64 + pl * d.img.plane_b));
            if constexpr ((UP && s == 2) || (BOTH && s < 3)) acco[pt][nt][pl] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bfr[pt][pl], acco[pt][nt][pl], 0, 0, 0);
            else acc[pt][nt][pl] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bfr[pt][pl], acc[pt][nt][pl], 0, 0, 0);
          }
        }
        if constexpr ((f % 2 == 1 || f + 1 == NF) && sf + CW < NSF) {
          c.w[sf % CW] = ldb(wbase + static_cast<unsigned long long>((sf + CW) * 1024), lane16);
          sched_pin();
        }
      });
    }
  };
  if constexpr (d.rounds == 2) {
    static_assert(NF % 2 == 0, "two rounds: the same number of fragments per round");
    half(std::integral_constant<int, 0>{}, std::integral_constant<int, NF / 2>{});
    lds_barrier();
    store_round2<I>(tid, p3, c.p4);
    lds_barrier();
    half(std::integral_constant<int, NF / 2>{}, std::integral_constant<int, NF>{});
  } else {
    half(std::integral_constant<int, 0>{}, std::integral_constant<int, NF>{});
  }
  if (FZ_LIKELY(t.active)) {
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
      const int pos = 16 * (t.a * PT + pt) + j;
      if (pos < VP) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const int eb = d.ex_b + ((BOTH ? 1 : t.ks) * VP + pos) * OPB + (16 * (t.b * NT + nt) + 4 * h) * 4;
          lds4(eb) = acc[pt][nt][0] + (acc[pt][nt][1] + acc[pt][nt][2]);
          if constexpr (UP) lds4(eb + d.N * 4) = acco[pt][nt][0] + (acco[pt][nt][1] + acco[pt][nt][2]);
          if constexpr (BOTH) lds4(eb - VP * OPB) = acco[pt][nt][0] + (acco[pt][nt][1] + acco[pt][nt][2]);      // slice 0: the next frame's sums
        }
      }
    }
  }
  if constexpr (prm_dword(d)) { if (FZ_LIKELY(tid < nparams(d))) lds1(SCR_B + tid * 4) = c.prm[0]; }
  else { if (FZ_LIKELY(tid < (nparams(d) + 3) / 4)) lds4(SCR_B + tid * 16) = c.prm; }
  FZ_STAMP(I, 1);
  lds_barrier();
  FZ_STAMP(I, 2);
  if constexpr (d.epl == 1) x_epilogue1<I>(cx, tid, c.yp);
  else x_epilogue<I>(cx, tid, c.yp);
  FZ_STAMP(I, 3);
  build_next<I>(tid, p1, c.p);
  FZ_STAMP(I, 4);
}

// ---- conv op, large layers: 32x32x16 bf16 tiles, whole LayerNorm groups per wave, epilogue in registers -----------------
// Lane (j = lane & 31, h = lane >> 5): A = weights of channel 32 T + j, B = position j, both for the 8 channels 8 h .. 8 h + 7
// of the K step (16 channels of one segment); three MFMAs per tile and step (hi, mid, lo plane) into the tile's accumulator.
// The two images that do not fit LDS as three planes are fp32 (img.fmt 0): their B fragments are split when they are read.

template <int I, int N1, int N3, int NX>
__device__ __forceinline__ void conv_r32b(const Ctx& cx, int tid, Carry<I>& c, const f32x4 (&p1)[N1], const f32x4 (&p3)[N3], f32x4 (&wx)[NX]) {
  constexpr OpD d = kOps[I];
  constexpr bool UP = is_up(d);
  constexpr int PT = d.PT, NT = d.NT, G = kgroups(d), NF = conv_nf(d), NSF = conv_nsf(d), CW = carry_w(I), NTOT = ntot(d), EXT = ext_sf(I);
  constexpr int NA = UP ? 2 * NT : NT;               // accumulator tiles per position tile (UP: NT for the even output row, NT for the odd one)
  constexpr int FMT = d.img.fmt;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const Task t = conv_task<I>(wave);
  const int j = lane & 31, h = lane >> 5;
  int lane_b[PT];
  // packed plans: virtual position 32 vt + j of tile vt = stream slot * P + position (a tile may hold several streams' positions when
  // P < 32; the two-tap convs, whose carried sums are laid out by whole tiles, always have P >= 32)
  static_assert(d.gs == 1 || !r32_two(d) || d.P % 32 == 0, "packed plan: two-tap conv on 32x32 tiles needs whole tiles per stream");
  int lane_g[PT], lane_p[PT];                       // per position tile: this lane's stream slot, its position inside the stream
#pragma unroll
  for (int pt = 0; pt < PT; ++pt) {
    const int vpos = 32 * (t.a * PT + pt) + j;
    lane_g[pt] = d.gs > 1 ? vpos >> clog2(d.P) : 0;
    lane_p[pt] = d.gs > 1 ? vpos & (d.P - 1) : vpos;
    lane_b[pt] = (d.gs > 1 ? lane_g[pt] * d.img.gstride_b : 0) + lane_p[pt] * d.img.pitch_b + (8 * esz_of(FMT)) * h;
  }
  f32x16 acc[PT][NA];
#pragma unroll
  for (int pt = 0; pt < PT; ++pt)
#pragma unroll
    for (int n = 0; n < NA; ++n)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[pt][n][e] = 0.f;
  const gcb_t wbase = cx.wb + static_cast<unsigned long long>(static_cast<unsigned>((d.w_off + t.wbase_f) * 4));
  const unsigned lane16 = static_cast<unsigned>(lane * 16);
  if (t.active) { pin_regs(c.w); if constexpr (EXT > 0) pin_regs(wx); }
  FZ_STAMP(I, 5);
  bf16x8 b[PT][3];
  auto half = [&](auto lo_, auto hi_) {
    constexpr int lo = decltype(lo_)::value, hi = decltype(hi_)::value;
    if (FZ_LIKELY(t.active)) {
      sfor<hi - lo>([&](auto ff) {
        constexpr int f = lo + decltype(ff)::value;
        constexpr int nt = f % NT, sg = f / NT, s = sg / G, g = sg % G;
        constexpr int na = UP ? (s == 2 ? NT : 0) + nt : nt;
        if constexpr (nt == 0) {
#pragma unroll
          for (int pt = 0; pt < PT; ++pt) {
            if constexpr (FMT == 1) {
#pragma unroll
              for (int pl = 0; pl < 3; ++pl) b[pt][pl] = as_bf(lds4(lane_b[pt] + d.seg_b[s] + g * 32 + pl * d.img.plane_b));
            } else {
              split8(lds4(lane_b[pt] + d.seg_b[s] + g * 64), lds4(lane_b[pt] + d.seg_b[s] + g * 64 + 16), b[pt]);
            }
          }
        }
        constexpr int sf = f / 2;
        const f32x4& wsf = (EXT > 0 && sf >= CW) ? wx[(EXT > 0 && sf >= CW) ? sf - CW : 0] : c.w[sf % CW];
        const bf16x8 a = wfrag(wsf, f % 2);
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
#pragma unroll
          for (int pt = 0; pt < PT; ++pt) acc[pt][na] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b[pt][pl], acc[pt][na], 0, 0, 0);
        if constexpr (EXT == 0 && (f % 2 == 1 || f + 1 == NF) && sf + CW < NSF) {
          c.w[sf % CW] = ldb(wbase + static_cast<unsigned long long>((sf + CW) * 1024), lane16);
          sched_pin();
        }
      });
    }
  };
  if constexpr (d.rounds == 2) {
    static_assert(NF % 2 == 0, "two rounds: the same number of fragments per round");
    half(std::integral_constant<int, 0>{}, std::integral_constant<int, NF / 2>{});
    lds_barrier();
    store_round2<I>(tid, p3, c.p4);
    lds_barrier();
    half(std::integral_constant<int, NF / 2>{}, std::integral_constant<int, NF>{});
  } else {
    half(std::integral_constant<int, 0>{}, std::integral_constant<int, NF>{});
  }
  if constexpr (r32_two(d)) {
    // waves 0..3 of a two-tap conv: W[tap 0] x_t, the next frame's partial sums, stored raw in accumulator order (prefetch_y)
    if (t.ks == 0) {
#pragma unroll
      for (int pt = 0; pt < PT; ++pt)
#pragma unroll
        for (int n = 0; n < NA; ++n)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const f32x4 v = {acc[pt][n][4 * q], acc[pt][n][4 * q + 1], acc[pt][n][4 * q + 2], acc[pt][n][4 * q + 3]};
            st_next(cx.ysw, r32_ys_off<I>(cx, t, pt, n, q) + static_cast<unsigned>(lane * 16), v);
          }
    } else {
      // waves 4..7: this frame's sums start from what the op handed over one frame ago
#pragma unroll
      for (int pt = 0; pt < PT; ++pt)
#pragma unroll
        for (int n = 0; n < NA; ++n)
#pragma unroll
          for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[pt][n][4 * q + e] += c.yp[(pt * NT + n) * 4 + q][e];
    }
  }
  if constexpr (prm_dword(d)) { if (FZ_LIKELY(tid < nparams(d))) lds1(SCR_B + tid * 4) = c.prm[0]; }
  else { if (FZ_LIKELY(tid < (nparams(d) + 3) / 4)) lds4(SCR_B + tid * 16) = c.prm; }
  FZ_STAMP(I, 1);
  lds_barrier();                      // every wave is done with this op's image; parameters are in LDS
  FZ_STAMP(I, 2);
  if (t.active && (!r32_two(d) || t.ks != 0)) {
    float alpha = 0.f;
    if constexpr (d.ln) alpha = lds1(SCR_B + (2 * NTOT + 2 * d.gc) * 4);
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
      const int pos = lane_p[pt], gi = lane_g[pt];      // position inside its stream, stream slot
      const unsigned go = gofs(cx, d.g0 + gi);
      // packed channel of accumulator element e of tile n: nb(n) + 8 (e >> 2) + 4 h + (e & 3)
      if constexpr (d.ln) {
        float s = 0.f;
#pragma unroll
        for (int n = 0; n < NA; ++n) {
          const int nb = 32 * (t.b * NT + n);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const f32x4 bi = lds4(SCR_B + (nb + 8 * q + 4 * h) * 4), sc = lds4(SCR_B + (NTOT + nb + 8 * q + 4 * h) * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { acc[pt][n][4 * q + e] = acc[pt][n][4 * q + e] * sc[e] + bi[e]; s += acc[pt][n][4 * q + e]; }
          }
        }
        const float mean = xor32_sum(s) * (1.0f / d.gc);
        float qq = 0.f;
#pragma unroll
        for (int n = 0; n < NA; ++n)
#pragma unroll
          for (int e = 0; e < 16; ++e) { acc[pt][n][e] -= mean; qq += acc[pt][n][e] * acc[pt][n][e]; }
        const float rstd = __builtin_amdgcn_rsqf(xor32_sum(qq) * (1.0f / d.gc) + FZ_LN_EPS);
#pragma unroll
        for (int n = 0; n < NA; ++n) {
          const int c0 = (32 * (t.b * NT + n)) & (d.gc - 1);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const f32x4 gm = lds4(SCR_B + (2 * NTOT + c0 + 8 * q + 4 * h) * 4);
            const f32x4 bt = lds4(SCR_B + (2 * NTOT + d.gc + c0 + 8 * q + 4 * h) * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float y = acc[pt][n][4 * q + e] * rstd * gm[e] + bt[e];
              acc[pt][n][4 * q + e] = y >= 0.f ? y : alpha * y;
            }
          }
        }
      } else {
#pragma unroll
        for (int n = 0; n < NA; ++n) {
          const int nb = UP ? ((n / NT) * d.N + 32 * (t.b * NT + n % NT)) : 32 * (t.b * NT + n);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const f32x4 bi = lds4(SCR_B + (nb + 8 * q + 4 * h) * 4), sc = lds4(SCR_B + (NTOT + nb + 8 * q + 4 * h) * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[pt][n][4 * q + e] = acc[pt][n][4 * q + e] * sc[e] + bi[e];
          }
        }
      }
      // stores: tile n covers channels c0 .. c0+31 of output row  pos * row_mul + row_add + r
#pragma unroll
      for (int n = 0; n < NA; ++n) {
        const int nb = UP ? ((n / NT) * d.N + 32 * (t.b * NT + n % NT)) : 32 * (t.b * NT + n);
        const int r = nb / d.gc, c0 = nb & (d.gc - 1);
        const int row = pos * d.row_mul + d.row_add + r;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 v = {acc[pt][n][4 * q], acc[pt][n][4 * q + 1], acc[pt][n][4 * q + 2], acc[pt][n][4 * q + 3]};
          const int cc = c0 + 8 * q + 4 * h;
          if constexpr (UP) FZ_TRACE4(cx, d.g0 + gi, 13 + d.bidx, row, 128, cc, v);
          if constexpr (d.fwd.on) { if (fwd_has<I>(gi)) fwd_st4g<I>(gi, row, cc, v); }
          if constexpr (d.d0_on) { if (FZ_D0(d, cx)) st_state<d.nt0 != 0>(d.d0_src == S_CUR ? cx.sbc : cx.sbs, static_cast<unsigned>((d.d0_off + row * d.d0_ld + cc) * 4) + go, v); }
          if constexpr (d.d1_on) { if (FZ_D1(d, cx)) st_state<d.nt1 != 0>(d.d1_src == S_CUR ? cx.sbc : cx.sbs, static_cast<unsigned>((d.d1_off + row * d.d1_ld + cc) * 4) + go, v); }
        }
      }
    }
  }
  FZ_STAMP(I, 3);
  build_next<I>(tid, p1, c.p);
  FZ_STAMP(I, 4);
}

// ---- input layer: 1 -> 64 conv + LN + PReLU (proposed.py:218-225), straight into the image of msfe6_en_in ---------------
template <int I>
__device__ __forceinline__ void input_op(const Ctx& cx, int tid) {
  constexpr OpD d = kOps[I];
  static_assert(d.gs == 1, "the input layer runs one stream at a time");
  const int c4 = tid & 15;
  const unsigned pb = static_cast<unsigned>(d.p_off * 4 + c4 * 16);
  const f32x4 w = ldb(cx.wb, pb), bb = ldb(cx.wb, pb + 256), gm = ldb(cx.wb, pb + 512), bt = ldb(cx.wb, pb + 768);
  const float alpha = ldb1(cx.wb, static_cast<unsigned>(d.p_off * 4 + 1024));
  float x[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) x[i] = cx.io_in[d.g0 * 256 + (tid >> 4) + 32 * i];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int pos = (tid >> 4) + 32 * i;
    f32x4 y = w * x[i] + bb;
    const float s = group_sum<16>(y[0] + y[1] + y[2] + y[3]);
    y -= s * (1.0f / 64.0f);
    const float q = group_sum<16>(y[0] * y[0] + y[1] * y[1] + y[2] * y[2] + y[3] * y[3]);
    const float rstd = __builtin_amdgcn_rsqf(q * (1.0f / 64.0f) + FZ_LN_EPS);
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float v = y[e] * rstd * gm[e] + bt[e];
      o[e] = v >= 0.f ? v : alpha * v;
    }
    fwd_st4<I>(pos, 4 * c4, o);
    FZ_TRACE4(cx, d.g0, 0, pos, 64, 4 * c4, o);
  }
}

// what op I requests for the ops after it: the far-ahead staged parts of the next image but one, the first weights / parameters / carried
// sums of op I + 2
template <int I>
__device__ __forceinline__ void late_loads(const Ctx& cx, int tid, Carry<I + 1>& n) {
  stage_load<nxt_of(I + 1), 2, THREADS>(cx, tid, n.p);
  stage_load<I + 1, 4, THREADS>(cx, tid, n.p4);
  prefetch_w<I + 2>(cx, tid, n.w2, n.prm2);
  prefetch_y<I + 2>(cx, tid, n.yp2);
}

// ---- LSTM cell + Dense (proposed.py:70-119; converter_proposed.py:234-237), in place on the next conv's image ------------
// the four int8 weights of a dword (gates i, f, g, o of one K row of a unit) as floats
__device__ __forceinline__ f32x4 sbytes4(float packed) {
  const int v = __builtin_bit_cast(int, packed);
  return f32x4{static_cast<float>(static_cast<signed char>(v)), static_cast<float>(static_cast<signed char>(v >> 8)),
               static_cast<float>(static_cast<signed char>(v >> 16)), static_cast<float>(static_cast<signed char>(v >> 24))};
}
template <int I>
__device__ __forceinline__ void lstm_op(const Ctx& cx, int tid, Carry<I>& c, Carry<I + 1>& n) {
  // z = b + s_x (Qx x) + s_h (Qh h) with the gate columns interleaved (one dword = the int8 (i, f, g, o) weights of a unit for one K row:
  // fused_plan.hpp, blob layout of an LSTM op).  K is cut into 16 slices of x (threads (u, slice), tid < 336) and 4 slices of h
  // (tid 336..419); every operand arrived in the carry (slot map: [0, S0) the int8x4 rows of the thread's slice,
  // [S0, S0+2) its h values, [S0+2, S0+8) the Dense row of its output, S0+8 the unit's bias, S0+9 (cell state it updates, s_x, s_h); packed
  // plans: the h values of stream slot gi >= 1 at S0 + 10 + 2 (gi - 1) + {0, 1} -- the weights serve every stream of the op; gates and
  // Dense outputs are dealt to the threads as (stream, unit) / (stream, output)).
  constexpr OpD d = kOps[I];
  constexpr int KN = lstm_kn(d.din), XS = clog2(d.x_cols), S0 = lstm_s0(d), GS = d.gs;
  constexpr int PART = d.scr_b, HN = d.scr_b + 20 * 21 * 16, SGB = d.scr_gstride_b;      // stream slot gi: + gi * SGB
  constexpr bool DI8 = lstm_dense_i8(d.dout);
  const int u = tid % 21, sl = tid / 21;
  pin_regs(c.w);
  if (tid < 336) {
    f32x4 a[GS];
#pragma unroll
    for (int gi = 0; gi < GS; ++gi) a[gi] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < KN; ++j) {
      const int k = sl * KN + j;
      const f32x4 wq = sbytes4(c.w[j / 4][j % 4]);
#pragma unroll
      for (int gi = 0; gi < GS; ++gi) a[gi] += wq * lds1(d.xcopy_b + gi * 1024 + k * 4);        // (the conv op before left its rows here as fp32, [row][x_cols] = element k)
    }
#pragma unroll
    for (int gi = 0; gi < GS; ++gi) lds4(PART + gi * SGB + (sl * 21 + u) * 16) = a[gi];
  } else if (tid < 420) {
    f32x4 wq[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) wq[j] = sbytes4(c.w[j / 4][j % 4]);
#pragma unroll
    for (int gi = 0; gi < GS; ++gi) {
      const int SH = gi == 0 ? S0 : S0 + 10 + 2 * (gi - 1);
      f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < 6; ++j) a += wq[j] * c.w[SH + j / 4][j % 4];
      lds4(PART + gi * SGB + (sl * 21 + u) * 16) = a;
    }
  }
  FZ_WSTAMP(I, 3);
  lds_barrier();
  FZ_WSTAMP(I, 4);
  // the Dense row of this thread's output as floats -- widened here, beside the 21 gate threads, not behind the barrier they release
  float wd[21], bd, sd = 1.0f;
  if constexpr (DI8) {
#pragma unroll
    for (int q = 0; q < 6; ++q) {
      const f32x4 w4 = sbytes4(c.w[S0 + 2 + q / 4][q % 4]);
#pragma unroll
      for (int e = 0; e < 4; ++e) if (4 * q + e < 21) wd[4 * q + e] = w4[e];
    }
    bd = c.w[S0 + 3][2];
    sd = c.w[S0 + 3][3];
  } else {
#pragma unroll
    for (int k = 0; k < 21; ++k) wd[k] = c.w[S0 + 2 + k / 4][k % 4];
    bd = c.w[S0 + 2 + 5][1];
  }
#pragma unroll
  for (int k = 0; k < 21; ++k) asm volatile("" : "+v"(wd[k]));
  if (tid < 21 * GS) {          // thread (stream slot gi, unit uu): the unit's four gates
    const int gi = GS > 1 ? tid / 21 : 0, uu = GS > 1 ? tid % 21 : tid;
    f32x4 zz[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) zz[q] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s2 = 0; s2 < 16; ++s2) zz[s2 & 3] += lds4(PART + gi * SGB + (s2 * 21 + uu) * 16);      // (four chains: the partial rows are not one dependent sum)
    f32x4 zh = lds4(PART + gi * SGB + (16 * 21 + uu) * 16) + lds4(PART + gi * SGB + (17 * 21 + uu) * 16);
    zh += lds4(PART + gi * SGB + (18 * 21 + uu) * 16) + lds4(PART + gi * SGB + (19 * 21 + uu) * 16);
    const f32x4 z = c.w[S0 + 8] + ((zz[0] + zz[1]) + (zz[2] + zz[3])) * c.w[S0 + 9][1] + zh * c.w[S0 + 9][2];
    const float c_old = c.w[S0 + 9][0];
    const float gi_ = fast_sigmoid(z[0]), gf = fast_sigmoid(z[1]), gg = fast_tanh(z[2]), go = fast_sigmoid(z[3]);
    const float c_new = gf * c_old + gi_ * gg;
    const float h_new = go * fast_tanh(c_new);
    const unsigned so = gofs(cx, d.g0 + gi);
    st_next1(cx.sbc, static_cast<unsigned>((d.c_off + uu) * 4) + so, c_new);      // (h, c: the next frame's)
    st_next1(cx.sbc, static_cast<unsigned>((d.h_off + uu) * 4) + so, h_new);
    lds1(HN + gi * SGB + uu * 4) = h_new;
  }
  FZ_WSTAMP(I, 5);
  lds_barrier();
  FZ_WSTAMP(I, 6);
  // Dense: output n of stream gi by thread (gi * dout + n) mod 512 (its row of the Dense kernel arrived in the carry: 512 is a multiple
  // of dout, so a thread's row is the same in every pass)
  constexpr int DPASS = (d.dout * GS + THREADS - 1) / THREADS;
#pragma unroll
  for (int ps = 0; ps < DPASS; ++ps) {
    const int idx = tid + THREADS * ps;
    if (idx < d.dout * GS) {
      const int gi = GS > 1 ? idx >> clog2(d.dout) : 0, n = GS > 1 ? idx & (d.dout - 1) : tid;
      float a0 = 0.f, a1 = 0.f;
#pragma unroll
      for (int q = 0; q < 5; ++q) {
        const f32x4 h4 = lds4(HN + gi * SGB + 16 * q);
        a0 += wd[4 * q] * h4[0] + wd[4 * q + 1] * h4[1];
        a1 += wd[4 * q + 2] * h4[2] + wd[4 * q + 3] * h4[3];
      }
      a0 = fmaf(wd[20], lds1(HN + gi * SGB + 80), a0);
      const float a = DI8 ? fmaf(a0 + a1, sd, bd) : (a0 + a1) + bd;
      const int f = n >> XS, cc = n & (d.x_cols - 1);
      img_st1<d.x_fmt>(d.y_b + gi * d.x_gstride_b + f * d.x_pitch_b + cc * esz_of(d.x_fmt), d.x_plane_b, a);
      if constexpr (d.ldst_on) st_state1<d.nt0 != 0>(cx.sbc, static_cast<unsigned>((d.ldst_off + f * d.ldst_ld + cc) * 4) + gofs(cx, d.g0 + gi), a);
    }
  }
}

#if FZ_BASE
// ---- dilated-dense bottleneck of the baseline variant (models/nunet_tls.py:277-359, streaming converter_nunet_tls.py:374-411):
//      ddb_device.hpp's workgroup form -- input rows from the state tensor in HBM (the op before drained its stores), history
//      rings in HBM, output to HBM and, here, straight into the next conv's image.
template <int I>
__device__ __forceinline__ void ddb_op(const Ctx& cx, int tid, Carry<I>& c) {
  constexpr OpD d = kOps[I];
  // (profiling build: the block's own stamps, wave 0, go to the op's phase slots -- loads issued, loads landed, o_0, chain, ring stores, out)
  unsigned long long* dbg = (FZ_PROF && cx.prof) ? reinterpret_cast<unsigned long long*>(lds + (DDB_LDS_B + 72 * 1024) / 4) : nullptr;
  constexpr int G = d.x_cols / 2, F = d.din / d.x_cols;
  static_assert(ddbz_lds_floats<G>(F) * 4 <= 70 * 1024, "dilated-dense scratch over its LDS region");
  static_assert(carry_w(I) == DdbzCarry<THREADS, G, F>::N, "carry slots of the dilated-dense op");
  // input rows: the fp32 copy the strided conv before left at XCOPY_B ([F][2G]); output rows: split into the next image
  auto put_y = [&](int fo, int cq, const f32x4& r) { img_st4<d.x_fmt>(d.y_b + fo * d.x_pitch_b + 4 * cq * esz_of(d.x_fmt), d.x_plane_b, r); };
  ddb_block_fz<THREADS, G, F>(ddbz_load_rec(cx.ddb + d.bidx), cx.stream, cx.step, lds + DDB_LDS_B / 4, tid, dbg, lds + XCOPY_B / 4, 2 * G, put_y, c.w);
  if (FZ_PROF && cx.prof && tid == 0) {
    constexpr int slot[6] = {0, 5, 6, 1, 2, 3};
#pragma unroll
    for (int k = 0; k < 6; ++k) cx.prof[kNumOps + 1 + 8 * I + slot[k]] = dbg[k + 1];
  }
}
#endif

// 64 -> 16 (ReLU) -> 64 perceptron of the CTFA gates, evaluated by ONE wave without barriers and without LDS (lane c = channel c).
// First layer: lane (u = lane & 15, q = lane >> 4) forms the partial sum of hidden unit u over the 16 input channels of row q -- those
// channels ARE the 16 lanes of its own row, so DPP row_share hands them over inside the FMA's operand fetch (w1x: the lane's 16 weights
// w1[u][16 q + i], the blob stores them lane-major) -- and two lane swaps add the four rows.  Second layer: hidden unit k sits in lane k
// of every row; row_share again.  (Rounds 2-4 went through LDS for the first layer: products written [channel][unit], read back
// [unit][channel] -- two LDS round trips in the middle of a dependent chain, twice per CTFA.)
__device__ __forceinline__ float gate_mlp(float in, const f32x4 (&w1x)[4], float b1_u, const f32x4 (&w2)[4], float b2) {
  float h0 = 0.f, h1 = 0.f;
  sfor<8>([&](auto kk) {
    constexpr int k = 2 * decltype(kk)::value;
    h0 = fmaf(w1x[k >> 2][k & 3], dpp_mov<0x150 + k>(in), h0);
    h1 = fmaf(w1x[(k + 1) >> 2][(k + 1) & 3], dpp_mov<0x150 + k + 1>(in), h1);
  });
  const float hsum = xor32_sum(xor16_sum(h0 + h1));
  const float hid = fmaxf(hsum + b1_u, 0.f);
  float a0 = b2, a1 = 0.f;
  sfor<8>([&](auto kk) {
    constexpr int k = 2 * decltype(kk)::value;
    a0 = fmaf(w2[k >> 2][k & 3], dpp_mov<0x150 + k>(hid), a0);
    a1 = fmaf(w2[(k + 1) >> 2][(k + 1) & 3], dpp_mov<0x150 + k + 1>(hid), a1);
  });
  return a0 + a1;
}

// gate perceptrons of CTFA op J (the wave that evaluates them -- wave gi for stream slot gi -- holds, per lane, 16 weights of each of the four
// matrices): g[0..3] ta first layer (lane-major), [4..7] ta second layer, [8..11] / [12..15] the same for fa, g[16] = (b1t, b2t, b1f, b2f)
template <int J>
__device__ __forceinline__ void gates_load(const Ctx& cx, int tid, f32x4 (&g)[17]) {
  constexpr OpD d = kOps[J];
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (wave < d.gs) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      g[q] = ldb(cx.wb, static_cast<unsigned>((d.cw_off + lane * 16 + 4 * q) * 4));
      g[4 + q] = ldb(cx.wb, static_cast<unsigned>((d.cw_off + 1040 + lane * 16 + 4 * q) * 4));
      g[8 + q] = ldb(cx.wb, static_cast<unsigned>((d.cw_off + 2128 + lane * 16 + 4 * q) * 4));
      g[12 + q] = ldb(cx.wb, static_cast<unsigned>((d.cw_off + 3168 + lane * 16 + 4 * q) * 4));
    }
    g[16][0] = ldb1(cx.wb, static_cast<unsigned>((d.cw_off + 1024 + (lane & 15)) * 4));
    g[16][1] = ldb1(cx.wb, static_cast<unsigned>((d.cw_off + 2064 + lane) * 4));
    g[16][2] = ldb1(cx.wb, static_cast<unsigned>((d.cw_off + 2128 + 1024 + (lane & 15)) * 4));
    g[16][3] = ldb1(cx.wb, static_cast<unsigned>((d.cw_off + 2128 + 2064 + lane) * 4));
  } else {
    opaque_regs(g);
  }
}

// ---- CTFA gate + residual (ctfa_rt, proposed.py:162-196), in place on the next image; the network's last
//      one also applies the output 1x1 conv (proposed.py:65) and writes the enhanced magnitudes ---------------------------
template <int I>
__device__ __forceinline__ void ctfa_op(const Ctx& cx, int tid, Carry<I>& c, Carry<I + 1>& n) {
  constexpr OpD d = kOps[I];
  constexpr int NI = ctfa_ni(d), GS = d.gs, SGB = d.scr_gstride_b;
  // scratch of stream slot gi at d.scr_b + gi * SGB: column sums of the 8 waves | gates | the perceptrons' exchange
  constexpr int PART = d.scr_b, GATE = d.scr_b + 512 * 4;
  const int c4 = tid & 15, rg = tid >> 4, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // gate perceptrons (wave gi for stream slot gi -- wave 0 in a one-stream plan: lane c holds its 16 input / output weights of each
  // of the four matrices): requested now, used after the column sums.  (In the carry they would cost every wave of the preceding
  // sub-pixel conv 64 registers.)
  float b1t = 0.f, b1f = 0.f, b2t = 0.f, b2f = 0.f, ta_prev = 0.f;
  f32x4 w1t[4], w2t[4], w1f[4], w2f[4];
  {
    f32x4 g[17];
    if constexpr (early_gates(I)) {
#pragma unroll
      for (int k = 0; k < 17; ++k) g[k] = c.cg[k];
    } else {
      gates_load<I>(cx, tid, g);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) { w1t[q] = g[q]; w2t[q] = g[4 + q]; w1f[q] = g[8 + q]; w2f[q] = g[12 + q]; }
    b1t = g[16][0]; b2t = g[16][1]; b1f = g[16][2]; b2f = g[16][3];
  }
  if (wave < GS) {
    // frequency branch: the sum of the time attention over the 31 frames before this one (causal32 mode; a vector of zeros in the
    // default frame mode, where the branch sees TA / 32 -- SURVEY F7), stage d.bidx of this wave's stream
    ta_prev = ldb1(cx.ta_sum, static_cast<unsigned>(((cx.stream + d.g0 + wave) * cx.ta_sum_sstride + d.bidx * cx.ta_sum_gstride + lane) * 4));
  }
  const f32x4 ow = c.w[NI * GS];
  const float ob = c.w[NI * GS + 1][0];
  constexpr bool PRE = feeds_ctfa_sums(I - 1);      // the column sums came with the rows: eight partial rows left by the conv op before
  constexpr int PARTR = PRE ? d.scr_b + CSUM_OFF_B : PART;
  if constexpr (!PRE) {
#pragma unroll
    for (int gi = 0; gi < GS; ++gi) {
      f32x4 s4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const int f = rg + 32 * i;
        if (f < d.F) s4 += fwd_ld4g<I>(gi, f, 4 * c4);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        s4[e] = xor32_sum(xor16_sum(s4[e]));
      }
      if (lane < 16) lds4(PART + gi * SGB + (wave * 64 + 4 * c4) * 4) = s4;
    }
    FZ_STAMP(I, 5);
    lds_barrier();
  } else {
    FZ_STAMP(I, 5);
  }
  FZ_STAMP(I, 1);
  if (wave < GS) {
    const int sb = wave * SGB;          // this wave's stream slot
    float m = 0.f;
#pragma unroll
    for (int r = 0; r < 8; ++r) m += lds1(PARTR + sb + (r * 64 + lane) * 4);
    m = m * (1.0f / d.F);
    const float ta = fast_sigmoid(gate_mlp(m, w1t, b1t, w2t, b2t));
    FZ_STAMP(I, 2);
    // (frame mode: ta_prev = 0 and the ring is one dump row -- proposed.py:179-183 with T = 1; causal32: the offline model's 32-frame
    //  average, proposed.py:143-147, the history in a ring of 32 rows per stage outside the arena, engine.cpp nutls_set_ctfa_mode)
    const float fa = fast_sigmoid(gate_mlp((ta_prev + ta) * (1.0f / 32.0f), w1f, b1f, w2f, b2f));
    stb1(cx.ta_ring, static_cast<unsigned>(((cx.stream + d.g0 + wave) * cx.ta_ring_sstride + d.bidx * cx.ta_ring_gstride + lane) * 4), ta);
    lds1(GATE + sb + lane * 4) = fa * ta;
  }
  FZ_STAMP(I, 3);
  lds_barrier();
  FZ_STAMP(I, 4);
#pragma unroll
  for (int gi = 0; gi < GS; ++gi) {
    const f32x4 g4 = lds4(GATE + gi * SGB + 16 * c4);
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int f = rg + 32 * i;
      if (f < d.F) {
        const f32x4 x = fwd_ld4g<I>(gi, f, 4 * c4);      // (re-read: cheaper than 32 registers held across the gates)
        const f32x4 y = x * g4 + c.w[gi * NI + i];
        FZ_TRACE4(cx, d.g0 + gi, 1 + d.bidx, f, 64, 4 * c4, y);
        if constexpr (d.last) {
          const float s = group_sum<16>(y[0] * ow[0] + y[1] * ow[1] + y[2] * ow[2] + y[3] * ow[3]);
          if (c4 == 0) cx.io_out[(d.g0 + gi) * 256 + f] = s + ob;
        } else {
          fwd_st4g<I>(gi, f, 4 * c4, y);
          // (packed plans: a consumer that is not the next op of this stream reads the rows from HBM)
          if constexpr (d.d0_on) st_state<d.nt0 != 0>(d.d0_src == S_CUR ? cx.sbc : cx.sbs, static_cast<unsigned>((d.d0_off + f * d.d0_ld + 4 * c4) * 4) + gofs(cx, d.g0 + gi), y);
        }
      }
    }
  }
}

// ---- one op -----------------------------------------------------------------------------------------------------------
template <int I, bool PROF>
__device__ __forceinline__ void run_op(const Ctx& cx, Carry<I>& c, Carry<I + 1>& n) {
  constexpr OpD d = kOps[I];
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));          // per-op thread id: nothing derived from it is hoisted across ops
  // (Tried in round 4: the index rebuilt per op from a scalar wave index + a volatile lane count, so that threadIdx.x is not live across
  //  all ops -- the 4-stream build spills it and re-loads it in every op prologue.  200 scratch loads fewer there, no change in its step
  //  time; the one-stream kernel 1 % slower (0.3777 -> 0.3813 ms, four alternating runs on one box).  Dropped.)
  if (PROF && cx.prof && tid == 0) cx.prof[I] = wall_clock64();
  FZ_WSTAMP(I, 0);
  // Drain point of an LSTM / CTFA op: every wave waits for its own earlier HBM stores BEFORE it issues this op's loads (a
  // wait at the end of the op would also wait for those loads -- a full HBM round trip per drain point); the op's own
  // barriers then order the completed stores before every load a later op issues (the planner's hand-off rule).
  constexpr bool DRAIN_FIRST = d.drain && (d.type == T_LSTM || d.type == T_CTFA);
  if constexpr (DRAIN_FIRST) drain_vm();
  FZ_WSTAMP(I, 2);
  // loads in the order they are needed: parts this op stores itself, its epilogue parameters, what the next op needs first
  f32x4 p1[cmax(1, nxt_regs(I, 1))];
  f32x4 p3[cmax(1, own_regs(I, 3))];
  f32x4 wx[cmax(1, ext_sf(I))];
  ext_load<I>(cx, tid, wx);             // (before the staging loads: a wait for these must not wait for those)
  constexpr int ST = stg_threads(I);
  if constexpr (ST == THREADS) {
    stage_load<nxt_of(I), 1, THREADS>(cx, tid, p1);
    stage_load<I, 3, THREADS>(cx, tid, p3);
  } else {
    if (__builtin_amdgcn_readfirstlane(tid >> 6) >= (THREADS - ST) / 64) {
      stage_load<nxt_of(I), 1, ST>(cx, tid - (THREADS - ST), p1);
      stage_load<I, 3, ST>(cx, tid - (THREADS - ST), p3);
    }
  }
  late_loads<I>(cx, tid, n);
#pragma unroll
  for (int k = 0; k < cmax(1, carry_w(I + 1)); ++k) n.w[k] = c.w2[k];
  n.prm = c.prm2;
#pragma unroll
  for (int k = 0; k < cmax(1, yp_regs(I + 1)); ++k) n.yp[k] = c.yp2[k];
  sched_pin();
  FZ_STAMP(I, 0);
  FZ_WSTAMP(I, 1);

  if constexpr (d.type == T_INPUT) {
    input_op<I>(cx, tid);
    build_next<I>(tid, p1, c.p);
  } else if constexpr (d.type == T_CONV) {
    if constexpr (d.path == P_X16B) conv_x16b<I>(cx, tid, c, p1, p3);
    else conv_r32b<I>(cx, tid, c, p1, p3, wx);
  } else if constexpr (d.type == T_LSTM) {
    lstm_op<I>(cx, tid, c, n);
#if FZ_BASE
  } else if constexpr (d.type == T_DDB) {
    ddb_op<I>(cx, tid, c);
#endif
  } else {
    ctfa_op<I>(cx, tid, c, n);
    if constexpr (nxt_of(I) >= 0) {
      // packed plans: an instance of the network's last CTFA that is followed by another stream's ops completes the image of the conv
      // after it (its loads went out in the prologue above) -- once every thread is done with the plain rows the image overlaps
      lds_barrier();
      build_next<I>(tid, p1, c.p);
    }
  }
  if constexpr (d.drain && !DRAIN_FIRST) drain_vm();
  if constexpr (early_gates(I + 1)) gates_load<I + 1>(cx, tid, n.cg);      // (the next op is a CTFA that starts at its gate perceptrons)
  FZ_WSTAMP(I, 9);
  lds_barrier();
  FZ_WSTAMP(I, 10);
}

template <int I, bool PROF>
__device__ __forceinline__ void run_from(const Ctx& cx, Carry<I>& c) {
  if constexpr (I < kNumOps) {
#if FZ_STOPAT
    if (cx.stop_at == I) return;      // (wave-uniform: one scalar compare per op)
#endif
    Carry<I + 1> n;
    run_op<I, PROF>(cx, c, n);
    run_from<I + 1, PROF>(cx, n);
  }
}

struct FzArgs {
  float* arena; long long sstride; const float* blob; const float* io_in; float* io_out; int B, par; unsigned long long* prof;
  const DdbParams* ddb;      // baseline variant: table [2 parities][13] (engine.cpp push_ddb), else null
  int step;                  // baseline variant: frames processed so far (position in the dilated-dense history rings)
  FzTa ta;                   // CTFA frequency branch: see nutls_internal.hpp
};

#if FZ_STREAMS == 2
#define FZ_KERNEL nutls_fused_step_g2_kernel
#define FZ_LAUNCH launch_fused_step_g2
#define FZ_ATTR fused_step_g2_set_attributes
#define FZ_NO_PROF_TWIN 1
#elif FZ_STREAMS == 4 && FZ_PROF
// (optional translation unit fused_step_g4_prof.hip, built with NUTLS_BUILD_G4_PROF=1: the packed kernel with op-boundary stamps)
#define FZ_KERNEL nutls_fused_step_g4_prof_kernel
#define FZ_LAUNCH launch_fused_step_g4_prof
#define FZ_ATTR fused_step_g4_prof_set_attributes
#elif FZ_STREAMS == 4
#define FZ_KERNEL nutls_fused_step_g4_kernel
#define FZ_LAUNCH launch_fused_step_g4
#define FZ_ATTR fused_step_g4_set_attributes
#define FZ_NO_PROF_TWIN 1
#define FZ_OPT_PROF_LAUNCH launch_fused_step_g4_prof
#define FZ_OPT_PROF_ATTR fused_step_g4_prof_set_attributes
#elif defined(FZ_STOP_TWIN)
#define FZ_KERNEL nutls_fused_step_stop_kernel
#define FZ_LAUNCH launch_fused_step_stop
#define FZ_ATTR fused_step_stop_set_attributes
#define FZ_NO_PROF_TWIN 1
#elif FZ_PROF && FZ_BASE
#define FZ_KERNEL nutls_fused_base_step_prof_kernel
#define FZ_LAUNCH launch_fused_base_step_prof
#define FZ_ATTR fused_base_step_prof_set_attributes
#elif FZ_PROF
#define FZ_KERNEL nutls_fused_step_prof_kernel
#define FZ_LAUNCH launch_fused_step_prof
#define FZ_ATTR fused_step_prof_set_attributes
#elif FZ_BASE
#define FZ_KERNEL nutls_fused_base_step_kernel
#define FZ_LAUNCH launch_fused_base_step
#define FZ_ATTR fused_base_step_set_attributes
#define FZ_LAUNCH_PROF launch_fused_base_step_prof
#define FZ_ATTR_PROF fused_base_step_prof_set_attributes
#else
#define FZ_KERNEL nutls_fused_step_kernel
#define FZ_LAUNCH launch_fused_step
#define FZ_ATTR fused_step_set_attributes
#define FZ_LAUNCH_PROF launch_fused_step_prof
#define FZ_ATTR_PROF fused_step_prof_set_attributes
#endif
__global__ __launch_bounds__(THREADS) void FZ_KERNEL(const FzArgs a) {
  constexpr bool PROF = FZ_PROF != 0;
  // One workgroup per stream, grid = B (the hardware queues the workgroups that do not fit).  No loop over streams here:
  // everything that depends only on kernel arguments (hundreds of `blob + offset` bases) would be hoisted out of such a
  // loop to the kernel entry, spilled, and re-loaded from scratch in the op prologues -- behind a full vmcnt(0) drain.
  // (packed plans: the workgroup owns streams kStreams * blockIdx.x .. + kStreams - 1; the host uses them only for B a multiple of kStreams)
  const int stream = blockIdx.x * NSTREAMS;
  if (stream >= a.B) return;
  const float* slice = a.arena + static_cast<size_t>(stream) * a.sstride;
  Ctx cx;
  cx.sbs = (gcb_t)(unsigned long long)slice;
  cx.sbc = (gcb_t)(unsigned long long)(slice + (a.par ? kParityStride : 0));
  cx.sbp = (gcb_t)(unsigned long long)(slice + (a.par ? 0 : kParityStride));
  cx.wb = (gcb_t)(unsigned long long)a.blob;
  cx.ysw = (gcb_t)(unsigned long long)(slice + kYsOff + (a.par ? kYsBlock : 0));
  cx.ysr = (gcb_t)(unsigned long long)(slice + kYsOff + (a.par ? 0 : kYsBlock));
  cx.io_in = (gcf_t)(unsigned long long)(a.io_in + static_cast<size_t>(stream) * 256);
  cx.io_out = (gf_t)(unsigned long long)(a.io_out + static_cast<size_t>(stream) * 256);
  cx.prof = (PROF && blockIdx.x == 0) ? a.prof : nullptr;          // (packed builds: workgroup 0 = streams 0 .. kStreams - 1)
  cx.ddb = a.ddb ? a.ddb + a.par * 13 : nullptr;
  cx.stream = stream;
  cx.step = a.step;
  cx.sstride_b = static_cast<unsigned>(a.sstride * 4);
  cx.ta_sum = (gcb_t)(unsigned long long)a.ta.sum;
  cx.ta_ring = (gcb_t)(unsigned long long)a.ta.ring;
  cx.ta_sum_sstride = a.ta.sum_sstride;
  cx.ta_sum_gstride = a.ta.sum_gstride;
  cx.ta_ring_sstride = a.ta.ring_sstride;
  cx.ta_ring_gstride = a.ta.ring_gstride;
  cx.eager = a.ta.eager;
  cx.dbg = (PROF && a.ta.dbg) ? (gcb_t)(unsigned long long)(a.ta.dbg + static_cast<size_t>(stream) * a.ta.dbg_sstride) : nullptr;
  cx.dbg_sstride_b = static_cast<unsigned>(a.ta.dbg_sstride * 4);
#if FZ_BASE
  {
    // one pass over the 13 parameter records of the dilated-dense ops: they stay in the scalar cache for the rest of
    // the launch (ddbz_load_rec reads them with scalar loads, twice per op)
    // (ONE statement, one destination register, the wait inside: a scalar load lands long after it was issued, and a destination the
    //  compiler considers dead after the statement is free for re-use -- the C++ loop this replaces let late loads overwrite whatever
    //  the allocator had put there since, harmlessly until the allocation changed: a base pointer, a memory fault)
    unsigned t;
    asm volatile(
        ".set .Lddb_warm, 0\n\t"
        ".rept %2\n\t"
        "s_load_dword %0, %1, .Lddb_warm\n\t"
        ".set .Lddb_warm, .Lddb_warm + 64\n\t"
        ".endr\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&s"(t) : "s"(cx.ddb), "n"((13 * sizeof(DdbParams) + 63) / 64) : "memory");
  }
#endif
#if FZ_STOPAT
  cx.stop_at = a.ta.skew;
  if (false) {
#else
  if (a.ta.skew > 0) {
#endif
    // start skew (FzTa::skew): every workgroup of a launch walks the same op sequence, so their staging bursts hit HBM together
    const int n = (blockIdx.x & 3) * a.ta.skew;
    for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(1);
  }
  Carry<0> c0;
  {
    int tid = threadIdx.x;
    prefetch_w<1>(cx, tid, c0.w2, c0.prm2);
    prefetch_y<1>(cx, tid, c0.yp2);
  }
  run_from<0, PROF>(cx, c0);
  if (PROF && cx.prof && threadIdx.x == 0) cx.prof[kNumOps] = wall_clock64();
}

}  // namespace fz

#if FZ_PROF
hipError_t FZ_LAUNCH(float* arena, long long sstride, const float* blob, const float* io_in, float* io_out, int B, int par,
                     unsigned long long* prof, const DdbParams* ddb, int step, int grid, hipStream_t s, const FzTa& ta) {
  fz::FzArgs a{arena, sstride, blob, io_in, io_out, B, par, prof, ddb, step, ta};
  hipLaunchKernelGGL(fz::FZ_KERNEL, dim3(grid), dim3(fz::THREADS), fz::LDS_BYTES, s, a);
  return hipGetLastError();
}
hipError_t FZ_ATTR() {
  return hipFuncSetAttribute(reinterpret_cast<const void*>(fz::FZ_KERNEL), hipFuncAttributeMaxDynamicSharedMemorySize, fz::LDS_BYTES);
}
#elif defined(FZ_NO_PROF_TWIN)
// packed builds: `grid` workgroups of kStreams streams each.  A profiling twin exists only as an optional translation unit (weak
// reference: null when it was not built -- `prof` then yields hipErrorNotSupported)
#ifdef FZ_OPT_PROF_LAUNCH
hipError_t FZ_OPT_PROF_LAUNCH(float* arena, long long sstride, const float* blob, const float* io_in, float* io_out, int B, int par,
                              unsigned long long* prof, const DdbParams* ddb, int step, int grid, hipStream_t s, const FzTa& ta) __attribute__((weak));
hipError_t FZ_OPT_PROF_ATTR() __attribute__((weak));
#endif
hipError_t FZ_LAUNCH(float* arena, long long sstride, const float* blob, const float* io_in, float* io_out, int B, int par,
                     unsigned long long* prof, const DdbParams* ddb, int step, int grid, hipStream_t s, const FzTa& ta) {
  if (prof) {
#ifdef FZ_OPT_PROF_LAUNCH
    if (FZ_OPT_PROF_LAUNCH) return FZ_OPT_PROF_LAUNCH(arena, sstride, blob, io_in, io_out, B, par, prof, ddb, step, grid, s, ta);
#endif
    return hipErrorNotSupported;
  }
  fz::FzArgs a{arena, sstride, blob, io_in, io_out, B, par, nullptr, ddb, step, ta};
  hipLaunchKernelGGL(fz::FZ_KERNEL, dim3(grid), dim3(fz::THREADS), fz::LDS_BYTES, s, a);
  return hipGetLastError();
}
hipError_t FZ_ATTR() {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fz::FZ_KERNEL), hipFuncAttributeMaxDynamicSharedMemorySize, fz::LDS_BYTES);
#ifdef FZ_OPT_PROF_ATTR
  if (e == hipSuccess && FZ_OPT_PROF_ATTR) e = FZ_OPT_PROF_ATTR();
#endif
  return e;
}
#else
hipError_t FZ_LAUNCH_PROF(float* arena, long long sstride, const float* blob, const float* io_in, float* io_out, int B, int par,
                          unsigned long long* prof, const DdbParams* ddb, int step, int grid, hipStream_t s, const FzTa& ta);
hipError_t FZ_ATTR_PROF();
hipError_t FZ_LAUNCH(float* arena, long long sstride, const float* blob, const float* io_in, float* io_out, int B, int par,
                     unsigned long long* prof, const DdbParams* ddb, int step, int grid, hipStream_t s, const FzTa& ta) {
  if (prof) return FZ_LAUNCH_PROF(arena, sstride, blob, io_in, io_out, B, par, prof, ddb, step, grid, s, ta);
  fz::FzArgs a{arena, sstride, blob, io_in, io_out, B, par, nullptr, ddb, step, ta};
  hipLaunchKernelGGL(fz::FZ_KERNEL, dim3(grid), dim3(fz::THREADS), fz::LDS_BYTES, s, a);
  return hipGetLastError();
}
hipError_t FZ_ATTR() {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fz::FZ_KERNEL), hipFuncAttributeMaxDynamicSharedMemorySize, fz::LDS_BYTES);
  return e != hipSuccess ? e : FZ_ATTR_PROF();
}
#endif

}  // namespace nutls
