// NUTLSW01 weight container parser + MFMA-order weight packing (host side).
#include <cstring>

#include "nutls_internal.hpp"

namespace nutls {

namespace {
struct Cursor {
  const uint8_t* p;
  size_t n, off = 0;
  bool take(void* dst, size_t k) {
    if (off + k > n) return false;
    std::memcpy(dst, p + off, k);
    off += k;
    return true;
  }
};
}  // namespace

bool parse_weight_blob(const void* blob, size_t n, WeightMap* out, std::string* err) {
  Cursor c{static_cast<const uint8_t*>(blob), n};
  char magic[8];
  if (!c.take(magic, 8) || std::memcmp(magic, "NUTLSW01", 8) != 0) {
    *err = "not a NUTLSW01 weight container";
    return false;
  }
  uint32_t count = 0;
  if (!c.take(&count, 4)) { *err = "truncated header"; return false; }
  for (uint32_t i = 0; i < count; ++i) {
    uint16_t nl = 0;
    if (!c.take(&nl, 2) || c.off + nl > n) { *err = "truncated tensor name"; return false; }
    std::string name(reinterpret_cast<const char*>(c.p + c.off), nl);
    c.off += nl;
    uint8_t dtype = 0, ndim = 0;
    if (!c.take(&dtype, 1) || !c.take(&ndim, 1) || ndim == 0 || ndim > 6) { *err = "bad tensor header: " + name; return false; }
    HostTensor t;
    size_t cnt = 1;
    for (int d = 0; d < ndim; ++d) {
      uint32_t v = 0;
      if (!c.take(&v, 4)) { *err = "truncated dims: " + name; return false; }
      // every dim positive, and the element count can never exceed the bytes that are left (no overflow, no huge resize)
      if (v == 0 || v > n || cnt > n / v) { *err = "bad dims: " + name; return false; }
      t.dims.push_back(static_cast<int>(v));
      cnt *= v;
    }
    uint32_t ns = 0;
    if (!c.take(&ns, 4)) { *err = "truncated scales: " + name; return false; }
    if (static_cast<size_t>(ns) > (n - c.off) / 4) { *err = "truncated scales: " + name; return false; }
    std::vector<float> scales(ns);
    if (ns && !c.take(scales.data(), 4ull * ns)) { *err = "truncated scales: " + name; return false; }
    if (cnt > (n - c.off) / (dtype == 0 ? 4 : 1)) { *err = "truncated payload: " + name; return false; }
    t.data.resize(cnt);
    if (dtype == 0) {
      if (!c.take(t.data.data(), 4 * cnt)) { *err = "truncated payload: " + name; return false; }
    } else if (dtype == 1) {
      if (c.off + cnt > n) { *err = "truncated payload: " + name; return false; }
      if (!(ns == 1 || ns == static_cast<uint32_t>(t.dims[0]))) { *err = "bad scale count: " + name; return false; }
      const int8_t* q = reinterpret_cast<const int8_t*>(c.p + c.off);
      const size_t inner = cnt / t.dims[0];
      for (size_t k = 0; k < cnt; ++k) t.data[k] = static_cast<float>(q[k]) * scales[ns == 1 ? 0 : k / inner];
      t.q.assign(q, q + cnt);
      t.scales = scales;
      c.off += cnt + ((4 - cnt % 4) % 4);
    } else {
      *err = "unknown dtype code in " + name;
      return false;
    }
    (*out)[name] = std::move(t);
  }
  if (c.off != n) { *err = "trailing bytes in weight container"; return false; }
  return true;
}

// Layout produced (float index), matching the streaming order of conv_mfma_kernel:
//   for t in [0,tt) for chunk in [0,cin/CC) for tap in taps_per_t for g in [0,CC/8) for nt for lane for j:
//     n'  = nt*32 + (lane & 31)
//     c   = chunk*CC + g*8 + 4*(lane >> 5) + j
//     val = W[perm[n']][t][tap.kw][c]          (tap.first = source time index, tap.second = kw)
// A tap with kw < 0 means "no source tap" (zeros).
std::vector<float> pack_conv_weights(const HostTensor& w, const std::vector<int>& perm,
                                     const std::vector<std::pair<int, int>>& taps_per_t,
                                     int tt, int cin, int nt) {
  const int th = w.dims[1], kw = w.dims[2], wc = w.dims[3];
  const int cc = cin < 64 ? cin : 64, nch = cin / cc, kf = static_cast<int>(taps_per_t.size());
  std::vector<float> out(static_cast<size_t>(tt) * nch * kf * (cc / 8) * nt * 64 * 4);
  size_t o = 0;
  for (int t = 0; t < tt; ++t)
    for (int ch = 0; ch < nch; ++ch)
      for (int k = 0; k < kf; ++k)
        for (int g = 0; g < cc / 8; ++g)
          for (int n = 0; n < nt; ++n)
            for (int lane = 0; lane < 64; ++lane)
              for (int j = 0; j < 4; ++j) {
                const int np = n * 32 + (lane & 31);
                const int c = ch * cc + g * 8 + 4 * (lane >> 5) + j;
                const int src_t = (tt == 1) ? taps_per_t[k].first : t;
                const int src_k = taps_per_t[k].second;
                float v = 0.f;
                if (src_k >= 0 && np < static_cast<int>(perm.size()) && perm[np] >= 0)
                  v = w.data[((static_cast<size_t>(perm[np]) * th + src_t) * kw + src_k) * wc + c];
                out[o++] = v;
              }
  return out;
}

// The int8 payload of the container widened to bf16 (exact), in the streaming order of conv_bf16x3_kernel:
//   for t for chunk for tap for g in [0,CC/16) for nt for lane for j in [0,8):
//     n' = nt*32 + (lane & 31),  c = chunk*CC + g*16 + 8*(lane >> 5) + j,  value = bf16(Q[perm[n']][t][tap.kw][c])
// Returned as floats holding two bf16 each (empty if the tensor has no int8 payload).
std::vector<float> pack_conv_weights_bf16(const HostTensor& w, const std::vector<int>& perm,
                                          const std::vector<std::pair<int, int>>& taps_per_t, int tt, int cin, int nt) {
  if (w.q.size() != w.data.size() || w.q.empty()) return {};
  const int th = w.dims[1], kw = w.dims[2], wc = w.dims[3];
  const int cc = cin < 64 ? cin : 64, nch = cin / cc, kf = static_cast<int>(taps_per_t.size());
  std::vector<uint16_t> out(static_cast<size_t>(tt) * nch * kf * (cc / 16) * nt * 64 * 8);
  size_t o = 0;
  for (int t = 0; t < tt; ++t)
    for (int ch = 0; ch < nch; ++ch)
      for (int k = 0; k < kf; ++k)
        for (int g = 0; g < cc / 16; ++g)
          for (int n = 0; n < nt; ++n)
            for (int lane = 0; lane < 64; ++lane)
              for (int j = 0; j < 8; ++j) {
                const int np = n * 32 + (lane & 31);
                const int c = ch * cc + g * 16 + 8 * (lane >> 5) + j;
                const int src_t = (tt == 1) ? taps_per_t[k].first : t;
                const int src_k = taps_per_t[k].second;
                float v = 0.f;
                if (src_k >= 0 && np < static_cast<int>(perm.size()) && perm[np] >= 0)
                  v = static_cast<float>(w.q[((static_cast<size_t>(perm[np]) * th + src_t) * kw + src_k) * wc + c]);
                uint32_t bits;
                std::memcpy(&bits, &v, 4);            // |q| <= 128: exact in bf16
                out[o++] = static_cast<uint16_t>(bits >> 16);
              }
  std::vector<float> packed(out.size() / 2);
  std::memcpy(packed.data(), out.data(), out.size() * 2);
  return packed;
}

std::vector<float> pack_conv_weights16(const HostTensor& w, const std::vector<int>& perm,
                                       const std::vector<std::pair<int, int>>& taps_per_t,
                                       int tt, int cin, int nt32) {
  const int th = w.dims[1], kw = w.dims[2], wc = w.dims[3];
  const int cc = cin < 64 ? cin : 64, nch = cin / cc, kf = static_cast<int>(taps_per_t.size());
  const int nrt = 2 * nt32;
  std::vector<float> out(static_cast<size_t>(tt) * nch * kf * (cc / 16) * nrt * 64 * 4);
  size_t o = 0;
  for (int t = 0; t < tt; ++t)
    for (int ch = 0; ch < nch; ++ch)
      for (int k = 0; k < kf; ++k)
        for (int g = 0; g < cc / 16; ++g)
          for (int rt = 0; rt < nrt; ++rt)
            for (int lane = 0; lane < 64; ++lane)
              for (int j = 0; j < 4; ++j) {
                const int np = rt * 16 + (lane & 15);
                const int c = ch * cc + g * 16 + 4 * (lane >> 4) + j;
                const int src_t = (tt == 1) ? taps_per_t[k].first : t;
                const int src_k = taps_per_t[k].second;
                float v = 0.f;
                if (src_k >= 0 && np < static_cast<int>(perm.size()) && perm[np] >= 0)
                  v = w.data[((static_cast<size_t>(perm[np]) * th + src_t) * kw + src_k) * wc + c];
                out[o++] = v;
              }
  return out;
}

static int ilog2_exact(int v) {
  int l = 0;
  while ((1 << l) < v) ++l;
  return l;
}

ConvPlan make_conv_plan(ConvKind k, const ConvParams& p) {
  const ConvShape sh = conv_shape(k);
  ConvPlan c{};
  c.cin = sh.cin; c.nt = sh.nt; c.stride = sh.stride; c.tt = sh.tt; c.kf = sh.kf; c.padl = sh.padl; c.epi_ln = sh.epi_ln; c.g = sh.g;
  c.cc = sh.cin < 64 ? sh.cin : 64;
  c.cc4_shift = ilog2_exact(c.cc / 4);
  c.pitch = sh.stride == 1 ? c.cc + 4 : 2 * c.cc + 4;
  c.rows = sh.stride == 1 ? p.F_out + sh.kf - 1 : 2 * (p.F_out + (sh.kf - 1) / 2);
  c.vrows = (c.rows - sh.padl < p.F_in) ? c.rows - sh.padl : p.F_in;
  c.n4p_shift = ilog2_exact(c.vrows * (c.cc / 4));
  const int nch = sh.cin / c.cc;
  c.nch_shift = ilog2_exact(nch);
  c.nph = sh.tt * nch;
  c.phase_floats = (sh.stride == 1 ? c.rows : c.rows / 2) * c.pitch;
  c.merged = (c.nph * c.phase_floats <= MK_LDS_IN_FLOATS) && (c.nph * (c.vrows * (c.cc / 4)) <= MK_STAGE_ITEMS);
  const int nstage = c.merged ? c.nph : 1;
  c.rounds = c.nph / nstage;
  // the staging code splits an item index into lane bits and pass bits: a phase-by-phase layer must start every phase on a 512 boundary
  if (c.rounds > 1 && c.n4p_shift < 9) throw std::runtime_error("conv plan: unmerged layer with < 512 items per phase");
  c.gpc = c.cc / 8;
  c.RG = nstage * sh.kf * c.gpc;
  c.PT = (p.F_out + 31) / 32;
  c.tiles = c.PT * sh.nt;
  c.nt_shift = ilog2_exact(sh.nt);
  c.tw = c.tiles > MK_NWAVES ? 2 : 1;                 // 16-tile layers: two position tiles per wave
  c.tasks = c.tiles / c.tw;
  c.tasks_shift = ilog2_exact(c.tasks);
  // K split: the largest slice count that keeps <= 8 wave tasks with whole 4-group chunks per slice
  int best = 1;
  for (int ks = 1; ks <= MK_NWAVES / c.tasks; ++ks) {
    if (c.RG % ks) continue;
    const int gpk = c.RG / ks;
    if (gpk % 4) continue;      // whole 4-group chunks; chunks never straddle a frequency-tap segment (gpc is 4 or 8)
    best = ks;
  }
  c.KS = best;
  c.gpk = c.RG / best;
  c.opitch = 32 * sh.nt + 4;
  c.slot_floats = c.PT * 32 * c.opitch;
  c.R = sh.nt / sh.g;
  c.lpg = 8 * sh.g;
  return c;
}

void apply_s16_plan(ConvPlan* cp, const ConvParams& p) {
  ConvPlan& c = *cp;
  if (p.F_out > 16 || !c.merged || c.rounds != 1 || !p.wpk16) return;
  const int nt16 = 2 * c.nt;
  if (nt16 > MK_NWAVES) return;
  const int gpc16 = c.cc / 16;
  const int RG = c.nph * c.kf * gpc16;
  int best = 0;
  for (int ks = 1; ks <= MK_NWAVES / nt16; ++ks)
    if (RG % ks == 0 && (RG / ks) % 4 == 0) best = ks;      // whole 4-fragment chunks per slice (two pairs; a pair never straddles a tap segment)
  if (!best) return;
  c.s16 = 1;
  c.nt = nt16;
  c.nt_shift = ilog2_exact(nt16);
  c.gpc = gpc16;
  c.RG = RG;
  c.PT = 1;
  c.tiles = nt16;
  c.tw = 1;
  c.tasks = nt16;
  c.tasks_shift = c.nt_shift;
  c.KS = best;
  c.gpk = RG / best;
}

static uint32_t off_of(const float* p, const float* base) {
  return p ? static_cast<uint32_t>(p - base) : MK_NULL_OFF;
}
static uint32_t pack4(int a, int b, int c, int d) {
  return (static_cast<uint32_t>(a) & 255u) | ((static_cast<uint32_t>(b) & 255u) << 8) | ((static_cast<uint32_t>(c) & 255u) << 16) |
         ((static_cast<uint32_t>(d) & 255u) << 24);
}
static uint32_t pack2(int lo, int hi) { return (static_cast<uint32_t>(lo) & 0xFFFFu) | (static_cast<uint32_t>(hi) << 16); }
static uint32_t fbits(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }

// Word layout (keep in sync with the decode_* functions of megakernel.hip); w[23] = op code.
CompactOp encode_op(const DevLaunch& d, const float* arena, const float* wbase) {
  CompactOp o{};
  o.w[23] = static_cast<uint32_t>(d.op);
  o.w[22] = pack4(d.nc_hand, d.nc_fwd_coff, 0, 0);     // non-conv ops: hand-off to the following conv layer (conv ops: see below)
  switch (d.op) {
    case DEV_OP_CONV: {
      // conv offsets are BYTES from the stream slice / the weight arena (decoders: cv_* in megakernel.hip)
      const ConvParams& p = d.conv;
      const ConvPlan& c = d.cp;
      auto ab = [&](const float* q) { return static_cast<uint32_t>((q - arena) * 4); };
      auto wbo = [&](const float* q) { return static_cast<uint32_t>((q - wbase) * 4); };
      o.w[0] = ab(p.src0);
      o.w[1] = p.src1 ? ab(p.src1) - ab(p.src0) : 0u;                   // second time tap relative to the first (mod 2^32)
      o.w[2] = ab(p.dst0); o.w[3] = p.dst1 ? ab(p.dst1) : 0u;
      o.w[4] = wbo(c.s16 ? p.wpk16 : p.wpk); o.w[5] = wbo(p.bias);
      o.w[6] = c.epi_ln ? wbo(p.gamma) : o.w[5];                        // no LayerNorm: the loads still happen, on the bias
      o.w[7] = c.epi_ln ? wbo(p.beta) : o.w[5];
      o.w[8] = fbits(p.alpha);
      const int gcode = c.g == 1 ? 0 : (c.g == 2 ? 1 : 2);
      const uint32_t flags = (p.row_mul & 3) | ((p.row_add & 1) << 2) | ((c.stride - 1) << 3) | ((c.tt - 1) << 4) | (c.epi_ln << 5) |
                             (c.merged << 6) | (c.staged_by_prev << 7) | ((c.hand_next ? 1 : 0) << 8) | ((c.fwd_sel ? 1 : 0) << 9) |
                             ((c.fwd_rmul == 2 ? 1 : 0) << 10) | ((c.fwd_radd & 1) << 11) | ((c.R == 2 ? 1 : 0) << 12) |
                             ((p.dst1 ? 1 : 0) << 13) | (static_cast<uint32_t>(gcode) << 14);
      o.w[9] = pack2(p.src_ld * 4, p.ld0 * 4);
      o.w[10] = pack2(p.ld1 * 4, static_cast<int>(flags));
      o.w[11] = pack2(p.F_in, p.F_out);
      o.w[12] = pack4(c.kf, c.padl, c.lpg, c.nt);
      o.w[13] = pack4(c.cc4_shift, c.n4p_shift, c.nch_shift, c.nt_shift);
      o.w[14] = pack2(c.pitch * 4, c.rows);
      o.w[15] = pack2(c.vrows, c.nph | (c.rounds << 8));
      o.w[16] = static_cast<uint32_t>(c.phase_floats) * 4u;
      // [17], [21], [22]: the hand-off loads of the next layer's image (0 without hand-off)
      o.w[17] = (c.hand_next && c.hx_src1) ? ab(c.hx_src1) - ab(c.hx_src0) : 0u;
      o.w[22] = c.hand_next ? ab(c.hx_src0) : 0u;
      o.w[18] = pack4(c.RG, c.KS, c.gpk, c.gpc);
      o.w[19] = pack4(c.tasks, c.tasks_shift, c.tw, c.fwd_coff4);
      o.w[20] = pack2(c.opitch * 4, p.F_out * c.R);
      o.w[21] = static_cast<uint32_t>(c.s16 & 1) | (static_cast<uint32_t>(c.hx_nhand & 7) << 1) | (static_cast<uint32_t>(c.hx_cc4_shift & 15) << 4) |
                (static_cast<uint32_t>(c.hx_n4p_shift & 15) << 8) | (static_cast<uint32_t>(c.hx_nch_shift & 15) << 12) |
                (static_cast<uint32_t>(c.hx_ld * 4) << 16);
      break;
    }
    case DEV_OP_LSTM: {
      const LstmParams& p = d.lstm;
      o.w[0] = off_of(p.x, arena); o.w[1] = pack2(p.x_ld, p.x_cols);
      o.w[2] = off_of(p.wxT, wbase); o.w[3] = off_of(p.whT, wbase); o.w[4] = off_of(p.bias, wbase);
      o.w[5] = off_of(p.wdT, wbase); o.w[6] = off_of(p.bd, wbase);
      o.w[7] = off_of(p.h_in, arena); o.w[8] = off_of(p.c_in, arena);
      o.w[9] = off_of(p.h_out, arena); o.w[10] = off_of(p.c_out, arena);
      o.w[11] = off_of(p.dst, arena); o.w[12] = pack2(p.dst_ld, p.dst_cols);
      o.w[13] = pack2(p.Din, p.Dout);
      break;
    }
    case DEV_OP_CTFA: {
      const CtfaParams& p = d.ctfa;
      o.w[0] = off_of(p.x, arena); o.w[1] = off_of(p.e0, arena); o.w[2] = off_of(p.y, arena);
      o.w[3] = pack4(p.x_ld, p.e0_ld, p.y_ld, 0);
      o.w[4] = off_of(p.ta_w1T, wbase); o.w[5] = off_of(p.ta_b1, wbase); o.w[6] = off_of(p.ta_w2T, wbase); o.w[7] = off_of(p.ta_b2, wbase);
      o.w[8] = off_of(p.fa_w1T, wbase); o.w[9] = off_of(p.fa_b1, wbase); o.w[10] = off_of(p.fa_w2T, wbase); o.w[11] = off_of(p.fa_b2, wbase);
      o.w[12] = static_cast<uint32_t>(p.F);
      o.w[13] = off_of(p.ta_w2, wbase); o.w[14] = off_of(p.fa_w2, wbase);
      break;
    }
    case DEV_OP_INLAYER: {
      const InLayerParams& p = d.inl;
      o.w[0] = off_of(p.y, arena);
      o.w[1] = off_of(p.w, wbase); o.w[2] = off_of(p.b, wbase); o.w[3] = off_of(p.gamma, wbase); o.w[4] = off_of(p.beta, wbase);
      o.w[5] = fbits(p.alpha);
      break;
    }
    case DEV_OP_DDB:
      o.w[0] = static_cast<uint32_t>(d.ddb_index);
      break;
    default: {
      const OutConvParams& p = d.outc;
      o.w[0] = off_of(p.x, arena); o.w[1] = static_cast<uint32_t>(p.x_ld);
      o.w[2] = off_of(p.w, wbase); o.w[3] = fbits(p.bias);
      break;
    }
  }
  return o;
}

}  // namespace nutls
