// Host side of the fused frame-step kernel (fused_step.hip): packs the weight blob in the order of the static
// plan (fused_plan_lstm.inc) and checks that the engine's arena layout is the one the plan addresses.
#include <cstring>

#include "nutls_internal.hpp"
#include "fused_plan.hpp"

namespace nutls {
namespace fz {
#include "fused_plan_lstm.inc"
}  // namespace fz

namespace {

const HostTensor* get(const WeightMap& w, const std::string& k, std::string* err) {
  auto it = w.find(k);
  if (it == w.end()) {
    if (err->empty()) *err = "weight tensor missing: " + k;
    return nullptr;
  }
  return &it->second;
}

// Output-channel order of a conv op: the sub-pixel shuffle (proposed.py:227-251) is folded into it, so that
// packed channel r * gc + c of position f IS out[2 f + r, c].
std::vector<int> channel_perm(const fz::OpD& d) {
  std::vector<int> p(d.N);
  for (int n = 0; n < d.N; ++n) p[n] = n;
  if (d.kind == fz::K_DL && d.N == 64)
    for (int r = 0; r < 2; ++r)
      for (int c = 0; c < 32; ++c) p[r * 32 + c] = 2 * c + r;
  if (d.kind == fz::K_DL && d.N == 128)
    for (int r = 0; r < 2; ++r)
      for (int c2 = 0; c2 < 64; ++c2) p[r * 64 + c2] = r * 64 + (c2 % 32) * 2 + c2 / 32;
  return p;
}

}  // namespace

int fused_blob_floats() { return fz::kBlobFloats; }
int fused_num_ops() { return fz::kNumOps; }
const char* fused_op_name(int i) { return (i >= 0 && i < fz::kNumOps) ? fz::kOpNames[i] : "?"; }
double fused_op_flops(int i) { return (i >= 0 && i < fz::kNumOps) ? fz::kOpFlops[i] : 0.0; }
int fused_parity_stride() { return fz::kParityStride; }
int fused_arena_floats() { return fz::kArenaFloats; }
int fused_num_states() { return fz::kNumStateOffs; }
const char* fused_state_name(int i) { return fz::kStateOffs[i].name; }
int fused_state_off(int i) { return fz::kStateOffs[i].off; }
int fused_num_scratch() { return static_cast<int>(sizeof(fz::kScratchOffs) / sizeof(fz::kScratchOffs[0])); }
const char* fused_scratch_name(int i) { return fz::kScratchOffs[i].name; }
int fused_scratch_off(int i) { return fz::kScratchOffs[i].off; }

// Weight blob of the fused kernel, in plan order.  Conv kernels stay int8 (the container's payload, what the reference's
// .tflite stores; `w = q * scale[out channel]`, converter_proposed.py:901) and the kernel applies the scale in its epilogue.
//   conv fragments: for wave task for fragment f of the task (4 fragments = 16 bytes per lane) for lane for q
//       fragment f -> K segment (time tap t, frequency tap kw), K group g, channel tile T  (same walk as fused_step.hip)
//       32x32x2 tiles: n' = 32 T + (lane & 31), c = 8 g + 4 (lane >> 5) + q
//       16x16x4 tiles: n' = 16 T + (lane & 15), c = 16 g + 4 (lane >> 4) + q
//       byte = Q[perm[n']][t][kw][c]                        (OHWI weights, converter_proposed.py Conv2D kernels)
bool fused_pack_blob(const WeightMap& wm, std::vector<float>* out, std::string* err) {
  out->assign(static_cast<size_t>(fz::kBlobFloats), 0.f);
  err->clear();
  for (int it = 0; it < fz::kNumBlobItems; ++it) {
    const fz::BlobItem& bi = fz::kBlobItems[it];
    const fz::OpD& d = fz::kOps[bi.op];
    float* dst = out->data() + bi.off;
    const std::string key = bi.key;
    if (bi.what == 0) {
      const HostTensor* w = get(wm, key + ".w", err);
      if (!w) return false;
      if (w->dims.size() != 4 || w->dims[0] != d.N || w->dims[3] != d.cin) { *err = "unexpected weight shape for " + key; return false; }
      if (w->q.size() != w->data.size() || (w->scales.size() != 1 && static_cast<int>(w->scales.size()) != d.N)) {
        *err = "fused mode keeps conv weights int8 on the device; " + key + ".w is not an int8 tensor of the container";
        return false;
      }
      const int th = w->dims[1], kw = w->dims[2];
      const std::vector<int> perm = channel_perm(d);
      if (d.path == fz::P_X4) {
        // 4x4x1 tiles: wave `wv` = (64-channel tile ct, K slices kw * VH + hv); lane = (block b, row i) holds channel
        // 64 ct + 4 (b mod 16/VH) + i; fragment f = 4 consecutive input channels of one K segment
        const int VH = d.N < 64 ? 2 : 1, KSc = d.KSg, cps = d.cin / KSc, fps = cps / 4, segw = d.nseg / d.KSt;
        const int nf = segw * fps, nsf = (nf + 3) / 4;
        if (8 * nsf * 256 != bi.floats) { *err = "fragment count mismatch for " + key; return false; }
        int8_t* dst8 = reinterpret_cast<int8_t*>(dst);
        for (int wv = 0; wv < 8; ++wv) {
          const int ct = wv % d.CG, kwv = wv / d.CG;
          for (int f = 0; f < nf; ++f)
            for (int lane = 0; lane < 64; ++lane) {
              const int b = lane >> 2, i = lane & 3, hv = VH == 2 ? (b >> 3) : 0;
              const int v = kwv * VH + hv, ks_t = v / KSc, ks_c = v % KSc;
              const int sg = ks_t * segw + f / fps, c0 = ks_c * cps + 4 * (f % fps);
              const int np = 64 * ct + 4 * (VH == 2 ? (b & 7) : b) + i;
              const int t = fz::kSegTk[bi.op][sg] >> 2, k = fz::kSegTk[bi.op][sg] & 3;
              if (sg >= d.nseg || t >= th || k >= kw || np >= d.N) { *err = "segment / channel outside the kernel of " + key; return false; }
              for (int q = 0; q < 4; ++q)
                dst8[((static_cast<size_t>(wv) * nsf + f / 4) * 64 + lane) * 16 + (f % 4) * 4 + q] =
                    w->q[((static_cast<size_t>(perm[np]) * th + t) * kw + k) * d.cin + c0 + q];
            }
        }
        continue;
      }
      const bool r32 = d.path == fz::P_R32;
      const int G = d.cin / (r32 ? 8 : 16), GW = G / d.KSg;
      const int segw = d.kind == fz::K_UP ? 3 : d.nseg / d.KSt;
      const int nf = segw * GW * d.NT, nsf = (nf + 3) / 4, wtasks = d.CG * d.KSt * d.KSg;
      if (wtasks * nsf * 256 != bi.floats) { *err = "fragment count mismatch for " + key; return false; }
      int8_t* dst8 = reinterpret_cast<int8_t*>(dst);
      for (int task = 0; task < wtasks; ++task) {
        const int ct = task % d.CG, ks = task / d.CG, ks_g = ks % d.KSg, ks_t = ks / d.KSg;
        for (int f = 0; f < nf; ++f) {
          int sg, g, T;
          if (r32) {
            const int nt = f % d.NT, sgi = f / d.NT;
            sg = sgi / G; g = sgi % G; T = ct * d.NT + nt;
          } else {
            sg = ks_t * segw + f / GW; g = ks_g * GW + f % GW; T = ct;
          }
          const int t = fz::kSegTk[bi.op][sg] >> 2, k = fz::kSegTk[bi.op][sg] & 3;
          if (sg >= d.nseg || t >= th || k >= kw) { *err = "segment outside the kernel of " + key; return false; }
          for (int lane = 0; lane < 64; ++lane)
            for (int q = 0; q < 4; ++q) {
              const int np = r32 ? 32 * T + (lane & 31) : 16 * T + (lane & 15);
              const int c = r32 ? 8 * g + 4 * (lane >> 5) + q : 16 * g + 4 * (lane >> 4) + q;
              dst8[((static_cast<size_t>(task) * nsf + f / 4) * 64 + lane) * 16 + (f % 4) * 4 + q] =
                  w->q[((static_cast<size_t>(perm[np]) * th + t) * kw + k) * d.cin + c];
            }
        }
      }
    } else if (bi.what == 1) {
      // bias | per-channel weight scale | gamma | beta | alpha   (packed channel order)
      const HostTensor* b = get(wm, key + ".b", err);
      const HostTensor* w = get(wm, key + ".w", err);
      if (!b || !w) return false;
      if (static_cast<int>(b->size()) != d.N) { *err = "unexpected bias size for " + key; return false; }
      if (w->scales.size() != 1 && static_cast<int>(w->scales.size()) != d.N) { *err = "unexpected scale count for " + key; return false; }
      const std::vector<int> perm = channel_perm(d);
      const int reps = d.kind == fz::K_UP ? 2 : 1;         // the up-sampling layer's parameters apply to even and odd output rows
      const int nt = reps * d.N;
      for (int r = 0; r < reps; ++r)
        for (int n = 0; n < d.N; ++n) {
          dst[r * d.N + n] = b->data[perm[n]];
          dst[nt + r * d.N + n] = w->scales.size() == 1 ? w->scales[0] : w->scales[perm[n]];
        }
      if (d.ln) {
        const HostTensor* g = get(wm, key + ".gamma", err);
        const HostTensor* bt = get(wm, key + ".beta", err);
        const HostTensor* al = get(wm, key + ".alpha", err);
        if (!g || !bt || !al) return false;
        if (static_cast<int>(g->size()) != d.gc || static_cast<int>(bt->size()) != d.gc || al->size() < 1) { *err = "unexpected LayerNorm / PReLU size for " + key; return false; }
        std::memcpy(dst + 2 * nt, g->data.data(), d.gc * sizeof(float));
        std::memcpy(dst + 2 * nt + d.gc, bt->data.data(), d.gc * sizeof(float));
        dst[2 * nt + 2 * d.gc] = al->data[0];
      }
    } else if (bi.what == 2) {
      const std::string ln = key.empty() ? "lstm" : key + "_lstm", dn = key.empty() ? "dense" : key + "_dense";
      const HostTensor* wx = get(wm, ln + ".wx", err);
      const HostTensor* wh = get(wm, ln + ".wh", err);
      const HostTensor* b = get(wm, ln + ".b", err);
      const HostTensor* wd = get(wm, dn + ".w", err);
      const HostTensor* bd = get(wm, dn + ".b", err);
      if (!wx || !wh || !b || !wd || !bd) return false;
      const int din = d.din, dout = d.dout;
      if (wx->dims.size() != 2 || wx->dims[0] != 84 || wx->dims[1] != din || wh->size() != 84u * 21u || b->size() != 84u ||
          wd->dims.size() != 2 || wd->dims[0] != dout || wd->dims[1] != 21 || static_cast<int>(bd->size()) != dout) {
        *err = "unexpected LSTM / Dense shape for " + ln;
        return false;
      }
      // [x ; h ; 3 zero rows] x 84 gate columns, column 4 u + g = gate g (i, f, g, o) of unit u; bias in the same order;
      // Dense as one row of 24 per output: 21 weights, bias, 2 x 0
      auto col = [](int n) { return 4 * (n % 21) + n / 21; };       // Keras column gate * 21 + unit -> interleaved
      for (int k = 0; k < din; ++k)
        for (int n = 0; n < 84; ++n) dst[k * 84 + col(n)] = wx->data[static_cast<size_t>(n) * din + k];
      for (int u = 0; u < 21; ++u)
        for (int n = 0; n < 84; ++n) dst[(din + u) * 84 + col(n)] = wh->data[static_cast<size_t>(n) * 21 + u];
      float* bp = dst + (din + 24) * 84;
      for (int n = 0; n < 84; ++n) bp[col(n)] = b->data[n];
      float* wdr = bp + 84;
      for (int n = 0; n < dout; ++n) {
        for (int u = 0; u < 21; ++u) wdr[n * 24 + u] = wd->data[static_cast<size_t>(n) * 21 + u];
        wdr[n * 24 + 21] = bd->data[n];
      }
    } else if (bi.what == 3) {
      int o = 0;
      for (const char* br : {"_ta", "_fa"}) {
        const HostTensor* w1 = get(wm, key + br + ".w1", err);
        const HostTensor* b1 = get(wm, key + br + ".b1", err);
        const HostTensor* w2 = get(wm, key + br + ".w2", err);
        const HostTensor* b2 = get(wm, key + br + ".b2", err);
        if (!w1 || !b1 || !w2 || !b2) return false;
        if (w1->size() != 16u * 64u || w2->size() != 64u * 16u || b1->size() != 16u || b2->size() != 64u) { *err = "unexpected CTFA shape " + key + br; return false; }
        for (int c = 0; c < 64; ++c)
          for (int u = 0; u < 16; ++u) dst[o + c * 16 + u] = w1->data[static_cast<size_t>(u) * 64 + c];     // w1T [64][16]
        std::memcpy(dst + o + 1024, b1->data.data(), 16 * sizeof(float));
        std::memcpy(dst + o + 1040, w2->data.data(), 1024 * sizeof(float));                                // w2 [64][16] as stored
        std::memcpy(dst + o + 2064, b2->data.data(), 64 * sizeof(float));
        o += 2128;
      }
      const HostTensor* ow = get(wm, "out_conv.w", err);
      const HostTensor* ob = get(wm, "out_conv.b", err);
      if (!ow || !ob) return false;
      if (ow->size() != 64u || ob->size() < 1) { *err = "unexpected output conv shape"; return false; }
      std::memcpy(dst + 4256, ow->data.data(), 64 * sizeof(float));
      dst[4320] = ob->data[0];
    } else {
      const HostTensor* iw = get(wm, "input_layer.w", err);
      const HostTensor* ib = get(wm, "input_layer.b", err);
      const HostTensor* ig = get(wm, "input_layer.gamma", err);
      const HostTensor* ibt = get(wm, "input_layer.beta", err);
      const HostTensor* ia = get(wm, "input_layer.alpha", err);
      if (!iw || !ib || !ig || !ibt || !ia) return false;
      if (iw->size() != 64u || ib->size() != 64u || ig->size() != 64u || ibt->size() != 64u || ia->size() < 1) { *err = "unexpected input layer shape"; return false; }
      std::memcpy(dst, iw->data.data(), 256);
      std::memcpy(dst + 64, ib->data.data(), 256);
      std::memcpy(dst + 128, ig->data.data(), 256);
      std::memcpy(dst + 192, ibt->data.data(), 256);
      dst[256] = ia->data[0];
    }
  }
  return true;
}

}  // namespace nutls
