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

}  // namespace nutls
