// Host engine behind the C ABI of include/nutls.h: weight upload, HBM-resident recurrent state
// (ping-pong: the `cur` tensors of frame t are the `prev` tensors of frame t+1, no copy), the
// per-frame launch plan that wires the kernels of kernels.hip exactly like
// TFL_SIGNITURE.nutls_lstm (/root/reference/dnn_model/converter_proposed.py:188-867), and
// optional hipGraph capture of that plan.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <exception>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/nutls.h"
#include "nutls_internal.hpp"

namespace nutls {

static thread_local std::string g_last_error;

static int fail(int code, const std::string& msg) {
  g_last_error = msg;
  return code;
}
#define HIP_TRY(expr)                                                                        \
  do {                                                                                       \
    hipError_t e__ = (expr);                                                                 \
    if (e__ != hipSuccess)                                                                   \
      return fail(NUTLS_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e__));       \
  } while (0)

// ---------------------------------------------------------------------------------------------
struct StageDesc {
  const char* prefix;
  int depth, f0;
  const char* conv_tag;
  const char* spconv_tag;
  const char* resample;
  int pair;  // decoder: index into kEncoder of the paired encoder stage; encoder: -1
};
// stage order / pairing: converter_proposed.py:221-727 (decoder pairs at :464,500,542,585,627,675)
static const StageDesc kEncoder[6] = {
    {"msfe6_en", 6, 256, "msfe6_ee", "msfe6_ed", "msfe6_down_sampling", -1},
    {"msfe5_en", 5, 128, "msfe5_ee", "msfe5_ed", "msfe5_down_sampling", -1},
    {"msfe4_en", 4, 64, "msfe4_ee", "msfe4_ed", "msfe4_down_sampling", -1},
    {"msfe4_en2", 4, 32, "msfe4_ee2", "msfe4_ed2", "msfe4_down_sampling2", -1},
    {"msfe4_en3", 4, 16, "msfe4_ee3", "msfe4_ed3", "msfe4_down_sampling3", -1},
    {"msfe3_en", 3, 8, "msfe3_ee", "msfe3_ed", "msfe3_down_sampling", -1},
};
static const StageDesc kDecoder[6] = {
    {"msfe3_de", 3, 8, "msfe3_de", "msfe3_dd", "msfe3_upsampling", 5},
    {"msfe4_de", 4, 16, "msfe4_de", "msfe4_dd", "msfe4_upsampling", 4},
    {"msfe4_de2", 4, 32, "msfe4_de2", "msfe4_dd2", "msfe4_upsampling2", 3},
    {"msfe4_de3", 4, 64, "msfe4_de3", "msfe4_dd3", "msfe4_upsampling3", 2},
    {"msfe5_de", 5, 128, "msfe5_de", "msfe5_dd", "msfe5_upsampling", 1},
    {"msfe6_de", 6, 256, "msfe6_de", "msfe6_dd", "msfe6_upsampling", 0},
};
static int decoder_of_encoder(int enc) { return 5 - enc; }

struct StateTensor {
  std::string name_prev, name_cur;
  int d0, d1;        // per-stream dims: (F, C) for conv states, (21, 1) for LSTM states, (d*F, k*G) for ring states
  float* buf[2];     // ping-pong, each [B, d0, d1]; ring / in-place states: buf[0] == buf[1]
  int ring_d = 0;    // > 0: dilated-dense history ring of ring_d frames (physical slot = (step + j) mod d)
  size_t per_stream() const { return static_cast<size_t>(d0) * d1; }
};

struct Launch {
  enum Kind { CONV, LSTM, CTFA, INLAYER, OUTCONV, DDB } kind;
  ConvKind ck;
  ConvParams conv;
  LstmParams lstm;
  CtfaParams ctfa;
  InLayerParams inl;
  OutConvParams outc;
  DdbParams ddb;
  int ddb_index = -1;
  std::string name;
  bool encoder_strided = false;   // one of the 26 encoder (2,3) stride-2 convs (the "encoder conv stack")
};

struct StageStates {
  std::vector<int> conv;    // state index of conv input i (1-based -> [i-1])
  std::vector<int> spconv;  // state index of sub-pixel conv input j
  int h, c;
};

struct ConvLayerW { float *wpk, *bias, *gamma, *beta; float alpha; float *wbf, *wscale; };
struct LstmW { float *wxT, *whT, *bias, *wdT, *bd; int din, dout; };
struct CtfaW { float *w1T, *b1, *w2T, *b2, *w2; };   // w2: [64][16] as stored, w2T: [16][64]

struct Engine {
  int B = 0, device = 0;
  int variant = 0;               // NUTLS_VARIANT_LSTM / NUTLS_VARIANT_BASELINE
  long long steps = 0;           // frames processed (ring position of the baseline's dilated-dense history)
  int* d_step = nullptr;         // the same counter on the device (read by the per-layer / plan-interpreter kernels)
  bool d_step_stale = false;     // fused-mode steps take `steps` by value and leave the device counter behind
  // carried partial sums of the fused kernel's two-tap convs (S = W[tap 0] x, fused_plan.hpp OpD::ys): two blocks in every stream's
  // arena slice; stale after the conv-input states were written from outside the fused kernel -- rebuilt before the next fused step
  float* ysum = nullptr;
  YsOp* d_ys_ops = nullptr;
  float* d_ys_w = nullptr;
  int n_ys_ops = 0;
  bool ys_dirty = false;
  // Lazily written states (fused_plan.hpp OpD::d0_on = 2: the input states of the strided convs, which the fused kernel never reads): a
  // fused step leaves them unwritten and marks them stale; whoever looks at states from outside the kernel -- nutls_state_get / _set /
  // _get_all, nutls_reset, a step of a per-layer mode, the rebuild of the carried sums -- goes through states_materialize first.
  LazyCopy* d_lazy = nullptr;
  int n_lazy = 0;
  bool eager_states = false, states_stale = false;
  std::vector<DdbParams> ddbs;   // baseline: the 13 dilated-dense blocks (host copy, per parity identical)
  DdbParams* d_ddb = nullptr;
  struct DdbStates { int in, blk[6], out; };
  DdbStates ddb_st[13];
  struct DdbW { float *w_in, *b_in, *wg[6], *bg[6], *w1[6], *b1[6], *gamma[6], *beta[6], *w_out, *b_out, *wsmall; float a_in, a_out, alpha[6]; };
  DdbW ddbw[13];
  hipStream_t stream = nullptr;
  std::vector<void*> allocs;
  float* warena = nullptr;       // all weights, one allocation
  size_t wcursor = 0;
  float* arena = nullptr;        // stream-major arena: stream b's tensors at arena + b*sstride + slot offset
  size_t sstride = 0;            // floats per stream
  size_t arena_cursor = 0;       // next free slot offset (floats) while the layout is being built
  std::vector<float**> arena_fixups;   // pointers that hold a slot offset until the arena is allocated
  std::vector<StateTensor> states;   // reserve()d up front: slot_reserve keeps pointers into it
  std::unordered_map<std::string, int> state_index;
  StageStates enc_st[6], dec_st[6];
  int central_h = -1, central_c = -1;
  std::vector<Launch> plan[2];
  float *io_in = nullptr, *io_out = nullptr;
  // STFT front / back end (allocated on first use): previous hop, overlap tail, phasors, windows, twiddles, staging
  float *fe_tail = nullptr, *fe_ola = nullptr, *fe_ph = nullptr, *fe_win = nullptr, *fe_inv = nullptr, *fe_tw = nullptr;
  float *fe_pcm_in = nullptr, *fe_pcm_out = nullptr;
  float *t_inlayer = nullptr, *t_y = nullptr, *t_d = nullptr, *t_up = nullptr;
  float* upcat[6] = {nullptr};
  int offline = 0;       // > 0: offline / block handle for up to this many frames per call (arena slot 0 = carried state)
  int outt = 1;          // utterances of an offline handle (nutls_create_offline_batch): utterance u owns arena slots [u (offline + 1), (u + 1) (offline + 1)):
                         // its carried state, then its frames
  std::vector<Launch> plan_off;   // plan[0] with 'previous frame' = one arena slot earlier
  bool off_bf16 = false;          // block mode: convs on the bf16 matrix pipe where the container holds int8 kernels (NUTLS_OFFLINE_FP32=1: the fp32-MFMA kernels)
  float* zx = nullptr;   // [offline + kScanReadAhead][84] LSTM input products of a block
  int ctfa_causal = 0;   // offline handles: 1 = true 32-frame causal average in the CTFA frequency branch (proposed.py:143-147)
  float* ta_hist = nullptr;   // [12 stages][31 + offline][64] time-attention history (causal mode)
  // block pipeline of an offline handle: the block is cut into chunks of consecutive frames, chunk c runs on its own
  // HIP stream one bottleneck behind chunk c-1 (every layer is causal in time: frame t needs frames <= t only)
  static constexpr int kMaxChunks = 16, kGroups = 16;
  std::vector<hipStream_t> ostream;
  std::vector<hipEvent_t> oev;          // [chunk stream][2 * kGroups]: slot g = the chunk's conv-like launches of group g are enqueued, kGroups + g = its LSTM of group g
  hipEvent_t oev_fork = nullptr;
  int ochunks = 0;                      // 0 = chosen from the block length
  std::vector<int> ogroup;              // launch index of plan_off -> group (a group ends with an LSTM)
  int next_parity = 0;   // parity the next step writes (`cur`); `prev` is read from 1 - next_parity
  int mode = 0;          // 0 plain per-layer launches, 1 per-layer hipGraph replay, 3 fused kernel (statically scheduled; both variants);
                         // (2 was the plan-interpreter kernel of rounds 1-3, retired)
  float* fz_blob = nullptr;              // weight blob of the fused kernel (plan order)
  int fz_streams_req = 0;                // nutls_create_plan: the caller's choice of plan (0: the library's)
  int fz_streams = 1;                    // streams per workgroup of the fused plan this handle runs (packed plans 2 / 4: chosen by the cost model in fused_setup or by nutls_create_plan)
  // CTFA frequency branch of the fused kernel (nutls_internal.hpp FzTa): fz_ta_zero = 64 zeros + a dump row (frame mode); causal32 mode of a
  // streaming handle (nutls_set_ctfa_mode): history ring [B][12][32][64] and the per-step sums [B][12][64]
  float *fz_ta_zero = nullptr, *fz_ta_ring = nullptr, *fz_ta_sum = nullptr;
  float* fz_dbg_buf = nullptr;
  int fz_stop_at = -1;         // >= 0: fused launches run the stop twin and end in front of this op (nutls_profile_production)
  int fz_skew = 0;             // FzTa::skew of the fused launches (start skew of the workgroups; experiment builds of the kernel: see nutls_debug_knob)
  float* fz_dbg = nullptr;     // activation trace [B][kDbgSlots][kDbgSlotFloats] (nutls_debug_trace): steps then run on the profiling build, which fills it
  std::string fz_reason;                 // why there is none (what the packer said), for nutls_set_mode(3)
  unsigned long long* fz_prof = nullptr; // op boundary stamps of workgroup 0 (profiling build)
  int n_cu = 256;
  hipGraphExec_t gexec[2] = {nullptr, nullptr};
  std::unordered_map<std::string, std::pair<float*, size_t>> debug;   // name -> (ptr, floats per stream)
  std::unordered_map<std::string, ConvLayerW> convw;
  std::unordered_map<std::string, LstmW> lstmw;
  std::unordered_map<std::string, CtfaW> ctfaw;
  float *in_w = nullptr, *in_b = nullptr, *in_g = nullptr, *in_bt = nullptr, *out_w = nullptr;
  float in_alpha = 0.f, out_bias = 0.f;

  ~Engine() {
    for (int i = 0; i < 2; ++i)
      if (gexec[i]) (void)hipGraphExecDestroy(gexec[i]);
    for (void* p : allocs) (void)hipFree(p);
    for (hipEvent_t ev : oev) (void)hipEventDestroy(ev);
    if (oev_fork) (void)hipEventDestroy(oev_fork);
    for (hipStream_t st : ostream) (void)hipStreamDestroy(st);
    if (stream) (void)hipStreamDestroy(stream);
  }
};

// ------------------------------------------------------------------------------- helpers ------
static int ilog2(int v) {
  int l = 0;
  while ((1 << l) < v) ++l;
  return l;
}

static int dev_alloc(Engine* e, size_t floats, float** out, bool zero) {
  void* p = nullptr;
  HIP_TRY(hipMalloc(&p, floats * sizeof(float)));
  e->allocs.push_back(p);
  if (zero) HIP_TRY(hipMemset(p, 0, floats * sizeof(float)));
  *out = static_cast<float*>(p);
  return NUTLS_OK;
}

// Reserves `floats` per stream inside the stream-major arena; *out temporarily holds the slot offset
// (as a fake pointer) and is patched to the stream-0 address by arena_commit().
static void slot_reserve(Engine* e, size_t floats, float** out) {
  const size_t off = e->arena_cursor;
  e->arena_cursor += (floats + 63) & ~static_cast<size_t>(63);     // 256-byte aligned slots
  *out = reinterpret_cast<float*>(off * sizeof(float));
  e->arena_fixups.push_back(out);
}

static int arena_commit(Engine* e) {
  // Stream stride = an ODD number of 256-byte lines: at any moment all workgroups touch the same
  // slot offset of their own stream, so a power-of-two-ish stride would line every CU up on the
  // same HBM channel / L2 slice.
  e->sstride = (e->arena_cursor + 63) & ~static_cast<size_t>(63);
  if (const char* sk = getenv("NUTLS_STRIDE_ALIGN")) { const size_t a = static_cast<size_t>(atol(sk)); e->sstride = (e->arena_cursor + a - 1) / a * a; }
  else if (((e->sstride / 64) & 1) == 0) e->sstride += 64;
  void* p = nullptr;
  const size_t bytes = e->sstride * sizeof(float) * e->B;
  HIP_TRY(hipMalloc(&p, bytes));
  e->allocs.push_back(p);
  HIP_TRY(hipMemset(p, 0, bytes));
  e->arena = static_cast<float*>(p);
  for (float** f : e->arena_fixups) *f = e->arena + reinterpret_cast<size_t>(*f) / sizeof(float);
  e->arena_fixups.clear();
  for (StateTensor& st : e->states)
    if (st.ring_d > 0) st.buf[1] = st.buf[0];
  return NUTLS_OK;
}

// per-stream tensor (stream-0 pointer `dev`, `per_stream` floats) <-> dense host array [B][per_stream]
static int copy_stream_tensor(Engine* e, float* dev, size_t per_stream, float* host, bool to_host, int stream_idx = -1) {
  const size_t dpitch = e->sstride * sizeof(float), hpitch = per_stream * sizeof(float);
  const int b0 = stream_idx < 0 ? 0 : stream_idx, nb = stream_idx < 0 ? e->B : 1;
  float* d = dev + static_cast<size_t>(b0) * e->sstride;
  if (to_host) HIP_TRY(hipMemcpy2D(host, hpitch, d, dpitch, hpitch, nb, hipMemcpyDeviceToHost));
  else HIP_TRY(hipMemcpy2D(d, dpitch, host, hpitch, hpitch, nb, hipMemcpyHostToDevice));
  return NUTLS_OK;
}

// All weights live in ONE device allocation (bump-allocated, 256-byte aligned pieces) so the
// persistent kernel can address them with 32-bit offsets from a single base.
static constexpr size_t kWeightArenaFloats = 8u << 20;   // 32 MiB: 11.5 MB of parameters + packing padding

static int upload(Engine* e, const std::vector<float>& v, float** out) {
  if (!e->warena) {
    void* p = nullptr;
    HIP_TRY(hipMalloc(&p, kWeightArenaFloats * sizeof(float)));
    e->allocs.push_back(p);
    HIP_TRY(hipMemset(p, 0, kWeightArenaFloats * sizeof(float)));
    e->warena = static_cast<float*>(p);
  }
  const size_t n = (v.size() + 63) & ~static_cast<size_t>(63);
  if (e->wcursor + n > kWeightArenaFloats) return fail(NUTLS_ERR_WEIGHTS, "weight arena exhausted");
  *out = e->warena + e->wcursor;
  e->wcursor += n;
  HIP_TRY(hipMemcpy(*out, v.data(), v.size() * sizeof(float), hipMemcpyHostToDevice));
  return NUTLS_OK;
}

static const HostTensor* find(const WeightMap& w, const std::string& k, std::string* err) {
  auto it = w.find(k);
  if (it == w.end()) {
    *err = "weight tensor missing: " + k;
    return nullptr;
  }
  return &it->second;
}

static std::vector<float> transpose2d(const HostTensor& t, int rows, int cols) {  // [rows][cols] -> [cols][rows]
  std::vector<float> o(static_cast<size_t>(rows) * cols);
  for (int r = 0; r < rows; ++r)
    for (int c = 0; c < cols; ++c) o[static_cast<size_t>(c) * rows + r] = t.data[static_cast<size_t>(r) * cols + c];
  return o;
}

// Uploads one conv-like layer in MFMA streaming order.
static int prep_conv(Engine* e, const WeightMap& wm, const std::string& layer, const std::string& key, ConvKind kind,
                     const std::vector<int>& perm, const std::vector<std::pair<int, int>>& taps) {
  std::string err;
  const ConvShape sh = conv_shape(kind);
  const HostTensor* w = find(wm, layer + ".w", &err);
  const HostTensor* b = find(wm, layer + ".b", &err);
  if (!w || !b) return fail(NUTLS_ERR_WEIGHTS, err);
  int max_kw = 0;
  for (const auto& tk : taps) max_kw = std::max(max_kw, tk.second);
  int max_perm = -1;
  for (int q : perm) max_perm = std::max(max_perm, q);
  if (w->dims.size() != 4 || w->dims[3] != sh.cin || static_cast<int>(perm.size()) != 32 * sh.nt || w->dims[0] <= max_perm ||
      w->dims[1] < sh.tt || w->dims[2] <= max_kw || static_cast<int>(b->size()) <= max_perm)
    return fail(NUTLS_ERR_WEIGHTS, "unexpected weight / bias shape for " + layer);
  ConvLayerW cw{};
  int rc = upload(e, pack_conv_weights(*w, perm, taps, sh.tt, sh.cin, sh.nt), &cw.wpk);
  if (rc) return rc;
  std::vector<float> bp(perm.size());
  for (size_t i = 0; i < perm.size(); ++i) bp[i] = b->data[perm[i]];
  if ((rc = upload(e, bp, &cw.bias))) return rc;
  if (e->off_bf16) {
    // block mode on an int8 container: the int8 payload as bf16 fragments + the per-channel scale (conv_bf16x3_kernel)
    const std::vector<float> wb = pack_conv_weights_bf16(*w, perm, taps, sh.tt, sh.cin, sh.nt);
    if (!wb.empty() && (w->scales.size() == 1 || static_cast<int>(w->scales.size()) == w->dims[0])) {
      std::vector<float> sc(perm.size(), 0.f);
      for (size_t i = 0; i < perm.size(); ++i)
        if (perm[i] >= 0) sc[i] = w->scales.size() == 1 ? w->scales[0] : w->scales[perm[i]];
      if ((rc = upload(e, wb, &cw.wbf))) return rc;
      if ((rc = upload(e, sc, &cw.wscale))) return rc;
    }
  }
  if (sh.epi_ln) {
    const HostTensor* g = find(wm, layer + ".gamma", &err);
    const HostTensor* bt = find(wm, layer + ".beta", &err);
    const HostTensor* al = find(wm, layer + ".alpha", &err);
    if (!g || !bt || !al) return fail(NUTLS_ERR_WEIGHTS, err);
    if (static_cast<int>(g->size()) != 32 * sh.g || static_cast<int>(bt->size()) != 32 * sh.g || al->size() < 1)
      return fail(NUTLS_ERR_WEIGHTS, "unexpected LayerNorm / PReLU size for " + layer);
    if ((rc = upload(e, g->data, &cw.gamma))) return rc;
    if ((rc = upload(e, bt->data, &cw.beta))) return rc;
    cw.alpha = al->data[0];
  }
  e->convw[key] = cw;
  return NUTLS_OK;
}

static std::vector<int> iota_perm(int n) {
  std::vector<int> p(n);
  for (int i = 0; i < n; ++i) p[i] = i;
  return p;
}
// Sub-pixel shuffle folded into the output-channel order (proposed.py:240-251):
// packed channel r*32+c of position f IS out[2f+r, c].
static std::vector<int> perm_shuffle64() {
  std::vector<int> p(64);
  for (int r = 0; r < 2; ++r)
    for (int c = 0; c < 32; ++c) p[r * 32 + c] = 2 * c + r;
  return p;
}
// 128-channel variant: the reference reshapes with the *input* channel count (32,2), so
// out[2f+r, C2] = y[f, r*64 + (C2%32)*2 + C2/32].
static std::vector<int> perm_shuffle128() {
  std::vector<int> p(128);
  for (int r = 0; r < 2; ++r)
    for (int c2 = 0; c2 < 64; ++c2) p[r * 64 + c2] = r * 64 + (c2 % 32) * 2 + c2 / 32;
  return p;
}

// [O][2][3][I] (OHWI) -> [t][kw][i][o]: the output channel becomes the fastest index
static std::vector<float> ohwi_to_tkio(const HostTensor& w) {
  const int O = w.dims[0], T = w.dims[1], K = w.dims[2], I = w.dims[3];
  std::vector<float> o(w.data.size());
  for (int oc = 0; oc < O; ++oc)
    for (int t = 0; t < T; ++t)
      for (int k = 0; k < K; ++k)
        for (int i = 0; i < I; ++i) o[((static_cast<size_t>(t) * K + k) * I + i) * O + oc] = w.data[((static_cast<size_t>(oc) * T + t) * K + k) * I + i];
  return o;
}

static int prep_ddb_weights(Engine* e, const WeightMap& wm) {
  std::string err;
  int rc;
  for (int b = 0; b < 13; ++b) {
    const bool central = b == 6;
    const std::string tag = central ? "ddb" : std::string((b < 6 ? kEncoder[b] : kDecoder[b - 7]).prefix) + "_ddb";
    Engine::DdbW& W = e->ddbw[b];
    auto conv_prelu = [&](const std::string& n, float** w, float** bias, float* alpha) -> int {
      const HostTensor* tw = find(wm, n + ".w", &err);
      const HostTensor* tb = find(wm, n + ".b", &err);
      const HostTensor* ta = find(wm, n + ".alpha", &err);
      if (!tw || !tb || !ta || tw->dims.size() != 4 || tw->dims[1] != 2 || tw->dims[2] != 3 || static_cast<int>(tb->size()) != tw->dims[0] || ta->size() < 1)
        return fail(NUTLS_ERR_WEIGHTS, err.empty() ? "bad ddb conv " + n : err);
      int r;
      if ((r = upload(e, ohwi_to_tkio(*tw), w))) return r;
      if ((r = upload(e, tb->data, bias))) return r;
      *alpha = ta->data[0];
      return NUTLS_OK;
    };
    if ((rc = conv_prelu(tag + "_in", &W.w_in, &W.b_in, &W.a_in))) return rc;
    if ((rc = conv_prelu(tag + "_out", &W.w_out, &W.b_out, &W.a_out))) return rc;
    // the LDS image of ddb_block_wg (ddb_device.hpp): wg all blocks [G][2][3][k] | w1 all blocks [G out][G in] |
    // bg, b1, gamma, beta per block | b_in | b_out -- a thread's grouped kernel and its 1x1 row are contiguous
    std::vector<float> pk_wg, pk_w1, pk_sm;
    for (int k = 1; k <= 6; ++k) {
      const std::string n = tag + "_" + std::to_string(k);
      const HostTensor* wg = find(wm, n + ".wg", &err);
      const HostTensor* bg = find(wm, n + ".bg", &err);
      const HostTensor* w1 = find(wm, n + ".w1", &err);
      const HostTensor* b1 = find(wm, n + ".b1", &err);
      const HostTensor* gm = find(wm, n + ".gamma", &err);
      const HostTensor* bt = find(wm, n + ".beta", &err);
      const HostTensor* al = find(wm, n + ".alpha", &err);
      if (!wg || !bg || !w1 || !b1 || !gm || !bt || !al) return fail(NUTLS_ERR_WEIGHTS, err);
      if (wg->dims.size() != 4 || wg->dims[3] != k || wg->dims[1] != 2 || wg->dims[2] != 3) return fail(NUTLS_ERR_WEIGHTS, "unexpected grouped-conv shape " + n);
      const int G = wg->dims[0];
      if (w1->size() != static_cast<size_t>(G) * G || static_cast<int>(bg->size()) != G || static_cast<int>(b1->size()) != G ||
          static_cast<int>(gm->size()) != G || static_cast<int>(bt->size()) != G || al->size() < 1)
        return fail(NUTLS_ERR_WEIGHTS, "unexpected 1x1 / LayerNorm size in " + n);
      if ((rc = upload(e, ohwi_to_tkio(*wg), &W.wg[k - 1]))) return rc;
      if ((rc = upload(e, bg->data, &W.bg[k - 1]))) return rc;
      if ((rc = upload(e, transpose2d(*w1, G, G), &W.w1[k - 1]))) return rc;
      if ((rc = upload(e, b1->data, &W.b1[k - 1]))) return rc;
      if ((rc = upload(e, gm->data, &W.gamma[k - 1]))) return rc;
      if ((rc = upload(e, bt->data, &W.beta[k - 1]))) return rc;
      W.alpha[k - 1] = al->data[0];
      pk_wg.insert(pk_wg.end(), wg->data.begin(), wg->data.end());
      pk_w1.insert(pk_w1.end(), w1->data.begin(), w1->data.end());
      for (const HostTensor* t : {bg, b1, gm, bt}) pk_sm.insert(pk_sm.end(), t->data.begin(), t->data.end());
    }
    pk_wg.insert(pk_wg.end(), pk_w1.begin(), pk_w1.end());
    pk_wg.insert(pk_wg.end(), pk_sm.begin(), pk_sm.end());
    for (const char* bn : {"_in.b", "_out.b"}) {
      const HostTensor* tb = find(wm, tag + bn, &err);
      if (!tb) return fail(NUTLS_ERR_WEIGHTS, err);
      pk_wg.insert(pk_wg.end(), tb->data.begin(), tb->data.end());
    }
    // the fused kernel's copy of the 1x1 kernels (ddb_fused.hpp): row g pre-rotated for DPP row rotations,
    // entry n = w1[g][((g - n) mod 16) + 16 h], first the lane's own half h of the row, then (G = 32) the other one
    {
      const int G = static_cast<int>(pk_sm.size()) / 24;
      for (int k = 0; k < 6; ++k)
        for (int g = 0; g < G; ++g)
          for (int half = 0; half < G / 16; ++half)
            for (int n = 0; n < 16; ++n) {
              const int h = half == 0 ? g >> 4 : 1 - (g >> 4);
              pk_wg.push_back(pk_w1[(static_cast<size_t>(k) * G + g) * G + ((g - n) & 15) + 16 * h]);
            }
    }
    if ((rc = upload(e, pk_wg, &W.wsmall))) return rc;
  }
  return NUTLS_OK;
}

static int prep_weights(Engine* e, const WeightMap& wm) {
  std::string err;
  int rc;
  const std::vector<std::pair<int, int>> taps3 = {{0, 0}, {0, 1}, {0, 2}};
  const std::vector<std::pair<int, int>> tap1 = {{0, 0}};
  for (int side = 0; side < 2; ++side)
    for (int s = 0; s < 6; ++s) {
      const StageDesc& st = side ? kDecoder[s] : kEncoder[s];
      const std::string P = st.prefix;
      if ((rc = prep_conv(e, wm, P + "_in", P + "_in", side ? CONV_IN_C128 : CONV_IN_C64, iota_perm(64), tap1))) return rc;
      for (int i = 1; i <= st.depth; ++i) {
        const int cin = (i == 1) ? (side ? 128 : 64) : (side ? 64 : 32);
        const ConvKind k = cin == 32 ? CONV_EL_C32 : cin == 64 ? CONV_EL_C64 : CONV_EL_C128;
        const std::string L = P + "_conv" + std::to_string(i);
        if ((rc = prep_conv(e, wm, L, L, k, iota_perm(32), taps3))) return rc;
      }
      for (int j = 1; j <= st.depth; ++j) {
        const std::string L = P + "_spconv" + std::to_string(j);
        if (j < st.depth) rc = prep_conv(e, wm, L, L, CONV_DL_N64, perm_shuffle64(), taps3);
        else rc = prep_conv(e, wm, L, L, CONV_DL_N128, perm_shuffle128(), taps3);
        if (rc) return rc;
      }
      if (!side) {
        if ((rc = prep_conv(e, wm, st.resample, st.resample, CONV_DOWN, iota_perm(64), taps3))) return rc;
      } else {
        // Conv2DTranspose (1,3) stride 2 (proposed.py:260-265):  out[2i] = W0 x[i] + W2 x[i-1],
        // out[2i+1] = W1 x[i]  
        if ((rc = prep_conv(e, wm, st.resample, std::string(st.resample) + "#even", CONV_UP_EVEN, iota_perm(128), {{0, 2}, {0, 0}}))) return rc;
        if ((rc = prep_conv(e, wm, st.resample, std::string(st.resample) + "#odd", CONV_UP_ODD, iota_perm(128), {{0, 1}}))) return rc;
      }
      // CTFA MLPs
      for (const char* br : {"_ta", "_fa"}) {
        const HostTensor* w1 = find(wm, P + br + ".w1", &err);
        const HostTensor* b1 = find(wm, P + br + ".b1", &err);
        const HostTensor* w2 = find(wm, P + br + ".w2", &err);
        const HostTensor* b2 = find(wm, P + br + ".b2", &err);
        if (!w1 || !b1 || !w2 || !b2) return fail(NUTLS_ERR_WEIGHTS, err);
        if (w1->size() != 16 * 64 || w2->size() != 64 * 16 || b1->size() != 16u || b2->size() != 64u) return fail(NUTLS_ERR_WEIGHTS, "unexpected CTFA shape " + P + br);
        CtfaW cw{};
        if ((rc = upload(e, transpose2d(*w1, 16, 64), &cw.w1T))) return rc;
        if ((rc = upload(e, b1->data, &cw.b1))) return rc;
        if ((rc = upload(e, transpose2d(*w2, 64, 16), &cw.w2T))) return rc;
        if ((rc = upload(e, w2->data, &cw.w2))) return rc;
        if ((rc = upload(e, b2->data, &cw.b2))) return rc;
        e->ctfaw[P + br] = cw;
      }
    }
  if (e->variant == NUTLS_VARIANT_BASELINE) {
    if ((rc = prep_ddb_weights(e, wm))) return rc;
  }
  // LSTM + Dense pairs (13)
  std::vector<std::pair<std::string, std::string>> lstms;
  for (int s = 0; s < 6; ++s) lstms.push_back({std::string(kEncoder[s].prefix) + "_lstm", std::string(kEncoder[s].prefix) + "_dense"});
  lstms.push_back({"lstm", "dense"});
  for (int s = 0; s < 6; ++s) lstms.push_back({std::string(kDecoder[s].prefix) + "_lstm", std::string(kDecoder[s].prefix) + "_dense"});
  if (e->variant != NUTLS_VARIANT_LSTM) lstms.clear();
  for (auto& ld : lstms) {
    const HostTensor* wx = find(wm, ld.first + ".wx", &err);
    const HostTensor* wh = find(wm, ld.first + ".wh", &err);
    const HostTensor* b = find(wm, ld.first + ".b", &err);
    const HostTensor* wd = find(wm, ld.second + ".w", &err);
    const HostTensor* bd = find(wm, ld.second + ".b", &err);
    if (!wx || !wh || !b || !wd || !bd) return fail(NUTLS_ERR_WEIGHTS, err);
    LstmW lw{};
    if (wx->dims.size() != 2 || wh->dims.size() != 2 || wd->dims.size() != 2 || wx->dims[0] != 84 || wh->dims[0] != 84 || wh->dims[1] != 21 ||
        wd->dims[1] != 21 || b->size() != 84u || static_cast<int>(bd->size()) != wd->dims[0])
      return fail(NUTLS_ERR_WEIGHTS, "unexpected LSTM shape " + ld.first);
    lw.din = wx->dims[1];
    lw.dout = wd->dims[0];
    if ((rc = upload(e, transpose2d(*wx, 84, lw.din), &lw.wxT))) return rc;
    if ((rc = upload(e, transpose2d(*wh, 84, 21), &lw.whT))) return rc;
    if ((rc = upload(e, b->data, &lw.bias))) return rc;
    if ((rc = upload(e, transpose2d(*wd, lw.dout, 21), &lw.wdT))) return rc;
    if ((rc = upload(e, bd->data, &lw.bd))) return rc;
    e->lstmw[ld.first] = lw;
  }
  // input layer / output conv
  const HostTensor* iw = find(wm, "input_layer.w", &err);
  const HostTensor* ib = find(wm, "input_layer.b", &err);
  const HostTensor* ig = find(wm, "input_layer.gamma", &err);
  const HostTensor* ibt = find(wm, "input_layer.beta", &err);
  const HostTensor* ia = find(wm, "input_layer.alpha", &err);
  const HostTensor* ow = find(wm, "out_conv.w", &err);
  const HostTensor* ob = find(wm, "out_conv.b", &err);
  if (!iw || !ib || !ig || !ibt || !ia || !ow || !ob) return fail(NUTLS_ERR_WEIGHTS, err);
  if (iw->size() != 64u || ib->size() != 64u || ig->size() != 64u || ibt->size() != 64u || ia->size() < 1 || ow->size() != 64u || ob->size() < 1)
    return fail(NUTLS_ERR_WEIGHTS, "unexpected input layer / output conv shape");
  if ((rc = upload(e, iw->data, &e->in_w))) return rc;
  if ((rc = upload(e, ib->data, &e->in_b))) return rc;
  if ((rc = upload(e, ig->data, &e->in_g))) return rc;
  if ((rc = upload(e, ibt->data, &e->in_bt))) return rc;
  if ((rc = upload(e, ow->data, &e->out_w))) return rc;
  e->in_alpha = ia->data[0];
  e->out_bias = ob->data[0];
  return NUTLS_OK;
}

// ------------------------------------------------------------------------------- state --------
static int add_state(Engine* e, const std::string& prev, const std::string& cur, int d0, int d1, int ring_d = 0) {
  StateTensor st;
  st.ring_d = ring_d;
  st.name_prev = prev;
  st.name_cur = cur;
  st.d0 = d0;
  st.d1 = d1;
  const int idx = static_cast<int>(e->states.size());
  e->states.push_back(st);
  e->states[idx].buf[0] = e->states[idx].buf[1] = nullptr;      // slots: allocate_states()
  e->state_index[prev] = idx;
  e->state_index[cur] = idx;
  return idx;
}

// Arena slots of the state tensors: all first buffers in signature order, then all second buffers in the same
// order -- `cur` and `prev` of EVERY ping-pong tensor are the same constant apart (the fused kernel addresses
// them as parity base + one offset, tools/gen_fused_plan.py mirrors this layout) -- then the in-place rings.
static void allocate_states(Engine* e) {
  for (int b = 0; b < 2; ++b)
    for (StateTensor& st : e->states)
      if (st.ring_d == 0) slot_reserve(e, st.per_stream(), &st.buf[b]);
  for (StateTensor& st : e->states)
    if (st.ring_d > 0) slot_reserve(e, st.per_stream(), &st.buf[0]);      // single buffer; buf[1] aliased after arena_commit
}

// State inventory in the order of the reference's signature (converter_proposed.py:27-186).
static int build_states(Engine* e) {
  for (int side = 0; side < 2; ++side)
    for (int s = 0; s < 6; ++s) {
      const StageDesc& st = side ? kDecoder[s] : kEncoder[s];
      StageStates& ss = side ? e->dec_st[s] : e->enc_st[s];
      for (int i = 1; i <= st.depth; ++i) {
        const int f = st.f0 >> (i - 1);
        const int c = (i == 1) ? (side ? 128 : 64) : (side ? 64 : 32);
        const std::string tag = st.conv_tag;
        int idx = add_state(e, tag + "_prev" + std::to_string(i), tag + "_cur" + std::to_string(i), f, c);
        if (idx < 0) return idx;
        ss.conv.push_back(idx);
      }
      for (int j = 1; j <= st.depth; ++j) {
        const int f = (st.f0 >> st.depth) << (j - 1);
        const std::string tag = st.spconv_tag;
        int idx = add_state(e, tag + "_prev" + std::to_string(j), tag + "_cur" + std::to_string(j), f, 64);
        if (idx < 0) return idx;
        ss.spconv.push_back(idx);
      }
    }
  if (e->variant == NUTLS_VARIANT_BASELINE) {
    // dilated-dense block states (converter_nunet_tls.py:173-180, :228-235): 13 bottlenecks in
    // network order -- 6 encoder stages, the central block ("ddb"), 6 decoder stages
    for (int b = 0; b < 13; ++b) {
      const bool central = b == 6;
      const StageDesc* st = central ? nullptr : (b < 6 ? &kEncoder[b] : &kDecoder[b - 7]);
      const int F = central ? 4 : (st->f0 >> st->depth), C = central ? 64 : 32, G = C / 2;
      const std::string tag = central ? "ddb" : std::string(st->prefix) + "_ddb";
      Engine::DdbStates& ds = e->ddb_st[b];
      if ((ds.in = add_state(e, tag + "_prev_in", tag + "_cur_in", F, C, 1)) < 0) return ds.in;
      for (int k = 1; k <= 6; ++k) {
        const int d = 1 << (k - 1);
        if ((ds.blk[k - 1] = add_state(e, tag + "_prev" + std::to_string(k), tag + "_cur" + std::to_string(k), d * F, k * G, d)) < 0)
          return ds.blk[k - 1];
      }
      if ((ds.out = add_state(e, tag + "_prev_out", tag + "_cur_out", F, G, 1)) < 0) return ds.out;
    }
    return NUTLS_OK;
  }
  auto add_hc = [&](const std::string& base, int* h, int* c) -> int {
    *h = add_state(e, base + "_h", base + "_h", NUTLS_LSTM_UNITS, 1);
    if (*h < 0) return *h;
    *c = add_state(e, base + "_c", base + "_c", NUTLS_LSTM_UNITS, 1);
    return *c < 0 ? *c : 0;
  };
  int rc;
  for (int s = 0; s < 6; ++s)
    if ((rc = add_hc(kEncoder[s].prefix, &e->enc_st[s].h, &e->enc_st[s].c)) < 0) return rc;
  if ((rc = add_hc("state", &e->central_h, &e->central_c)) < 0) return rc;
  for (int s = 0; s < 6; ++s)
    if ((rc = add_hc(kDecoder[s].prefix, &e->dec_st[s].h, &e->dec_st[s].c)) < 0) return rc;
  return NUTLS_OK;
}

// ------------------------------------------------------------------------------- plan ---------
static void push_conv(Engine* e, std::vector<Launch>* plan, const std::string& wkey, ConvKind k, const float* src0,
                      const float* src1, int src_ld, int f_in, int f_out, float* dst0, int ld0, float* dst1, int ld1,
                      int row_mul, int row_add, bool enc_strided = false) {
  const ConvLayerW& w = e->convw.at(wkey);
  Launch L{};
  L.kind = Launch::CONV;
  L.ck = k;
  L.name = wkey;
  L.encoder_strided = enc_strided;
  ConvParams& p = L.conv;
  p.src0 = src0; p.src1 = src1; p.wpk = w.wpk; p.bias = w.bias; p.gamma = w.gamma; p.beta = w.beta;
  p.wbf = w.wbf; p.wscale = w.wscale; p.use_bf16 = 0;
  p.dst0 = dst0; p.dst1 = dst1; p.src_ld = src_ld; p.ld0 = ld0; p.ld1 = ld1;
  p.B = e->B; p.F_in = f_in; p.F_out = f_out; p.log2_fout = ilog2(f_out);
  p.row_mul = row_mul; p.row_add = row_add; p.alpha = w.alpha; p.sstride = static_cast<long long>(e->sstride);
  plan->push_back(L);
}

// Baseline bottleneck b (0..12): same input / output placement as the LSTM + Dense it replaces.
static void push_ddb(Engine* e, std::vector<Launch>* plan, int par, int b, const std::string& name, const float* x, int x_ld, float* dst,
                     int dst_ld, int F, int C) {
  const Engine::DdbW& W = e->ddbw[b];
  const Engine::DdbStates& S = e->ddb_st[b];
  DdbParams p{};
  p.x = x; p.x_ld = x_ld; p.dst = dst; p.dst_ld = dst_ld;
  p.st_in = e->states[S.in].buf[0];
  for (int k = 0; k < 6; ++k) p.st_blk[k] = e->states[S.blk[k]].buf[0];
  p.st_out = e->states[S.out].buf[0];
  p.w_in = W.w_in; p.b_in = W.b_in; p.a_in = W.a_in;
  for (int k = 0; k < 6; ++k) {
    p.wg[k] = W.wg[k]; p.bg[k] = W.bg[k]; p.w1[k] = W.w1[k]; p.b1[k] = W.b1[k];
    p.gamma[k] = W.gamma[k]; p.beta[k] = W.beta[k]; p.alpha[k] = W.alpha[k];
  }
  p.w_out = W.w_out; p.b_out = W.b_out; p.a_out = W.a_out;
  p.wsmall = W.wsmall;
  p.step = e->d_step; p.F = F; p.C = C; p.B = e->B; p.sstride = static_cast<long long>(e->sstride);
  Launch L{};
  L.kind = Launch::DDB;
  L.name = name;
  L.ddb = p;
  L.ddb_index = par * 13 + b;          // x / dst live in parity-specific state tensors: one table entry per parity
  plan->push_back(L);
  if (e->ddbs.size() < 26) e->ddbs.resize(26);
  e->ddbs[L.ddb_index] = p;
}

static void push_lstm(Engine* e, std::vector<Launch>* plan, const std::string& lname, const float* x, int x_ld, int x_rows,
                      int x_cols, float* dst, int dst_ld, int dst_rows, int dst_cols, int h_idx, int c_idx, int par) {
  const LstmW& w = e->lstmw.at(lname);
  Launch L{};
  L.kind = Launch::LSTM;
  L.name = lname;
  LstmParams& p = L.lstm;
  p.x = x; p.x_ld = x_ld; p.x_rows = x_rows; p.x_cols = x_cols;
  p.wxT = w.wxT; p.whT = w.whT; p.bias = w.bias; p.wdT = w.wdT; p.bd = w.bd;
  p.h_in = e->states[h_idx].buf[1 - par]; p.c_in = e->states[c_idx].buf[1 - par];
  p.h_out = e->states[h_idx].buf[par]; p.c_out = e->states[c_idx].buf[par];
  p.dst = dst; p.dst_ld = dst_ld; p.dst_rows = dst_rows; p.dst_cols = dst_cols;
  p.Din = w.din; p.Dout = w.dout; p.B = e->B; p.sstride = static_cast<long long>(e->sstride);
  plan->push_back(L);
}

// One MSFE stage (converter_proposed.py:225-262 encoder, :467-498 decoder).
static void build_stage(Engine* e, std::vector<Launch>* plan, int side, int s, int par, const float* x_src, int x_ld,
                        float* y_dst, int y_ld) {
  const StageDesc& st = side ? kDecoder[s] : kEncoder[s];
  const StageStates& ss = side ? e->dec_st[s] : e->enc_st[s];
  const StageStates* pair_dec = side ? nullptr : &e->dec_st[decoder_of_encoder(s)];
  const std::string P = st.prefix;
  const int D = st.depth, FD = st.f0 >> D;
  auto cur = [&](int idx) { return e->states[idx].buf[par]; };
  auto prev = [&](int idx) { return e->states[idx].buf[1 - par]; };
  const int c1 = side ? 128 : 64;
  // e0 = inconv(x)  -> channels [0,64) of the first strided conv's input
  push_conv(e, plan, P + "_in", side ? CONV_IN_C128 : CONV_IN_C64, x_src, nullptr, x_ld, st.f0, st.f0, cur(ss.conv[0]), c1,
            nullptr, 0, 1, 0);
  // e_i = EL_i([prev_i ; cur_i]); e_i feeds conv i+1 and the stage's own sub-pixel conv D-i+1
  for (int i = 1; i <= D; ++i) {
    const int ci = e->states[ss.conv[i - 1]].d1, fi = st.f0 >> (i - 1);
    const ConvKind k = ci == 32 ? CONV_EL_C32 : ci == 64 ? CONV_EL_C64 : CONV_EL_C128;
    float *d0, *d1 = nullptr;
    int l0, l1 = 0;
    if (i < D) {
      d0 = cur(ss.conv[i]); l0 = e->states[ss.conv[i]].d1;
      d1 = cur(ss.spconv[D - i]) + 32; l1 = 64;          // sub-pixel conv j = D-i+1 -> index D-i
    } else {
      d0 = cur(ss.spconv[0]) + 32; l0 = 64;
    }
    push_conv(e, plan, P + "_conv" + std::to_string(i), k, prev(ss.conv[i - 1]), cur(ss.conv[i - 1]), ci, fi, fi / 2, d0, l0,
              d1, l1, 1, 0, side == 0);
  }
  // d_0 = Dense(LSTM(flatten(e_D)))  -> channels [0,32) of the first sub-pixel conv's input
  if (e->variant == NUTLS_VARIANT_BASELINE)
    push_ddb(e, plan, par, side ? 7 + s : s, P + "_ddb", cur(ss.spconv[0]) + 32, 64, cur(ss.spconv[0]), 64, FD, 32);
  else
    push_lstm(e, plan, P + "_lstm", cur(ss.spconv[0]) + 32, 64, FD, 32, cur(ss.spconv[0]), 64, FD, 32, ss.h, ss.c, par);
  // d_j = DL_j([prev_j ; cur_j])
  const float* dD = nullptr;
  int dD_ld = 0;
  for (int j = 1; j <= D; ++j) {
    const int fj = FD << (j - 1);
    float *d0, *d1 = nullptr;
    int l0, l1 = 0;
    if (j < D) {
      d0 = cur(ss.spconv[j]); l0 = 64;
      if (pair_dec) { d1 = cur(pair_dec->conv[D - j]) + 32; l1 = 64; }   // decoder conv i = D-j+1 (second-level skip)
      push_conv(e, plan, P + "_spconv" + std::to_string(j), CONV_DL_N64, prev(ss.spconv[j - 1]), cur(ss.spconv[j - 1]), 64, fj,
                fj, d0, l0, d1, l1, 2, 0);
    } else {
      if (pair_dec) { d0 = cur(pair_dec->conv[0]) + 64; l0 = 128; }      // skip into decoder conv 1
      else { d0 = e->t_d; l0 = 64; }
      dD = d0; dD_ld = l0;
      push_conv(e, plan, P + "_spconv" + std::to_string(j), CONV_DL_N128, prev(ss.spconv[j - 1]), cur(ss.spconv[j - 1]), 64, fj,
                fj, d0, l0, nullptr, 0, 2, 0);
    }
  }
  // y = d_D * (TA*FA) + e0
  Launch L{};
  L.kind = Launch::CTFA;
  L.name = P + "_ctfa";
  CtfaParams& c = L.ctfa;
  const CtfaW& ta = e->ctfaw.at(P + "_ta");
  const CtfaW& fa = e->ctfaw.at(P + "_fa");
  c.x = dD; c.x_ld = dD_ld; c.e0 = cur(ss.conv[0]); c.e0_ld = c1; c.y = y_dst; c.y_ld = y_ld;
  c.ta_w1T = ta.w1T; c.ta_b1 = ta.b1; c.ta_w2T = ta.w2T; c.ta_b2 = ta.b2;
  c.fa_w1T = fa.w1T; c.fa_b1 = fa.b1; c.fa_w2T = fa.w2T; c.fa_b2 = fa.b2;
  c.ta_w2 = ta.w2; c.fa_w2 = fa.w2;
  c.B = e->B; c.F = st.f0; c.sstride = static_cast<long long>(e->sstride);
  plan->push_back(L);
}

static void build_plan(Engine* e, int par) {
  std::vector<Launch>* plan = &e->plan[par];
  plan->clear();
  {
    Launch L{};
    L.kind = Launch::INLAYER;
    L.name = "input_layer";
    L.inl = InLayerParams{e->io_in, e->t_inlayer, e->in_w, e->in_b, e->in_g, e->in_bt, e->in_alpha, e->B * NUTLS_BINS,
                          static_cast<long long>(e->sstride)};
    plan->push_back(L);
  }
  const float* x = e->t_inlayer;
  int x_ld = 64;
  for (int s = 0; s < 6; ++s) {
    const StageDesc& st = kEncoder[s];
    build_stage(e, plan, 0, s, par, x, x_ld, e->t_y, 64);
    // down-sampling output lives in channels [64,128) of the paired decoder's up-sampling input
    float* cat = e->upcat[decoder_of_encoder(s)];
    push_conv(e, plan, st.resample, CONV_DOWN, e->t_y, nullptr, 64, st.f0, st.f0 / 2, cat + 64, 128, nullptr, 0, 1, 0);
    x = cat + 64;
    x_ld = 128;
  }
  // central LSTM over flatten([4,64]) (converter_proposed.py:456-459)
  if (e->variant == NUTLS_VARIANT_BASELINE)
    push_ddb(e, plan, par, 6, "ddb", e->upcat[0] + 64, 128, e->upcat[0], 128, 4, 64);
  else
    push_lstm(e, plan, "lstm", e->upcat[0] + 64, 128, 4, 64, e->upcat[0], 128, 4, 64, e->central_h, e->central_c, par);
  for (int s = 0; s < 6; ++s) {
    const StageDesc& st = kDecoder[s];
    const int fin = st.f0 / 2;
    push_conv(e, plan, std::string(st.resample) + "#even", CONV_UP_EVEN, e->upcat[s], nullptr, 128, fin, fin, e->t_up, 128, nullptr,
              0, 2, 0);
    push_conv(e, plan, std::string(st.resample) + "#odd", CONV_UP_ODD, e->upcat[s], nullptr, 128, fin, fin, e->t_up, 128, nullptr, 0,
              2, 1);
    float* y = (s < 5) ? e->upcat[s + 1] : e->t_y;
    build_stage(e, plan, 1, s, par, e->t_up, 128, y, (s < 5) ? 128 : 64);
  }
  Launch L{};
  L.kind = Launch::OUTCONV;
  L.name = "out_conv";
  L.outc = OutConvParams{e->t_y, 64, e->io_out, e->out_w, e->out_bias, e->B * NUTLS_BINS, static_cast<long long>(e->sstride)};
  plan->push_back(L);
}

static hipError_t run_launch(const Launch& L, hipStream_t s) {
  switch (L.kind) {
    case Launch::CONV: return launch_conv(L.ck, L.conv, s);
    case Launch::LSTM: return launch_lstm(L.lstm, s);
    case Launch::CTFA: return launch_ctfa(L.ctfa, s);
    case Launch::INLAYER: return launch_input_layer(L.inl, s);
    case Launch::OUTCONV: return launch_out_conv(L.outc, s);
    case Launch::DDB: return launch_ddb(L.ddb, s);
  }
  return hipErrorInvalidValue;
}

static int run_plan(Engine* e, int par, hipStream_t s) {
  for (const Launch& L : e->plan[par]) {
    hipError_t err = run_launch(L, s);
    if (err != hipSuccess) return fail(NUTLS_ERR_HIP, "launch " + L.name + ": " + hipGetErrorString(err));
  }
  if (e->variant == NUTLS_VARIANT_BASELINE) HIP_TRY(launch_incr_step(e->d_step, s));   // ring position of the dilated-dense history
  return NUTLS_OK;
}

// Regions of HBM a launch writes, for the hand-off analysis: (row-0 pointer incl. channel offset, ld, channels)
struct WriteRegion { const float* ptr; int ld, nchan, launch; };

static void collect_writes(const Launch& L, int idx, std::vector<WriteRegion>* out) {
  switch (L.kind) {
    case Launch::CONV: {
      const ConvShape sh = conv_shape(L.ck);
      const int gc = 32 * sh.g;
      out->push_back({L.conv.dst0, L.conv.ld0, gc, idx});
      if (L.conv.dst1) out->push_back({L.conv.dst1, L.conv.ld1, gc, idx});
      break;
    }
    case Launch::LSTM: out->push_back({L.lstm.dst, L.lstm.dst_ld, L.lstm.dst_cols, idx}); break;
    case Launch::CTFA: out->push_back({L.ctfa.y, L.ctfa.y_ld, 64, idx}); break;
    case Launch::INLAYER: out->push_back({L.inl.y, 64, 64, idx}); break;
    case Launch::OUTCONV: break;
    case Launch::DDB: out->push_back({L.ddb.dst, L.ddb.dst_ld, L.ddb.C, idx}); break;
  }
}

// Decides, for every pair of consecutive conv layers (L, N), whether L completes N's LDS image
// (hand-off) -- possible when N keeps all its phases resident and every channel of N's current-frame
// input was written either by L itself (then L forwards those rows from its epilogue) or by a launch
// before L (then L prefetches them from HBM while its own MFMAs run).
// The baseline variant's 13 dilated-dense blocks, as the device-side table every kernel family reads (host copy: Engine::ddbs).
static int upload_ddb_table(Engine* e) {
  if (!e->ddbs.empty()) {
    void* t = nullptr;
    HIP_TRY(hipMalloc(&t, e->ddbs.size() * sizeof(DdbParams)));
    e->allocs.push_back(t);
    HIP_TRY(hipMemcpy(t, e->ddbs.data(), e->ddbs.size() * sizeof(DdbParams), hipMemcpyHostToDevice));
    e->d_ddb = static_cast<DdbParams*>(t);
  }
  return NUTLS_OK;
}

// ---- fused kernel (mode 3): weight blob in plan order + the check that the arena is laid out as the plan says ----
static int fused_setup(Engine* e, const WeightMap& wm) {
  const int v = e->variant;
  const bool same_count = static_cast<int>(e->states.size()) == fused_num_states(v);
  bool ok = same_count && e->sstride >= static_cast<size_t>(fused_arena_floats(v));
  for (int i = 0; ok && i < fused_num_states(v); ++i) {
    const StateTensor& st = e->states[i];
    const bool ring = i >= fused_num_pingpong(v);        // baseline: the dilated-dense history rings are updated in place
    ok = st.name_prev == fused_state_name(v, i) && st.buf[0] - e->arena == fused_state_off(v, i) &&
         st.buf[1] - st.buf[0] == (ring ? 0 : fused_parity_stride(v));
  }
  const float* scratch[11] = {e->t_inlayer, e->t_y, e->t_d, e->t_up, e->upcat[0], e->upcat[1], e->upcat[2], e->upcat[3], e->upcat[4], e->upcat[5], e->ysum};
  ok = ok && fused_num_scratch(v) == 11 && e->ysum && e->ysum - e->arena == fused_ys_off(v);
  for (int i = 0; ok && i < fused_num_scratch(v) && i < 11; ++i) ok = scratch[i] - e->arena == fused_scratch_off(v, i);
  if (ok && v == NUTLS_VARIANT_BASELINE) ok = e->d_ddb != nullptr && e->ddbs.size() == 26;
  if (!ok) return fail(NUTLS_ERR_ARG, "fused plan (tools/gen_fused_plan.py) does not match the engine's arena layout");
  // Which plan: one stream per workgroup, or a packed plan (two / four streams per workgroup: one weight fetch / conversion and one latency
  // chain for all of them in the layers whose images fit LDS that often).  NUTLS_FUSED_STREAMS=1 / 2 / 4 overrides (the stream count must be
  // a multiple).
  // Choice by a two-number cost model: a step takes ceil(workgroups / CUs) rounds of the plan's step time, and those are 1 : 1.65 : 3.44 for
  // 1 / 2 / 4 streams per workgroup (0.284 / 0.467 / 0.976 ms, round 6: profiles/plan_cost_model.json, tools/gpu_plan_cost.py -- the cache policy
  // of round 6 took 15 % off the one- and two-stream kernels and nothing off the four-stream one; round 4 measured 1 : 1.58 : 2.82).  256 streams:
  // one per workgroup (one round); 300 .. 512: two (one round instead of two); 768: one (three rounds of 1.0 < one round of fours at 3.44);
  // 1024, 1536, 2048: two (2 / 3 / 4 rounds of 1.65 < 1 / 2 / 2 rounds of 3.44 -- measured: 1 048 k against 996 k frames/s at 1024 streams,
  // 1 073 k against 1 018 k at 2048, tools/exp/plan_ab.py).  The four-stream plan is still built and selectable (nutls_create_plan).
  int streams = 1;
  {
    const double t_plan[5] = {0.0, 1.0, 1.65, 0.0, 3.44};
    double best = 0.0;
    for (int g : {1, 2, 4}) {
      if (e->B % g != 0 || !fused_has_plan(v, g)) continue;
      const int wgs = e->B / g, rounds = (wgs + e->n_cu - 1) / e->n_cu;
      const double t = rounds * t_plan[g];
      if (best == 0.0 || t < best * 0.98) { best = t; streams = g; }      // (ties and near-ties: the smaller group)
    }
  }
  if (const char* ev = getenv("NUTLS_FUSED_STREAMS")) streams = atoi(ev);      // (developer override; falls back like the library's own choice)
  if (streams < 1 || !fused_has_plan(v, streams) || e->B % streams != 0) streams = 1;
  if (e->fz_streams_req > 0) {          // nutls_create_plan: the caller's choice wins -- or the call fails, it never silently becomes another plan
    if (!fused_has_plan(v, e->fz_streams_req))
      return fail(NUTLS_ERR_ARG, "nutls_create_plan: no fused plan with that many streams per workgroup for this variant (plans: 1, 2, 4 for the LSTM variant, 1 for the baseline)");
    if (e->B % e->fz_streams_req != 0)
      return fail(NUTLS_ERR_ARG, "nutls_create_plan: the batch must be a multiple of streams_per_workgroup");
    streams = e->fz_streams_req;
  }
  if (streams > 1 && (fused_plan_arena_floats(v, streams) != fused_arena_floats(v) || fused_plan_parity_stride(v, streams) != fused_parity_stride(v) ||
                      fused_plan_ys_off(v, streams) != fused_ys_off(v) || fused_plan_ys_block(v, streams) != fused_ys_block(v)))
    return fail(NUTLS_ERR_ARG, "packed fused plan does not share the arena layout of the one-stream plan");
  e->fz_streams = streams;
  std::vector<float> blob;
  std::string err;
  if (fused_pack_blob(v, wm, &blob, &err, streams) != FZ_PACK_OK) {
    // Not packable for the fused kernel -- float conv kernels (no int8 payload), only some of them int8, a scale count that
    // does not match ... -- is not an error of the handle: the per-layer modes only need the de-quantised floats, the handle
    // runs on them (hipGraph replay, mode 1, chosen at the end of nutls_create), and nutls_set_mode(3) reports the reason kept here.
    e->fz_reason = err;
    // ... unless the caller asked for a plan of the fused kernel by name (nutls_create_plan): that request never silently becomes another
    // plan or another kernel family
    if (e->fz_streams_req > 0)
      return fail(NUTLS_ERR_WEIGHTS, "nutls_create_plan: the container cannot run on the fused kernel (" + err + "); nutls_create picks the per-layer kernels for it");
    return NUTLS_OK;
  }
  void* p = nullptr;
  HIP_TRY(hipMalloc(&p, blob.size() * sizeof(float)));
  e->allocs.push_back(p);
  HIP_TRY(hipMemcpy(p, blob.data(), blob.size() * sizeof(float), hipMemcpyHostToDevice));
  e->fz_blob = static_cast<float*>(p);
  int rcz = dev_alloc(e, 128, &e->fz_ta_zero, true);
  if (rcz) return rcz;
  void* q = nullptr;
  // op starts + 8 phase stamps per op (wave 0) + the per-wave trace of the FZ_WTRACE build: 8 waves x ops x 12 shader-clock stamps
  const size_t n_stamps = static_cast<size_t>(fused_plan_num_ops(v, streams)) * (9 + 8 * 12) + 1;
  HIP_TRY(hipMalloc(&q, n_stamps * sizeof(unsigned long long)));
  e->allocs.push_back(q);
  HIP_TRY(hipMemset(q, 0, n_stamps * sizeof(unsigned long long)));
  e->fz_prof = static_cast<unsigned long long*>(q);
  HIP_TRY(v == NUTLS_VARIANT_BASELINE ? fused_base_step_set_attributes()
                                      : (streams == 4 ? fused_step_g4_set_attributes() : (streams == 2 ? fused_step_g2_set_attributes() : fused_step_set_attributes())));
  // the table for rebuilding the carried partial sums (ysum_refresh)
  std::vector<YsOp> yops;
  std::vector<float> yw;
  if (!fused_ys_table(v, wm, &yops, &yw, &err, streams)) return fail(NUTLS_ERR_WEIGHTS, "fused plan: " + err);
  void* yo = nullptr;
  HIP_TRY(hipMalloc(&yo, yops.size() * sizeof(YsOp)));
  e->allocs.push_back(yo);
  HIP_TRY(hipMemcpy(yo, yops.data(), yops.size() * sizeof(YsOp), hipMemcpyHostToDevice));
  e->d_ys_ops = static_cast<YsOp*>(yo);
  e->n_ys_ops = static_cast<int>(yops.size());
  int rc = upload(e, yw, &e->d_ys_w);
  if (rc) return rc;
  // the lazily written states of the one-stream plans (the packed plans hand rows over through some of them: they write everything)
  std::vector<LazyCopy> lazy;
  if (streams == 1) fused_lazy_table(v, &lazy);
  e->n_lazy = static_cast<int>(lazy.size());
  if (e->n_lazy) {
    void* lz = nullptr;
    HIP_TRY(hipMalloc(&lz, lazy.size() * sizeof(LazyCopy)));
    e->allocs.push_back(lz);
    HIP_TRY(hipMemcpy(lz, lazy.data(), lazy.size() * sizeof(LazyCopy), hipMemcpyHostToDevice));
    e->d_lazy = static_cast<LazyCopy*>(lz);
  }
  if (const char* ev = getenv("NUTLS_EAGER_STATES")) e->eager_states = atoi(ev) != 0;      // (developer knob: every launch writes every state)
  if (const char* ev = getenv("NUTLS_FUSED_SKEW")) e->fz_skew = atoi(ev);
  return NUTLS_OK;
}

// The state tensors the last fused step left unwritten, rebuilt from their second copy (the skip-connection slices it did write), in the
// parity that step wrote -- the one every reader outside the kernel looks at.
static int states_materialize(Engine* e, hipStream_t s) {
  if (!e->states_stale || !e->n_lazy) { e->states_stale = false; return NUTLS_OK; }
  if (!s) HIP_TRY(hipDeviceSynchronize());      // (called from a host-side accessor: the step may have run on any stream)
  const int block = (1 - e->next_parity) ? fused_parity_stride(e->variant) : 0;
  HIP_TRY(launch_lazy_states(e->arena, static_cast<long long>(e->sstride), block, e->d_lazy, e->n_lazy, e->B, s));
  e->states_stale = false;
  return NUTLS_OK;
}

// Before a fused step: the partial sums the step reads (the block of the parity it does not write) from the conv-input states it
// would have read as the previous frame -- if anything but the fused kernel wrote those since (ys_dirty).
static int ysum_refresh(Engine* e, int par, hipStream_t s) {
  if (!e->ys_dirty || !e->n_ys_ops) return NUTLS_OK;
  if (int rc = states_materialize(e, s)) return rc;      // (the sums are rebuilt from the conv-input states)
  const int v = e->variant;
  const int x_block = par ? 0 : fused_parity_stride(v);                       // the `prev` parity of this step
  const int ys_block = fused_ys_off(v) + (par ? 0 : fused_ys_block(v));       // the block this step reads
  HIP_TRY(launch_ysum_refresh(e->arena, static_cast<long long>(e->sstride), x_block, ys_block, e->d_ys_ops, e->d_ys_w, e->n_ys_ops, e->B, s));
  e->ys_dirty = false;
  return NUTLS_OK;
}

// (mag_in / mag_out: the caller's device buffers, read by the input layer and written by the last op directly -- no staging copies)
static int run_fused(Engine* e, int par, hipStream_t s, bool prof, const float* mag_in = nullptr, float* mag_out = nullptr) {
  if (!e->fz_blob) return fail(NUTLS_ERR_ARG, "fused mode is not available for this handle");
  const bool base = e->variant == NUTLS_VARIANT_BASELINE;
  if (int rc = ysum_refresh(e, par, s)) return rc;
  auto launch = base ? launch_fused_base_step : (e->fz_streams == 4 ? launch_fused_step_g4 : (e->fz_streams == 2 ? launch_fused_step_g2 : launch_fused_step));
  int skew = e->fz_skew;            // (NUTLS_FUSED_SKEW at creation, nutls_debug_knob(h, "skew", v) later)
  if (e->fz_stop_at >= 0) {         // (nutls_profile_production: one-stream LSTM plan only, checked there)
    launch = launch_fused_step_stop;
    skew = e->fz_stop_at;
  }
  const int eager = (e->eager_states || !e->n_lazy) ? 1 : 0;
  float* const dbg = prof ? e->fz_dbg : nullptr;      // (activation trace: the profiling builds only)
  const long long dbg_ss = static_cast<long long>(kDbgSlots) * kDbgSlotFloats;
  FzTa ta{e->fz_ta_zero, e->fz_ta_zero + 64, 0, 0, 0, 0, skew, eager, dbg, dbg_ss};
  if (e->ctfa_causal && e->fz_ta_ring) {
    const int slot = static_cast<int>(e->steps & 31);          // this frame's row of the history: the sums leave it out, the step overwrites it
    HIP_TRY(launch_ta_sum(e->fz_ta_ring, e->fz_ta_sum, slot, e->B, s));
    ta = FzTa{e->fz_ta_sum, e->fz_ta_ring + slot * 64, 12 * 64, 64, 12 * 32 * 64, 32 * 64, skew, eager, dbg, dbg_ss};
  }
  hipError_t err = launch(e->arena, static_cast<long long>(e->sstride), e->fz_blob, mag_in ? mag_in : e->io_in,
                          mag_out ? mag_out : e->io_out, e->B, par, prof ? e->fz_prof : nullptr,
                          base ? e->d_ddb : nullptr, static_cast<int>(e->steps & 0x3fffffff), e->B / e->fz_streams, s, ta);
  if (err == hipErrorNotSupported && prof)
    return fail(NUTLS_ERR_ARG, "this packed fused plan has no profiling build in the library (NUTLS_BUILD_G4_PROF=1 python -m nunet_amd.build adds the 4-stream one; "
                               "NUTLS_FUSED_STREAMS=1 selects the one-stream plan)");
  if (err != hipSuccess) return fail(NUTLS_ERR_HIP, std::string("fused step launch: ") + hipGetErrorString(err));
  if (base) e->d_step_stale = true;      // ring position of the dilated-dense history went in by value: one launch per step
  e->states_stale = !eager;              // (materialised on demand: states_materialize)
  return NUTLS_OK;
}

// Before a step of any other mode: bring the device-side frame counter up to date if fused-mode steps ran since.
static int sync_step_counter(Engine* e, hipStream_t s) {
  if (e->d_step_stale && e->d_step) {
    HIP_TRY(launch_set_step(e->d_step, static_cast<int>(e->steps & 0x3fffffff), s));
    e->d_step_stale = false;
  }
  return NUTLS_OK;
}

static int capture_graphs(Engine* e) {
  for (int par = 0; par < 2; ++par) {
    if (e->gexec[par]) continue;
    hipGraph_t g = nullptr;
    HIP_TRY(hipStreamBeginCapture(e->stream, hipStreamCaptureModeThreadLocal));
    int rc = run_plan(e, par, e->stream);
    hipError_t ee = hipStreamEndCapture(e->stream, &g);
    if (rc) { if (g) (void)hipGraphDestroy(g); return rc; }      // (a launch failed during capture: the half-built graph is not kept)
    if (ee != hipSuccess) { if (g) (void)hipGraphDestroy(g); return fail(NUTLS_ERR_HIP, std::string("hipStreamEndCapture: ") + hipGetErrorString(ee)); }
    hipError_t ie = hipGraphInstantiate(&e->gexec[par], g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    if (ie != hipSuccess) return fail(NUTLS_ERR_HIP, std::string("hipGraphInstantiate: ") + hipGetErrorString(ie));
  }
  return NUTLS_OK;
}

}  // namespace nutls

// =================================================================================================
//  C ABI
// =================================================================================================
using namespace nutls;

struct nutls_handle {
  Engine eng;
};

// ---- STFT front / back end state ---------------------------------------------------------------
static int frontend_init(Engine* e) {
  if (e->fe_tail) return NUTLS_OK;
  const size_t hop = static_cast<size_t>(e->B) * NUTLS_FRAME_STEP;
  auto dalloc = [&](float** p, size_t n) -> int {
    void* q = nullptr;
    HIP_TRY(hipMalloc(&q, n * sizeof(float)));
    HIP_TRY(hipMemset(q, 0, n * sizeof(float)));
    e->allocs.push_back(q);
    *p = static_cast<float*>(q);
    return NUTLS_OK;
  };
  int rc;
  float* tail = nullptr;
  if ((rc = dalloc(&tail, hop)) || (rc = dalloc(&e->fe_ola, hop)) || (rc = dalloc(&e->fe_ph, static_cast<size_t>(e->B) * (NUTLS_FRAME_STEP + 1) * 2)) ||
      (rc = dalloc(&e->fe_win, NUTLS_FRAME_LEN)) || (rc = dalloc(&e->fe_inv, NUTLS_FRAME_LEN)) || (rc = dalloc(&e->fe_tw, NUTLS_FRAME_LEN)) ||
      (rc = dalloc(&e->fe_pcm_in, hop)) || (rc = dalloc(&e->fe_pcm_out, hop)))
    return rc;
  // windows in float32 like tf.signal.hann_window (interpreter_proposed.py:20-26); twiddles from double
  std::vector<float> hann(NUTLS_FRAME_LEN), win(NUTLS_FRAME_LEN), inv(NUTLS_FRAME_LEN), tw(NUTLS_FRAME_LEN);
  for (int k = 0; k < NUTLS_FRAME_LEN; ++k) {
    const float arg = 6.28318530717958647692f * static_cast<float>(k) / static_cast<float>(NUTLS_FRAME_LEN);
    hann[k] = 0.5f - 0.5f * std::cos(arg);
  }
  win = hann;
  win[0] = 1e-7f; win[NUTLS_FRAME_LEN - 1] = 1e-7f;
  for (int k = 0; k < NUTLS_FRAME_LEN; ++k) {
    const int k2 = (k + NUTLS_FRAME_STEP) % NUTLS_FRAME_LEN;
    inv[k] = hann[k] / (hann[k] * hann[k] + hann[k2] * hann[k2]);
  }
  for (int k = 0; k < NUTLS_FRAME_LEN / 2; ++k) {
    const double a = -2.0 * 3.14159265358979323846 * k / NUTLS_FRAME_LEN;
    tw[2 * k] = static_cast<float>(std::cos(a));
    tw[2 * k + 1] = static_cast<float>(std::sin(a));
  }
  HIP_TRY(hipMemcpy(e->fe_win, win.data(), win.size() * sizeof(float), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(e->fe_inv, inv.data(), inv.size() * sizeof(float), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(e->fe_tw, tw.data(), tw.size() * sizeof(float), hipMemcpyHostToDevice));
  e->fe_tail = tail;
  return NUTLS_OK;
}

extern "C" {

const char* nutls_last_error(void) { return g_last_error.c_str(); }
const char* nutls_version(void) { return "nutls-hip 0.4 (gfx950; fused step: fp32 results on the bf16 matrix pipe)"; }

static int build_offline_plan(Engine* e);

static int create_body(const void* weights, size_t n_bytes, int variant, int batch, int device, int offline_frames, nutls_handle** out, int streams_req);

// (nothing may be thrown through the C ABI: a malformed container or an allocation failure is an error code)
static int create_common(const void* weights, size_t n_bytes, int variant, int batch, int device, int offline_frames, nutls_handle** out,
                         int streams_req = 0) {
  try {
    return create_body(weights, n_bytes, variant, batch, device, offline_frames, out, streams_req);
  } catch (const std::bad_alloc&) {
    return fail(NUTLS_ERR_WEIGHTS, "nutls_create: out of host memory (malformed weight container?)");
  } catch (const std::exception& ex) {
    return fail(NUTLS_ERR_WEIGHTS, std::string("nutls_create: ") + ex.what());
  }
}

static int create_body(const void* weights, size_t n_bytes, int variant, int batch, int device, int offline_frames, nutls_handle** out, int streams_req) {
  if (!weights || !out || batch < 1) return fail(NUTLS_ERR_ARG, "nutls_create: null pointer or batch < 1");
  if (variant != NUTLS_VARIANT_LSTM && variant != NUTLS_VARIANT_BASELINE) return fail(NUTLS_ERR_ARG, "nutls_create: unknown variant");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(NUTLS_ERR_NO_DEVICE, "no HIP device visible: this library has no CPU fallback");
  if (device < 0 || device >= ndev) return fail(NUTLS_ERR_ARG, "nutls_create: device ordinal out of range");
  HIP_TRY(hipSetDevice(device));
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, device));
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(NUTLS_ERR_NO_DEVICE, std::string("device is ") + prop.gcnArchName + ", kernels are built for gfx950 only");
  WeightMap wm;
  std::string err;
  if (!parse_weight_blob(weights, n_bytes, &wm, &err)) return fail(NUTLS_ERR_WEIGHTS, err);
  std::unique_ptr<nutls_handle> h(new nutls_handle());
  Engine* e = &h->eng;
  e->B = batch;
  e->device = device;
  e->variant = variant;
  e->off_bf16 = offline_frames > 0 && getenv("NUTLS_OFFLINE_FP32") == nullptr;      // (developer knob: block mode on the fp32-MFMA kernels)
  HIP_TRY(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking));
  int rc;
  if ((rc = prep_weights(e, wm))) return rc;
  {
    void* ds = nullptr;
    HIP_TRY(hipMalloc(&ds, sizeof(int)));
    e->allocs.push_back(ds);
    HIP_TRY(hipMemset(ds, 0, sizeof(int)));
    e->d_step = static_cast<int*>(ds);
  }
  e->states.reserve(320);   // slot_reserve keeps pointers into this vector: it must never reallocate (130 or 208 states)
  if ((rc = build_states(e))) return rc;
  allocate_states(e);
  const size_t B = static_cast<size_t>(batch);
  if ((rc = dev_alloc(e, B * NUTLS_BINS, &e->io_in, true))) return rc;
  if ((rc = dev_alloc(e, B * NUTLS_BINS, &e->io_out, true))) return rc;
  slot_reserve(e, 256 * 64, &e->t_inlayer);
  slot_reserve(e, 256 * 64, &e->t_y);
  slot_reserve(e, 256 * 64, &e->t_d);
  slot_reserve(e, 256 * 128, &e->t_up);
  for (int s = 0; s < 6; ++s) slot_reserve(e, static_cast<size_t>(kDecoder[s].f0 / 2) * 128, &e->upcat[s]);
  if (offline_frames == 0) slot_reserve(e, static_cast<size_t>(2) * fused_ys_block(variant), &e->ysum);      // (streaming handles: the fused kernel's carried partial sums)
  if ((rc = arena_commit(e))) return rc;
  build_plan(e, 0);
  build_plan(e, 1);
  try {
    rc = upload_ddb_table(e);
  } catch (const std::exception& ex) {     // planning invariants (weights.cpp) are reported, never thrown through the C ABI
    return fail(NUTLS_ERR_ARG, std::string("plan: ") + ex.what());
  }
  if (rc) return rc;
  e->fz_streams_req = streams_req;
  e->n_cu = prop.multiProcessorCount;
  if (offline_frames == 0) {
    try {
      rc = fused_setup(e, wm);
    } catch (const std::exception& ex) {
      return fail(NUTLS_ERR_WEIGHTS, std::string("fused plan weights: ") + ex.what());
    }
    if (rc) return rc;
    if (e->fz_blob) {
      e->mode = 3;          // the default for streaming handles whose container holds int8 conv kernels
    } else {
      // float containers: the per-layer kernels, replayed as a hipGraph -- the same default for C and Python callers
      if ((rc = capture_graphs(e))) return rc;
      e->mode = 1;
    }
  }
  e->debug["input_layer"] = {e->t_inlayer, 256 * 64};
  e->debug["msfe6_de.y"] = {e->t_y, 256 * 64};
  e->debug["msfe6_de.up"] = {e->t_up, 256 * 128};
  e->debug["msfe6_de.d"] = {e->t_d, 256 * 64};
  for (int s = 0; s < 6; ++s) e->debug[std::string(kDecoder[s].prefix) + ".upcat"] = {e->upcat[s], static_cast<size_t>(kDecoder[s].f0 / 2) * 128};
  if (offline_frames > 0) {
    e->offline = offline_frames;
    e->outt = batch / (offline_frames + 1);
    e->mode = 0;
    if ((rc = build_offline_plan(e))) return rc;
  }
  HIP_TRY(hipDeviceSynchronize());
  *out = h.release();
  return NUTLS_OK;
}

int nutls_create(const void* weights, size_t n_bytes, int variant, int batch, int device, nutls_handle** out) {
  return create_common(weights, n_bytes, variant, batch, device, 0, out);
}

int nutls_create_plan(const void* weights, size_t n_bytes, int variant, int batch, int device, int streams_per_workgroup, nutls_handle** out) {
  if (streams_per_workgroup < 0) return fail(NUTLS_ERR_ARG, "nutls_create_plan: streams_per_workgroup must be 0 (library's choice), 1, 2 or 4");
  return create_common(weights, n_bytes, variant, batch, device, 0, out, streams_per_workgroup);
}

int nutls_create_offline(const void* weights, size_t n_bytes, int max_frames, int device, nutls_handle** out) {
  return nutls_create_offline_batch(weights, n_bytes, max_frames, 1, device, out);
}

int nutls_create_offline_batch(const void* weights, size_t n_bytes, int max_frames, int utterances, int device, nutls_handle** out) {
  if (max_frames < 1 || max_frames > 4096) return fail(NUTLS_ERR_ARG, "nutls_create_offline: max_frames must be 1..4096");
  if (utterances < 1 || utterances > 256) return fail(NUTLS_ERR_ARG, "nutls_create_offline_batch: utterances must be 1..256");
  if (static_cast<long long>(utterances) * max_frames > 65536) return fail(NUTLS_ERR_ARG, "nutls_create_offline_batch: utterances x max_frames must be <= 65536");
  // per utterance: one arena slot for the state carried in from the previous block, then max_frames slots for the frames of the block
  return create_common(weights, n_bytes, NUTLS_VARIANT_LSTM, utterances * (max_frames + 1), device, max_frames, out);
}

// ---- offline / block mode -------------------------------------------------------------------------
// plan[0] reads the previous-frame tap of every state tensor from its second ping-pong buffer and writes the
// current frame into the first.  The block plan keeps ONE buffer per tensor: the current frame of block frame
// t is arena slot t+1, its previous frame slot t -- i.e. "cur" pointers move one slot up, "prev" pointers become
// the first buffer at slot 0; the kernels then index slots with the frame number.
static int build_offline_plan(Engine* e) {
  const size_t S = e->sstride;
  auto rw = [&](const float* q) -> float* {
    if (!q) return nullptr;
    float* p = const_cast<float*>(q);
    if (p < e->arena || p >= e->arena + S) return p;               // weights, I/O staging
    for (const StateTensor& st : e->states)
      if (st.buf[1] != st.buf[0] && p >= st.buf[1] && p < st.buf[1] + st.per_stream()) return st.buf[0] + (p - st.buf[1]);
    return p + S;
  };
  e->plan_off = e->plan[0];
  for (Launch& L : e->plan_off) {
    switch (L.kind) {
      case Launch::CONV:
        L.conv.src0 = rw(L.conv.src0); L.conv.src1 = rw(L.conv.src1); L.conv.dst0 = rw(L.conv.dst0); L.conv.dst1 = rw(L.conv.dst1);
        break;
      case Launch::LSTM:
        L.lstm.x = rw(L.lstm.x); L.lstm.dst = rw(L.lstm.dst);
        L.lstm.h_in = rw(L.lstm.h_in); L.lstm.c_in = rw(L.lstm.c_in); L.lstm.h_out = rw(L.lstm.h_out); L.lstm.c_out = rw(L.lstm.c_out);
        break;
      case Launch::CTFA: L.ctfa.x = rw(L.ctfa.x); L.ctfa.e0 = rw(L.ctfa.e0); L.ctfa.y = rw(L.ctfa.y); break;
      case Launch::INLAYER: L.inl.y = rw(L.inl.y); break;
      case Launch::OUTCONV: L.outc.x = rw(L.outc.x); break;
      case Launch::DDB: return fail(NUTLS_ERR_ARG, "offline mode: LSTM variant only");
    }
  }
  int g = 0;
  for (const Launch& L : e->plan_off) {
    e->ogroup.push_back(g);
    if (L.kind == Launch::LSTM) ++g;
  }
  if (g + 2 > Engine::kGroups) return fail(NUTLS_ERR_ARG, "offline plan: more bottlenecks than pipeline groups");      // (the last event of a chunk is its join event)
  // (the chunk streams and their events are created when a block first runs with that many chunks: ensure_chunk_streams)
  int rc = dev_alloc(e, (static_cast<size_t>(e->outt) * e->offline + kScanReadAhead) * 84, &e->zx, true);
  if (rc) return rc;
  return dev_alloc(e, static_cast<size_t>(e->outt) * 12 * (31 + e->offline) * 64, &e->ta_hist, true);      // [utterance][stage][31 + frame][64]
}

int nutls_offline_set_pipeline(nutls_handle* h, int chunks) {
  if (!h || !h->eng.offline) return fail(NUTLS_ERR_ARG, "nutls_offline_set_pipeline: not an offline handle");
  if (chunks < 0 || chunks > Engine::kMaxChunks) return fail(NUTLS_ERR_ARG, "nutls_offline_set_pipeline: chunks must be 0 (automatic) .. 16");
  h->eng.ochunks = chunks;
  return NUTLS_OK;
}

// Launches [first, last) of the block plan for frames [t0, t0 + n) on stream s: every per-frame tensor (arena slots,
// magnitudes in / out, LSTM input products, time-attention history) is addressed from frame t0.
static int launch_block_range(Engine* e, size_t first, size_t last, int t0, int n, bool roll_hist, hipStream_t s, int n_block = 0) {
  // (several utterances: the launches run all of them -- dense stream index u * n + t, SlotMap: utterance u's frames start (offline + 1) slots
  //  after utterance u - 1's; the magnitudes [U, n_block, 256] of the block: a chunk's n frames of utterance u sit n_block rows after those of u - 1)
  const int U = e->outt;
  const SlotMap sm_io = U > 1 ? make_slot_map(n, (n_block > 0 ? n_block : n) - n) : make_slot_map(0, 0);
  const SlotMap sm = U > 1 ? make_slot_map(n, e->offline + 1 - n) : make_slot_map(0, 0);
  const long long utt_stride = static_cast<long long>(e->offline + 1) * static_cast<long long>(e->sstride);
  const long long hist_ustride = static_cast<long long>(12) * (31 + e->offline) * 64;
  const float* a0 = e->arena;
  const float* a1 = e->arena + static_cast<size_t>(U) * (static_cast<size_t>(e->offline) + 1) * e->sstride;
  const size_t d = static_cast<size_t>(t0) * e->sstride;
  auto shc = [&](const float*& q) { if (q && q >= a0 && q < a1) q += d; };
  auto sh = [&](float*& q) { if (q && q >= a0 && q < a1) q += d; };
  int n_ctfa = 0;
  for (size_t i = 0; i < first; ++i) n_ctfa += e->plan_off[i].kind == Launch::CTFA;
  for (size_t i = first; i < last; ++i) {
    Launch L = e->plan_off[i];
    hipError_t err = hipSuccess;
    switch (L.kind) {
      case Launch::CONV:
        shc(L.conv.src0); shc(L.conv.src1); sh(L.conv.dst0); sh(L.conv.dst1);
        L.conv.B = U * n;
        L.conv.sm = sm;
        L.conv.use_bf16 = e->off_bf16 && L.conv.wbf != nullptr;
        err = launch_conv(L.ck, L.conv, s);
        break;
      case Launch::LSTM:
        shc(L.lstm.x); sh(L.lstm.dst); shc(L.lstm.h_in); shc(L.lstm.c_in); sh(L.lstm.h_out); sh(L.lstm.c_out);
        L.lstm.B = U * n;
        L.lstm.sm = sm;
        err = launch_lstm_block(L.lstm, e->zx + static_cast<size_t>(U) * t0 * 84, n, s, U, utt_stride);
        break;
      case Launch::CTFA: {
        shc(L.ctfa.x); shc(L.ctfa.e0); sh(L.ctfa.y);
        L.ctfa.B = U * n;
        L.ctfa.sm = sm;
        float* hist = e->ta_hist + static_cast<size_t>(n_ctfa) * (31 + e->offline) * 64 + static_cast<size_t>(t0) * 64;
        if (e->ctfa_causal) err = launch_ctfa_causal(L.ctfa, hist, roll_hist, s, U, hist_ustride);
        else err = launch_ctfa(L.ctfa, s);
        ++n_ctfa;
        break;
      }
      case Launch::INLAYER:
        L.inl.x += static_cast<size_t>(t0) * NUTLS_BINS; sh(L.inl.y);
        L.inl.n_pos = U * n * NUTLS_BINS;
        L.inl.sm = sm;
        L.inl.sm_io = sm_io;
        err = launch_input_layer(L.inl, s);
        break;
      case Launch::OUTCONV:
        shc(L.outc.x); L.outc.y += static_cast<size_t>(t0) * NUTLS_BINS;
        L.outc.n_pos = U * n * NUTLS_BINS;
        L.outc.sm = sm;
        L.outc.sm_io = sm_io;
        err = launch_out_conv(L.outc, s);
        break;
      default: err = hipErrorInvalidValue;
    }
    if (err != hipSuccess) return fail(NUTLS_ERR_HIP, "block launch " + L.name + ": " + hipGetErrorString(err));
  }
  return NUTLS_OK;
}

// Streams + events of the block pipeline for `chunks` chunks, created on first use (a handle that never pipelines owns none).
static int ensure_chunk_streams(Engine* e, int chunks) {
  if (!e->oev_fork) HIP_TRY(hipEventCreateWithFlags(&e->oev_fork, hipEventDisableTiming));
  while (static_cast<int>(e->ostream.size()) < chunks) {
    hipStream_t st = nullptr;
    HIP_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    e->ostream.push_back(st);
    for (int k = 0; k < 2 * Engine::kGroups; ++k) {      // per group: convs done, LSTM done
      hipEvent_t ev = nullptr;
      HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
      e->oev.push_back(ev);
    }
  }
  return NUTLS_OK;
}

int nutls_offline_set_ctfa_mode(nutls_handle* h, int mode) {
  if (!h || !h->eng.offline) return fail(NUTLS_ERR_ARG, "nutls_offline_set_ctfa_mode: not an offline handle");
  if (mode != NUTLS_CTFA_FRAME && mode != NUTLS_CTFA_CAUSAL32) return fail(NUTLS_ERR_ARG, "nutls_offline_set_ctfa_mode: unknown mode");
  Engine* e = &h->eng;
  if (e->ctfa_causal == (mode == NUTLS_CTFA_CAUSAL32)) return NUTLS_OK;      // already in effect: the history stays
  HIP_TRY(hipSetDevice(e->device));
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemset(e->ta_hist, 0, static_cast<size_t>(e->outt) * 12 * (31 + e->offline) * 64 * sizeof(float)));      // a mode switch starts a new history
  e->ctfa_causal = mode == NUTLS_CTFA_CAUSAL32;
  return NUTLS_OK;
}

/* Streaming handles: the same choice for the fused kernel's CTFA (mode 3).  Causal32 keeps, outside the arena, the time attention of the
 * last 32 frames of every stream and stage; the sums over the 31 frames before the current one are formed by a small kernel in front of
 * the step (two launches per frame in this mode), the step itself is the same kernel. */
int nutls_set_ctfa_mode(nutls_handle* h, int mode) {
  if (!h) return fail(NUTLS_ERR_ARG, "nutls_set_ctfa_mode: null handle");
  if (h->eng.offline) return nutls_offline_set_ctfa_mode(h, mode);
  if (mode != NUTLS_CTFA_FRAME && mode != NUTLS_CTFA_CAUSAL32) return fail(NUTLS_ERR_ARG, "nutls_set_ctfa_mode: unknown mode");
  Engine* e = &h->eng;
  if (e->ctfa_causal == (mode == NUTLS_CTFA_CAUSAL32)) return NUTLS_OK;      // already in effect: the history stays
  if (mode == NUTLS_CTFA_CAUSAL32 && (!e->fz_blob || e->mode != 3))
    return fail(NUTLS_ERR_ARG, "nutls_set_ctfa_mode: the causal32 CTFA of a streaming handle runs on the fused kernel (mode 3) only");
  HIP_TRY(hipSetDevice(e->device));
  HIP_TRY(hipDeviceSynchronize());
  const size_t ring = static_cast<size_t>(e->B) * 12 * 32 * 64;
  if (mode == NUTLS_CTFA_CAUSAL32 && !e->fz_ta_ring) {
    int rc = dev_alloc(e, ring, &e->fz_ta_ring, true);
    if (!rc) rc = dev_alloc(e, static_cast<size_t>(e->B) * 12 * 64, &e->fz_ta_sum, true);
    if (rc) return rc;
  }
  if (e->fz_ta_ring) HIP_TRY(hipMemset(e->fz_ta_ring, 0, ring * sizeof(float)));      // a mode switch starts a new history
  e->ctfa_causal = mode == NUTLS_CTFA_CAUSAL32;
  return NUTLS_OK;
}

int nutls_process_block(nutls_handle* h, const float* mag_in, float* mag_out, int n_frames, void* stream) {
  if (!h || !mag_in || !mag_out) return fail(NUTLS_ERR_ARG, "nutls_process_block: null pointer");
  Engine* e = &h->eng;
  if (!e->offline) return fail(NUTLS_ERR_ARG, "nutls_process_block: not an offline handle (nutls_create_offline)");
  if (n_frames < 1 || n_frames > e->offline) return fail(NUTLS_ERR_ARG, "nutls_process_block: n_frames out of range");
  HIP_TRY(hipSetDevice(e->device));
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int U = e->outt;
  const size_t bytes = static_cast<size_t>(U) * n_frames * NUTLS_BINS * sizeof(float);
  if (mag_in != e->io_in) HIP_TRY(hipMemcpyAsync(e->io_in, mag_in, bytes, hipMemcpyDeviceToDevice, s));
  int C = e->ochunks;
  if (C == 0) C = n_frames >= 768 ? 3 : n_frames >= 256 ? 2 : 1;      // (four compute queues are served at a time: chunk 0 rides on the caller's stream, three chunks = three queues)
  C = std::max(1, std::min({C, static_cast<int>(Engine::kMaxChunks), n_frames}));
  // (several utterances: every launch runs all of them -- their 13 scans side by side on their own wavefronts, the small layers U times fuller;
  //  the chunks cut the frames of every utterance alike)
  if (C == 1) {
    int rc = launch_block_range(e, 0, e->plan_off.size(), 0, n_frames, true, s);
    if (rc) return rc;
  } else {
    // chunk c, group g (= the layers up to and including bottleneck g) starts when chunk c-1 has finished group g: then
    // the previous-frame taps of all its layers, the LSTM's h / c and the time-attention history of frame t0-1 exist
    const int per = (n_frames + C - 1) / C;
    const int n_groups = e->ogroup.back() + 1;
    // chunk 0 runs on the caller's stream, chunk c > 0 on chunk stream c-1: a block with C chunks keeps C hardware queues
    // busy, not C + 1 with the caller's queue parked on the join -- the GPU serves four compute queues at a time, and a fifth
    // one that holds a dependency of the others serialises the whole pipeline (4 chunks: 11.7 ms instead of < 4.5)
    if (int rc0 = ensure_chunk_streams(e, C - 1)) return rc0;
    auto cs = [&](int c) { return c == 0 ? s : e->ostream[c - 1]; };
    HIP_TRY(hipEventRecord(e->oev_fork, s));
    for (int c = 1; c < C; ++c) HIP_TRY(hipStreamWaitEvent(cs(c), e->oev_fork, 0));
    int rc = NUTLS_OK;
    size_t first = 0;
    for (int g = 0; g < n_groups && rc == NUTLS_OK; ++g) {
      size_t last = first;
      while (last < e->plan_off.size() && e->ogroup[last] == g) ++last;
      // a group = its conv-like layers, then its LSTM (input products, scan, Dense) if it has one: two dependencies per
      // group -- chunk c's convs start when chunk c-1's convs of the group are done (previous-frame taps, time-attention
      // history), its scan when that chunk's scan is (h / c); the convs do not wait for a scan they do not read
      size_t mid = last;
      for (size_t i = first; i < last; ++i)
        if (e->plan_off[i].kind == Launch::LSTM) { mid = i; break; }
      constexpr int EV = 2 * Engine::kGroups;
      for (int c = 0; c < C && rc == NUTLS_OK; ++c) {
        const int t0 = c * per, n = std::min(per, n_frames - t0);
        if (n <= 0) continue;
        for (int half = 0; half < 2 && rc == NUTLS_OK; ++half) {
          const size_t a = half ? mid : first, b = half ? last : mid;
          if (a == b) continue;
          const int slot = half * Engine::kGroups + g;
          if (c > 0 && hipStreamWaitEvent(cs(c), e->oev[(c - 1) * EV + slot], 0) != hipSuccess) rc = fail(NUTLS_ERR_HIP, "block pipeline: hipStreamWaitEvent");
          if (rc == NUTLS_OK) rc = launch_block_range(e, a, b, t0, n, false, cs(c), n_frames);
          if (rc == NUTLS_OK && c + 1 < C && hipEventRecord(e->oev[c * EV + slot], cs(c)) != hipSuccess) rc = fail(NUTLS_ERR_HIP, "block pipeline: hipEventRecord");
        }
      }
      first = last;
    }
    // join: the caller's stream continues after every chunk stream -- also when a launch failed half way, so that
    // whatever was enqueued is ordered before the caller's next work
    for (int c = 1; c < C; ++c) {
      hipEvent_t done = e->oev[(c - 1) * 2 * Engine::kGroups + Engine::kGroups - 1];      // a spare slot of chunk stream c-1's events (groups end at kGroups - 2)
      if (hipEventRecord(done, cs(c)) == hipSuccess) (void)hipStreamWaitEvent(s, done, 0);
    }
    if (rc) return rc;
    if (e->ctfa_causal)
      for (int k = 0; k < 12; ++k) {
        hipError_t err = launch_ctfa_hist_roll(e->ta_hist + static_cast<size_t>(k) * (31 + e->offline) * 64, n_frames, s, U, static_cast<long long>(12) * (31 + e->offline) * 64);
        if (err != hipSuccess) return fail(NUTLS_ERR_HIP, std::string("time-attention history roll: ") + hipGetErrorString(err));
      }
  }
  if (mag_out != e->io_out) HIP_TRY(hipMemcpyAsync(mag_out, e->io_out, bytes, hipMemcpyDeviceToDevice, s));
  // the last frame's slot of every utterance becomes its carried state of the next block
  {
    const size_t pitch = (static_cast<size_t>(e->offline) + 1) * e->sstride * sizeof(float);
    HIP_TRY(hipMemcpy2DAsync(e->arena, pitch, e->arena + static_cast<size_t>(n_frames) * e->sstride, pitch, e->sstride * sizeof(float), U, hipMemcpyDeviceToDevice, s));
  }
  e->steps += n_frames;
  return NUTLS_OK;
}

int nutls_process_block_host(nutls_handle* h, const float* mag_in, float* mag_out, int n_frames) {
  if (!h || !mag_in || !mag_out) return fail(NUTLS_ERR_ARG, "nutls_process_block_host: null pointer");
  Engine* e = &h->eng;
  if (!e->offline || n_frames < 1 || n_frames > e->offline) return fail(NUTLS_ERR_ARG, "nutls_process_block_host: not an offline handle or n_frames out of range");
  HIP_TRY(hipSetDevice(e->device));
  const size_t bytes = static_cast<size_t>(e->outt) * n_frames * NUTLS_BINS * sizeof(float);
  HIP_TRY(hipMemcpyAsync(e->io_in, mag_in, bytes, hipMemcpyHostToDevice, e->stream));
  int rc = nutls_process_block(h, e->io_in, e->io_out, n_frames, e->stream);
  if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(mag_out, e->io_out, bytes, hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));
  return NUTLS_OK;
}

int nutls_destroy(nutls_handle* h) {
  if (!h) return NUTLS_OK;
  (void)hipSetDevice(h->eng.device);
  (void)hipDeviceSynchronize();
  delete h;
  return NUTLS_OK;
}

int nutls_batch(nutls_handle* h) { return h ? h->eng.B : fail(NUTLS_ERR_ARG, "null handle"); }
int nutls_streams_per_workgroup(nutls_handle* h) { return h ? (h->eng.fz_blob ? h->eng.fz_streams : 1) : fail(NUTLS_ERR_ARG, "null handle"); }
int nutls_launches_per_step(nutls_handle* h) { return h ? static_cast<int>(h->eng.plan[0].size()) : fail(NUTLS_ERR_ARG, "null handle"); }

int nutls_io_buffers(nutls_handle* h, float** mag_in, float** mag_out) {
  if (!h || !mag_in || !mag_out) return fail(NUTLS_ERR_ARG, "nutls_io_buffers: null pointer");
  *mag_in = h->eng.io_in;
  *mag_out = h->eng.io_out;
  return NUTLS_OK;
}

// The causal32 CTFA of a streaming handle lives in the fused kernel only: the per-layer kernels compute the frame-mode attention and
// never write the history ring, so every path that would run them on such a handle refuses instead of mixing the two silently.
static int refuse_per_layer_in_causal32(const Engine* e, const char* who) {
  if (e->ctfa_causal && !e->offline)
    return fail(NUTLS_ERR_ARG, std::string(who) + ": the causal32 CTFA of a streaming handle runs on the fused kernel (mode 3) only -- nutls_set_ctfa_mode(NUTLS_CTFA_FRAME) first");
  return NUTLS_OK;
}

int nutls_use_graph(nutls_handle* h, int enable) {
  if (!h) return fail(NUTLS_ERR_ARG, "null handle");
  Engine* e = &h->eng;
  if (int rc = refuse_per_layer_in_causal32(e, "nutls_use_graph")) return rc;
  HIP_TRY(hipSetDevice(e->device));
  if (enable) {
    int rc = capture_graphs(e);
    if (rc) return rc;
  }
  e->mode = enable ? 1 : 0;
  return NUTLS_OK;
}

int nutls_set_mode(nutls_handle* h, int mode) {
  if (!h || mode < 0 || mode > 3) return fail(NUTLS_ERR_ARG, "nutls_set_mode: mode must be 0, 1 or 3");
  if (mode == 2) return fail(NUTLS_ERR_ARG, "nutls_set_mode: mode 2 (the plan-interpreter kernel of rounds 1-3) was retired: 3 = fused kernel, 1 / 0 = one kernel per layer");
  if (mode == 3 && !h->eng.fz_blob)
    return fail(NUTLS_ERR_ARG, "nutls_set_mode: mode 3 (fused kernel) needs a streaming handle made from a container with int8 conv kernels" +
                                   (h->eng.fz_reason.empty() ? std::string() : " (" + h->eng.fz_reason + ")"));
  if (h->eng.offline && mode != 0) return fail(NUTLS_ERR_ARG, "nutls_set_mode: offline handles run per-layer launches (mode 0)");
  if (mode != 3)
    if (int rc = refuse_per_layer_in_causal32(&h->eng, "nutls_set_mode")) return rc;
  if (mode == 1) return nutls_use_graph(h, 1);
  h->eng.mode = mode;
  return NUTLS_OK;
}

int nutls_step(nutls_handle* h, const float* mag_in, float* mag_out, void* stream) {
  if (!h || !mag_in || !mag_out) return fail(NUTLS_ERR_ARG, "nutls_step: null pointer");
  Engine* e = &h->eng;
  if (e->offline) return fail(NUTLS_ERR_ARG, "nutls_step: offline handle, use nutls_process_block");
  HIP_TRY(hipSetDevice(e->device));
  hipStream_t s = static_cast<hipStream_t>(stream);
  const size_t bytes = static_cast<size_t>(e->B) * NUTLS_BINS * sizeof(float);
  const bool direct = e->mode == 3;      // the fused kernel takes the caller's buffers as they are
  if (!direct && mag_in != e->io_in) HIP_TRY(hipMemcpyAsync(e->io_in, mag_in, bytes, hipMemcpyDeviceToDevice, s));
  const int par = e->next_parity;
  if (e->mode != 3)
    if (int rc = states_materialize(e, s)) return rc;      // (the per-layer kernels read every conv-input state)
  if (e->mode == 3) {
    int rc = run_fused(e, par, s, e->fz_dbg != nullptr, mag_in, mag_out);
    if (rc) return rc;
  } else if (e->mode == 1) {
    e->ys_dirty = true;
    int rc = sync_step_counter(e, s);
    if (rc) return rc;
    HIP_TRY(hipGraphLaunch(e->gexec[par], s));
  } else {
    e->ys_dirty = true;
    int rc = sync_step_counter(e, s);
    if (!rc) rc = run_plan(e, par, s);
    if (rc) return rc;
  }
  if (!direct && mag_out != e->io_out) HIP_TRY(hipMemcpyAsync(mag_out, e->io_out, bytes, hipMemcpyDeviceToDevice, s));
  e->next_parity = 1 - par;
  e->steps += 1;
  return NUTLS_OK;
}

// ---- page-locked host buffers (nutls_host_alloc): base -> bytes ---------------------------------------------------------
static std::mutex g_pin_mu;
static std::map<const char*, size_t> g_pins;

void* nutls_host_alloc(size_t bytes) {
  void* p = nullptr;
  if (bytes == 0 || hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess || !p) {
    (void)hipGetLastError();
    fail(NUTLS_ERR_HIP, "nutls_host_alloc: hipHostMalloc of " + std::to_string(bytes) + " bytes failed");
    return nullptr;
  }
  std::lock_guard<std::mutex> lk(g_pin_mu);
  g_pins[static_cast<const char*>(p)] = bytes;
  return p;
}

void nutls_host_free(void* p) {
  if (!p) return;
  {
    std::lock_guard<std::mutex> lk(g_pin_mu);
    if (!g_pins.erase(static_cast<const char*>(p))) return;      // not ours (or freed twice): leave it alone
  }
  (void)hipHostFree(p);
}

static bool host_pinned(const void* p, size_t bytes) {
  std::lock_guard<std::mutex> lk(g_pin_mu);
  auto it = g_pins.upper_bound(static_cast<const char*>(p));
  if (it == g_pins.begin()) return false;
  --it;
  return static_cast<const char*>(p) + bytes <= it->first + it->second;
}

int nutls_step_host(nutls_handle* h, const float* mag_in, float* mag_out) {
  if (!h || !mag_in || !mag_out) return fail(NUTLS_ERR_ARG, "nutls_step_host: null pointer");
  Engine* e = &h->eng;
  HIP_TRY(hipSetDevice(e->device));
  const size_t bytes = static_cast<size_t>(e->B) * NUTLS_BINS * sizeof(float);
  if (e->mode == 3 && host_pinned(mag_in, bytes) && host_pinned(mag_out, bytes)) {
    // the fused kernel takes the caller's buffers as they are: the frame crosses the link inside the launch, no copy commands
    // (B = 1024: 0.976 ms per call against 1.048 through two DMA copies of the same pinned buffers and 1.10-1.11 from pageable memory)
    int rc = nutls_step(h, mag_in, mag_out, e->stream);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(e->stream));
    return NUTLS_OK;
  }
  HIP_TRY(hipMemcpyAsync(e->io_in, mag_in, bytes, hipMemcpyHostToDevice, e->stream));
  int rc = nutls_step(h, e->io_in, e->io_out, e->stream);
  if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(mag_out, e->io_out, bytes, hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));
  return NUTLS_OK;
}

int nutls_stft_hop(nutls_handle* h, const float* pcm_in, void* stream) {
  if (!h || !pcm_in) return fail(NUTLS_ERR_ARG, "nutls_stft_hop: null pointer");
  Engine* e = &h->eng;
  HIP_TRY(hipSetDevice(e->device));
  int rc = frontend_init(e);
  if (rc) return rc;
  HIP_TRY(launch_stft_hop(pcm_in, e->fe_tail, e->fe_win, e->fe_tw, e->io_in, e->fe_ph, e->B, static_cast<hipStream_t>(stream)));
  return NUTLS_OK;
}

int nutls_istft_hop(nutls_handle* h, float* pcm_out, int dc_mode, void* stream) {
  if (!h || !pcm_out) return fail(NUTLS_ERR_ARG, "nutls_istft_hop: null pointer");
  if (dc_mode != NUTLS_DC_EDGE && dc_mode != NUTLS_DC_ZERO) return fail(NUTLS_ERR_ARG, "dc_mode must be NUTLS_DC_EDGE or NUTLS_DC_ZERO");
  Engine* e = &h->eng;
  HIP_TRY(hipSetDevice(e->device));
  int rc = frontend_init(e);
  if (rc) return rc;
  HIP_TRY(launch_istft_hop(e->io_out, e->fe_ph, e->fe_inv, e->fe_tw, e->fe_ola, pcm_out, dc_mode == NUTLS_DC_EDGE ? 1 : 0, e->B,
                           static_cast<hipStream_t>(stream)));
  return NUTLS_OK;
}

int nutls_enhance_hop(nutls_handle* h, const float* pcm_in, float* pcm_out, int dc_mode, void* stream) {
  if (!h || !pcm_in || !pcm_out) return fail(NUTLS_ERR_ARG, "nutls_enhance_hop: null pointer");
  int rc = nutls_stft_hop(h, pcm_in, stream);
  if (rc) return rc;
  Engine* e = &h->eng;
  if ((rc = nutls_step(h, e->io_in, e->io_out, stream))) return rc;
  return nutls_istft_hop(h, pcm_out, dc_mode, stream);
}

int nutls_enhance_hop_host(nutls_handle* h, const float* pcm_in, float* pcm_out, int dc_mode) {
  if (!h || !pcm_in || !pcm_out) return fail(NUTLS_ERR_ARG, "nutls_enhance_hop_host: null pointer");
  Engine* e = &h->eng;
  HIP_TRY(hipSetDevice(e->device));
  int rc = frontend_init(e);
  if (rc) return rc;
  const size_t bytes = static_cast<size_t>(e->B) * NUTLS_FRAME_STEP * sizeof(float);
  HIP_TRY(hipMemcpyAsync(e->fe_pcm_in, pcm_in, bytes, hipMemcpyHostToDevice, e->stream));
  if ((rc = nutls_enhance_hop(h, e->fe_pcm_in, e->fe_pcm_out, dc_mode, e->stream))) return rc;
  HIP_TRY(hipMemcpyAsync(pcm_out, e->fe_pcm_out, bytes, hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));
  return NUTLS_OK;
}

int nutls_state_count(nutls_handle* h) { return h ? static_cast<int>(h->eng.states.size()) : fail(NUTLS_ERR_ARG, "null handle"); }

int nutls_state_info(nutls_handle* h, int index, const char** name, int* dim0, int* dim1) {
  if (!h || index < 0 || index >= static_cast<int>(h->eng.states.size())) return fail(NUTLS_ERR_ARG, "nutls_state_info: bad index");
  const StateTensor& st = h->eng.states[index];
  if (name) *name = st.name_prev.c_str();
  if (dim0) *dim0 = st.d0;
  if (dim1) *dim1 = st.d1;
  return NUTLS_OK;
}

// Ring states: the device keeps frame j of the reference's [d, F, C] history (0 = oldest) in physical
// slot (steps + j) mod d.  to_logical: physical -> reference order (get); else reference -> physical (set).
static void rotate_ring(Engine* e, const StateTensor& st, float* host, bool to_logical) {
  const int d = st.ring_d;
  const size_t frame = st.per_stream() / d;
  std::vector<float> tmp(st.per_stream());
  for (int b = 0; b < e->B; ++b) {
    float* base = host + static_cast<size_t>(b) * st.per_stream();
    for (int j = 0; j < d; ++j) {
      const int slot = static_cast<int>((e->steps + j) % d);
      const float* src = base + static_cast<size_t>(to_logical ? slot : j) * frame;
      float* dst = tmp.data() + static_cast<size_t>(to_logical ? j : slot) * frame;
      std::memcpy(dst, src, frame * sizeof(float));
    }
    std::memcpy(base, tmp.data(), st.per_stream() * sizeof(float));
  }
}

static int state_lookup(Engine* e, const char* name, size_t n_floats, StateTensor** out) {
  if (!name) return fail(NUTLS_ERR_ARG, "state name is null");
  auto it = e->state_index.find(name);
  if (it == e->state_index.end()) return fail(NUTLS_ERR_ARG, std::string("unknown state tensor: ") + name);
  StateTensor* st = &e->states[it->second];
  const size_t nb = e->offline ? static_cast<size_t>(e->outt) : static_cast<size_t>(e->B);      // an offline handle: its utterances (carried state of utterance u in arena slot u (offline + 1))
  if (n_floats != st->per_stream() * nb)
    return fail(NUTLS_ERR_ARG, std::string("size mismatch for ") + name + ": expected " + std::to_string(st->per_stream() * nb) +
                                   " floats, got " + std::to_string(n_floats));
  *out = st;
  return NUTLS_OK;
}

int nutls_state_get(nutls_handle* h, const char* name, float* host_buf, size_t n_floats) {
  if (!h || !host_buf) return fail(NUTLS_ERR_ARG, "nutls_state_get: null pointer");
  Engine* e = &h->eng;
  StateTensor* st;
  int rc = state_lookup(e, name, n_floats, &st);
  if (rc) return rc;
  HIP_TRY(hipSetDevice(e->device));
  if (int rcm = states_materialize(e, nullptr)) return rcm;      // (lazily written states: brought up to date before anything outside the kernel looks)
  HIP_TRY(hipDeviceSynchronize());
  if (e->offline) {
    for (int u = 0; u < e->outt; ++u)
      if (int rcu = copy_stream_tensor(e, st->buf[0], st->per_stream(), host_buf + static_cast<size_t>(u) * st->per_stream(), true, u * (e->offline + 1))) return rcu;
    return NUTLS_OK;
  }
  rc = copy_stream_tensor(e, st->buf[1 - e->next_parity], st->per_stream(), host_buf, true);
  if (rc == NUTLS_OK && st->ring_d > 1) rotate_ring(e, *st, host_buf, true);
  return rc;
}

int nutls_state_set(nutls_handle* h, const char* name, const float* host_buf, size_t n_floats) {
  if (!h || !host_buf) return fail(NUTLS_ERR_ARG, "nutls_state_set: null pointer");
  Engine* e = &h->eng;
  StateTensor* st;
  int rc = state_lookup(e, name, n_floats, &st);
  if (rc) return rc;
  HIP_TRY(hipSetDevice(e->device));
  if (int rcm = states_materialize(e, nullptr)) return rcm;      // (lazily written states: brought up to date before anything outside the kernel looks)
  HIP_TRY(hipDeviceSynchronize());
  if (e->offline) {
    for (int u = 0; u < e->outt; ++u)
      if (int rcu = copy_stream_tensor(e, st->buf[0], st->per_stream(), const_cast<float*>(host_buf) + static_cast<size_t>(u) * st->per_stream(), false, u * (e->offline + 1))) return rcu;
    return NUTLS_OK;
  }
  e->ys_dirty = true;      // a conv-input state changed under the fused kernel's carried partial sums: rebuilt before its next step
  // (causal32 CTFA: the 31-frame time-attention history of a streaming handle is library state outside the ABI's tensors.  It is NOT touched
  //  here: nutls_state_set takes [B, ...] buffers, and the per-stream workflow -- get, change one stream's row, set -- must leave the other
  //  B - 1 live streams alone.  A caller that loads a new utterance into stream b calls nutls_reset(h, b) first: nutls.h, nutls_state_set.)
  if (st->ring_d > 1) {
    std::vector<float> tmp(host_buf, host_buf + n_floats);
    rotate_ring(e, *st, tmp.data(), false);
    return copy_stream_tensor(e, st->buf[0], st->per_stream(), tmp.data(), false);
  }
  return copy_stream_tensor(e, st->buf[1 - e->next_parity], st->per_stream(), const_cast<float*>(host_buf), false);
}

/* All state tensors of ONE stream in signature order, concatenated (what the compat runner returns per frame): one
 * device-to-host copy of the stream's `prev`-side state block instead of one copy per tensor. */
int nutls_state_get_all(nutls_handle* h, int stream_idx, float* host_buf, size_t n_floats) {
  if (!h || !host_buf) return fail(NUTLS_ERR_ARG, "nutls_state_get_all: null pointer");
  Engine* e = &h->eng;
  if (e->offline) {      // (offline handles: stream_idx = utterance; its carried state lives in arena slot u (offline + 1))
    if (stream_idx < 0 || stream_idx >= e->outt) return fail(NUTLS_ERR_ARG, "nutls_state_get_all: utterance index out of range");
    stream_idx *= e->offline + 1;
  }
  if (stream_idx < 0 || stream_idx >= e->B) return fail(NUTLS_ERR_ARG, "nutls_state_get_all: stream index out of range");
  size_t total = 0;
  for (const StateTensor& st : e->states) total += st.per_stream();
  if (n_floats != total) return fail(NUTLS_ERR_ARG, "nutls_state_get_all: expected " + std::to_string(total) + " floats");
  HIP_TRY(hipSetDevice(e->device));
  if (int rcm = states_materialize(e, nullptr)) return rcm;      // (lazily written states: brought up to date before anything outside the kernel looks)
  HIP_TRY(hipDeviceSynchronize());
  // Only what is asked for crosses the bus: the buffers of the `prev`-side parity are one contiguous block of the stream's
  // arena slice (allocate_states), the baseline's history rings a second one -- one copy per run of adjacent buffers, then
  // the tensors are picked out of the host image.
  auto want = [&](const StateTensor& st) { return static_cast<size_t>((e->offline ? st.buf[0] : st.buf[1 - e->next_parity]) - e->arena); };
  std::vector<std::pair<size_t, size_t>> runs;      // [begin, end) offsets inside the slice, sorted and merged
  for (const StateTensor& st : e->states) runs.emplace_back(want(st), want(st) + st.per_stream());
  std::sort(runs.begin(), runs.end());
  size_t span = 0, n_runs = 0;
  for (const auto& r : runs) {
    if (n_runs && r.first <= runs[n_runs - 1].second + 1024) runs[n_runs - 1].second = std::max(runs[n_runs - 1].second, r.second);   // (slot padding between neighbours)
    else runs[n_runs++] = r;
    span = std::max(span, r.second);
  }
  runs.resize(n_runs);
  std::vector<float> slice(span);
  for (const auto& r : runs)
    HIP_TRY(hipMemcpy(slice.data() + r.first, e->arena + e->sstride * stream_idx + r.first, (r.second - r.first) * sizeof(float), hipMemcpyDeviceToHost));
  size_t o = 0;
  for (const StateTensor& st : e->states) {
    const float* src = slice.data() + want(st);
    std::memcpy(host_buf + o, src, st.per_stream() * sizeof(float));
    if (st.ring_d > 1) {      // physical ring order -> the reference's oldest-first order
      const size_t frame = st.per_stream() / st.ring_d;
      for (int j = 0; j < st.ring_d; ++j)
        std::memcpy(host_buf + o + j * frame, src + ((e->steps + j) % st.ring_d) * frame, frame * sizeof(float));
    }
    o += st.per_stream();
  }
  return NUTLS_OK;
}

int nutls_reset(nutls_handle* h, int stream_idx) {
  if (!h) return fail(NUTLS_ERR_ARG, "null handle");
  Engine* e = &h->eng;
  if (stream_idx >= (e->offline ? e->outt : e->B)) return fail(NUTLS_ERR_ARG, "nutls_reset: stream index out of range");
  const int utt = stream_idx;          // offline handles: the utterance (or -1: all); its carried state lives in arena slot u (offline + 1)
  if (e->offline && stream_idx >= 0) stream_idx *= e->offline + 1;
  HIP_TRY(hipSetDevice(e->device));
  if (int rcm = states_materialize(e, nullptr)) return rcm;      // (lazily written states: brought up to date before anything outside the kernel looks)
  HIP_TRY(hipDeviceSynchronize());
  // a stream's whole slice of the arena (state of both parities + scratch) is contiguous
  if (stream_idx < 0) HIP_TRY(hipMemset(e->arena, 0, e->sstride * sizeof(float) * e->B));
  else HIP_TRY(hipMemset(e->arena + e->sstride * stream_idx, 0, e->sstride * sizeof(float)));
  if (e->ta_hist) {      // offline handles, causal32 CTFA: the utterance's (all utterances') time-attention history
    const size_t per = static_cast<size_t>(12) * (31 + e->offline) * 64;
    if (utt < 0) HIP_TRY(hipMemset(e->ta_hist, 0, per * e->outt * sizeof(float)));
    else HIP_TRY(hipMemset(e->ta_hist + per * utt, 0, per * sizeof(float)));
  }
  if (e->fz_ta_ring) {      // streaming causal32 CTFA: the stream's (all streams') time-attention history
    const size_t per = static_cast<size_t>(12) * 32 * 64;
    if (stream_idx < 0) HIP_TRY(hipMemset(e->fz_ta_ring, 0, per * e->B * sizeof(float)));
    else HIP_TRY(hipMemset(e->fz_ta_ring + per * stream_idx, 0, per * sizeof(float)));
  }
  if (e->fe_tail) {   // STFT front / back end: previous hop and overlap tail
    const size_t hop = NUTLS_FRAME_STEP * sizeof(float);
    if (stream_idx < 0) {
      HIP_TRY(hipMemset(e->fe_tail, 0, hop * e->B));
      HIP_TRY(hipMemset(e->fe_ola, 0, hop * e->B));
    } else {
      HIP_TRY(hipMemset(e->fe_tail + static_cast<size_t>(NUTLS_FRAME_STEP) * stream_idx, 0, hop));
      HIP_TRY(hipMemset(e->fe_ola + static_cast<size_t>(NUTLS_FRAME_STEP) * stream_idx, 0, hop));
    }
  }
  HIP_TRY(hipDeviceSynchronize());
  return NUTLS_OK;
}

int nutls_debug_get(nutls_handle* h, const char* name, float* host_buf, size_t n_floats) {
  if (!h || !name || !host_buf) return fail(NUTLS_ERR_ARG, "nutls_debug_get: null pointer");
  Engine* e = &h->eng;
  {   // the I/O staging buffers and the STFT phasors are plain [B, n] arrays
    const std::string nm(name);
    const float* src = nullptr;
    size_t per = 0;
    if (nm == "mag_in") { src = e->io_in; per = NUTLS_BINS; }
    else if (nm == "mag_out") { src = e->io_out; per = NUTLS_BINS; }
    else if (nm == "phasor") { src = e->fe_ph; per = 2 * (NUTLS_FRAME_STEP + 1); }
    if (per) {
      if (!src) return fail(NUTLS_ERR_ARG, "debug tensor not allocated yet: " + nm);
      if (n_floats != per * e->B) return fail(NUTLS_ERR_ARG, "size mismatch for debug tensor " + nm);
      HIP_TRY(hipSetDevice(e->device));
      if (int rcm = states_materialize(e, nullptr)) return rcm;      // (lazily written states: brought up to date before anything outside the kernel looks)
      HIP_TRY(hipDeviceSynchronize());
      HIP_TRY(hipMemcpy(host_buf, src, n_floats * sizeof(float), hipMemcpyDeviceToHost));
      return NUTLS_OK;
    }
  }
  if (e->fz_dbg && e->mode == 3) {
    // activation trace of the fused kernel's profiling build (nutls_debug_trace): "<stage>.y" of all 12 stages, "<stage>.up" of the 6
    // decoder stages, "input_layer" -- tensors the fused kernel keeps in LDS
    const std::string nm(name);
    int slot = -1;
    size_t per = 0;
    if (nm == "input_layer") { slot = 0; per = 256 * 64; }
    for (int s = 0; s < 6 && slot < 0; ++s) {
      if (nm == std::string(kEncoder[s].prefix) + ".y") { slot = 1 + s; per = static_cast<size_t>(kEncoder[s].f0) * 64; }
      else if (nm == std::string(kDecoder[s].prefix) + ".y") { slot = 7 + s; per = static_cast<size_t>(kDecoder[s].f0) * 64; }
      else if (nm == std::string(kDecoder[s].prefix) + ".up") { slot = 13 + s; per = static_cast<size_t>(kDecoder[s].f0) * 128; }
    }
    if (slot >= 0) {
      if (n_floats != per * e->B) return fail(NUTLS_ERR_ARG, std::string("size mismatch for debug tensor ") + name);
      HIP_TRY(hipSetDevice(e->device));
      HIP_TRY(hipDeviceSynchronize());
      for (int b = 0; b < e->B; ++b)
        HIP_TRY(hipMemcpy(host_buf + per * b, e->fz_dbg + (static_cast<size_t>(b) * kDbgSlots + slot) * kDbgSlotFloats, per * sizeof(float), hipMemcpyDeviceToHost));
      return NUTLS_OK;
    }
  }
  auto it = e->debug.find(name);
  if (it == e->debug.end()) return fail(NUTLS_ERR_ARG, std::string("unknown debug tensor: ") + name);
  if (n_floats != it->second.second * e->B) return fail(NUTLS_ERR_ARG, std::string("size mismatch for debug tensor ") + name);
  HIP_TRY(hipSetDevice(e->device));
  HIP_TRY(hipDeviceSynchronize());
  return copy_stream_tensor(e, it->second.first, it->second.second, host_buf, true);
}

int nutls_debug_knob(nutls_handle* h, const char* name, int value) {
  if (!h || !name) return fail(NUTLS_ERR_ARG, "nutls_debug_knob: null pointer");
  Engine* e = &h->eng;
  if (std::strcmp(name, "skew") == 0) { e->fz_skew = value; return NUTLS_OK; }
  return fail(NUTLS_ERR_ARG, std::string("nutls_debug_knob: unknown knob ") + name);
}

int nutls_debug_trace(nutls_handle* h, int enable) {
  if (!h) return fail(NUTLS_ERR_ARG, "nutls_debug_trace: null handle");
  Engine* e = &h->eng;
  if (!enable) { e->fz_dbg = nullptr; return NUTLS_OK; }      // (the buffer stays allocated with the handle)
  if (e->offline || !e->fz_blob) return fail(NUTLS_ERR_ARG, "nutls_debug_trace: the activation trace is the fused kernel's (streaming handle, int8 container)");
  if (e->fz_streams != 1) return fail(NUTLS_ERR_ARG, "nutls_debug_trace: the packed plans have no profiling build in the library (nutls_create_plan(..., 1) for the one-stream plan)");
  if (e->B > 64) return fail(NUTLS_ERR_ARG, "nutls_debug_trace: at most 64 streams (2.5 MB of trace per stream)");
  HIP_TRY(hipSetDevice(e->device));
  if (!e->fz_dbg_buf) {
    int rc = dev_alloc(e, static_cast<size_t>(e->B) * kDbgSlots * kDbgSlotFloats, &e->fz_dbg_buf, true);
    if (rc) return rc;
  }
  e->fz_dbg = e->fz_dbg_buf;
  return NUTLS_OK;
}

static const char* family_name(const Launch& L, int B) {
  static const char* conv_names[CONV_KIND_COUNT] = {"conv_el_c32", "conv_el_c64", "conv_el_c128", "conv_dl_n64", "conv_dl_n128",
                                                    "conv_in_c64", "conv_in_c128", "conv_down", "conv_up_even", "conv_up_odd"};
  static std::string tmp[2 * CONV_KIND_COUNT];
  switch (L.kind) {
    case Launch::CONV: {
      const int nw = conv_pick_nw(L.ck, B, L.conv.F_out);
      std::string& t = tmp[2 * L.ck + (nw == 4)];
      t = std::string(conv_names[L.ck]) + (nw == 4 ? "/w4" : "/w1");
      return t.c_str();
    }
    case Launch::LSTM: return "lstm_dense";
    case Launch::CTFA: return "ctfa";
    case Launch::INLAYER: return "input_layer";
    case Launch::OUTCONV: return "out_conv";
    case Launch::DDB: return "dilated_dense";
  }
  return "?";
}

int nutls_launch_info(nutls_handle* h, int index, const char** layer, const char** family, double* flops, double* bytes) {
  if (!h || index < 0 || index >= static_cast<int>(h->eng.plan[0].size())) return fail(NUTLS_ERR_ARG, "nutls_launch_info: bad index");
  const Engine* e = &h->eng;
  const Launch& L = e->plan[0][index];
  const double B = e->B;
  double fl = 0, by = 0;
  switch (L.kind) {
    case Launch::CONV: {
      const ConvShape sh = conv_shape(L.ck);
      const double k = static_cast<double>(sh.tt) * sh.kf * sh.cin, n = 32.0 * sh.nt;
      fl = 2.0 * B * L.conv.F_out * k * n;
      by = 4.0 * B * (static_cast<double>(sh.tt) * L.conv.F_in * sh.cin + L.conv.F_out * n);
      break;
    }
    case Launch::LSTM:
      fl = 2.0 * B * (84.0 * (L.lstm.Din + 21) + 21.0 * L.lstm.Dout);
      by = 4.0 * B * (L.lstm.Din + L.lstm.Dout + 4 * 21);
      break;
    case Launch::CTFA:
      fl = B * (3.0 * L.ctfa.F * 64 + 2.0 * 4 * 64 * 16);
      by = 4.0 * B * 3 * L.ctfa.F * 64;
      break;
    case Launch::INLAYER:
      fl = 2.0 * L.inl.n_pos * 64;
      by = 4.0 * L.inl.n_pos * 65;
      break;
    case Launch::OUTCONV:
      fl = 2.0 * L.outc.n_pos * 64;
      by = 4.0 * L.outc.n_pos * 65;
      break;
    case Launch::DDB: {
      const double F = L.ddb.F, C = L.ddb.C, G = C / 2;
      double mac = 2 * 6.0 * C * G * F;                               // in + out convs
      for (int k = 1; k <= 6; ++k) mac += F * G * (6.0 * k + G);       // grouped dilated conv + 1x1
      fl = 2.0 * B * mac;
      by = 4.0 * B * F * (2 * C + 2 * 321.0 * G);                     // history read + written once per step
      break;
    }
  }
  if (layer) *layer = L.name.c_str();
  if (family) *family = family_name(L, e->B);
  if (flops) *flops = fl;
  if (bytes) *bytes = by;
  return NUTLS_OK;
}

int nutls_profile_step(nutls_handle* h, float* ms, int n) {
  if (!h || !ms) return fail(NUTLS_ERR_ARG, "nutls_profile_step: null pointer");
  Engine* e = &h->eng;
  const int par = e->next_parity;
  const std::vector<Launch>& plan = e->plan[par];
  if (n != static_cast<int>(plan.size())) return fail(NUTLS_ERR_ARG, "nutls_profile_step: n must equal nutls_launches_per_step");
  if (int rc = refuse_per_layer_in_causal32(e, "nutls_profile_step")) return rc;
  HIP_TRY(hipSetDevice(e->device));
  std::vector<hipEvent_t> ev(plan.size() + 1);
  for (auto& x : ev) HIP_TRY(hipEventCreate(&x));
  if (int rc = sync_step_counter(e, e->stream)) return rc;
  if (int rc = states_materialize(e, e->stream)) return rc;
  e->ys_dirty = true;
  HIP_TRY(hipEventRecord(ev[0], e->stream));
  for (size_t i = 0; i < plan.size(); ++i) {
    HIP_TRY(run_launch(plan[i], e->stream));
    HIP_TRY(hipEventRecord(ev[i + 1], e->stream));
  }
  if (e->variant == NUTLS_VARIANT_BASELINE) HIP_TRY(launch_incr_step(e->d_step, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));
  for (size_t i = 0; i < plan.size(); ++i) HIP_TRY(hipEventElapsedTime(&ms[i], ev[i], ev[i + 1]));
  for (auto& x : ev) (void)hipEventDestroy(x);
  e->next_parity = 1 - par;
  e->steps += 1;
  return NUTLS_OK;
}

static bool known_variant(int variant) { return variant == NUTLS_VARIANT_LSTM || variant == NUTLS_VARIANT_BASELINE; }

int nutls_fused_num_ops(int variant) { return known_variant(variant) ? fused_num_ops(variant) : 0; }

int nutls_fused_blob_floats(int variant) { return known_variant(variant) ? fused_blob_floats(variant) : 0; }

/* Host-only (no GPU needed): the weight blob of the fused kernel for a container, for tests of the packing. */
int nutls_fused_pack_blob(const void* weights, size_t n_bytes, int variant, float* out, size_t n_floats) {
  return nutls_fused_pack_blob_plan(weights, n_bytes, variant, 1, out, n_floats);
}

int nutls_fused_plan_num_ops(int variant, int streams) {
  return (known_variant(variant) && fused_has_plan(variant, streams)) ? fused_plan_num_ops(variant, streams) : 0;
}

int nutls_fused_plan_op_info(int variant, int streams, int index, const char** name, double* flops) {
  if (!known_variant(variant) || !fused_has_plan(variant, streams)) return fail(NUTLS_ERR_ARG, "nutls_fused_plan_op_info: no such plan");
  if (index < 0 || index >= fused_plan_num_ops(variant, streams)) return fail(NUTLS_ERR_ARG, "nutls_fused_plan_op_info: bad index");
  if (name) *name = fused_plan_op_name(variant, streams, index);
  if (flops) *flops = fused_plan_op_flops(variant, streams, index);
  return NUTLS_OK;
}

int nutls_fused_plan_blob_floats(int variant, int streams) {
  return (known_variant(variant) && fused_has_plan(variant, streams)) ? fused_plan_blob_floats(variant, streams) : 0;
}

int nutls_fused_pack_blob_plan(const void* weights, size_t n_bytes, int variant, int streams, float* out, size_t n_floats) {
  if (!weights || !out) return fail(NUTLS_ERR_ARG, "nutls_fused_pack_blob: null pointer");
  if (!known_variant(variant)) return fail(NUTLS_ERR_ARG, "nutls_fused_pack_blob: unknown variant");
  if (!fused_has_plan(variant, streams)) return fail(NUTLS_ERR_ARG, "nutls_fused_pack_blob: no plan for that many streams per workgroup");
  if (n_floats != static_cast<size_t>(fused_plan_blob_floats(variant, streams))) return fail(NUTLS_ERR_ARG, "nutls_fused_pack_blob: n_floats must equal nutls_fused_blob_floats()");
  WeightMap wm;
  std::string err;
  std::vector<float> blob;
  try {
    if (!parse_weight_blob(weights, n_bytes, &wm, &err)) return fail(NUTLS_ERR_WEIGHTS, err);
    if (fused_pack_blob(variant, wm, &blob, &err, streams) != FZ_PACK_OK) return fail(NUTLS_ERR_WEIGHTS, err);
  } catch (const std::exception& ex) {
    return fail(NUTLS_ERR_WEIGHTS, std::string("weight container: ") + ex.what());
  }
  std::memcpy(out, blob.data(), blob.size() * sizeof(float));
  return NUTLS_OK;
}

int nutls_fused_op_info(int variant, int index, const char** name, double* flops) {
  if (!known_variant(variant)) return fail(NUTLS_ERR_ARG, "nutls_fused_op_info: unknown variant");
  if (index < 0 || index >= fused_num_ops(variant)) return fail(NUTLS_ERR_ARG, "nutls_fused_op_info: bad index");
  if (name) *name = fused_op_name(variant, index);
  if (flops) *flops = fused_op_flops(variant, index);
  return NUTLS_OK;
}

int nutls_profile_production(nutls_handle* h, double* cum_us, int n, int reps, int steps) {
  if (!h || !cum_us) return fail(NUTLS_ERR_ARG, "nutls_profile_production: null pointer");
  Engine* e = &h->eng;
  if (e->offline || e->mode != 3 || !e->fz_blob || e->variant != NUTLS_VARIANT_LSTM || e->fz_streams != 1 || e->ctfa_causal)
    return fail(NUTLS_ERR_ARG, "nutls_profile_production: a streaming handle of the LSTM variant in the fused mode on the one-stream plan, per-frame CTFA "
                               "(the stop twin exists for that kernel only)");
  const int nops = fused_plan_num_ops(e->variant, 1);
  if (n != nops + 1) return fail(NUTLS_ERR_ARG, "nutls_profile_production: n must equal nutls_fused_num_ops(variant) + 1");
  if (reps < 1 || steps < 1) return fail(NUTLS_ERR_ARG, "nutls_profile_production: reps and steps must be positive");
  HIP_TRY(hipSetDevice(e->device));
  HIP_TRY(fused_step_stop_set_attributes());
  hipEvent_t ev[2];
  HIP_TRY(hipEventCreate(&ev[0]));
  HIP_TRY(hipEventCreate(&ev[1]));
  int rc = NUTLS_OK;
  auto run = [&](int stop, int count) {
    e->fz_stop_at = stop;
    for (int i = 0; i < count && !rc; ++i) {
      rc = run_fused(e, e->next_parity, e->stream, false);
      e->next_parity = 1 - e->next_parity;
      e->steps += 1;
    }
  };
  run(nops, 300);      // (clocks; op index nops is never reached: the whole step)
  for (int stop = 0; stop <= nops && !rc; ++stop) {
    double best = 1e30;
    run(stop, 8);
    for (int r = 0; r < reps && !rc; ++r) {
      if (hipEventRecord(ev[0], e->stream) != hipSuccess) { rc = fail(NUTLS_ERR_HIP, "nutls_profile_production: hipEventRecord"); break; }
      run(stop, steps);
      float ms = 0.f;
      if (rc || hipEventRecord(ev[1], e->stream) != hipSuccess || hipEventSynchronize(ev[1]) != hipSuccess ||
          hipEventElapsedTime(&ms, ev[0], ev[1]) != hipSuccess) { if (!rc) rc = fail(NUTLS_ERR_HIP, "nutls_profile_production: event timing"); break; }
      best = std::min(best, 1e3 * static_cast<double>(ms) / steps);
    }
    cum_us[stop] = best;
  }
  e->fz_stop_at = -1;
  (void)hipEventDestroy(ev[0]);
  (void)hipEventDestroy(ev[1]);
  if (rc) return rc;
  return nutls_reset(h, -1);      // (the truncated launches left every stream's state between two frames)
}

int nutls_profile_fused(nutls_handle* h, double* us, int n) {
  if (!h || !us) return fail(NUTLS_ERR_ARG, "nutls_profile_fused: null pointer");
  Engine* e = &h->eng;
  if (n != fused_plan_num_ops(e->variant, e->fz_streams))
    return fail(NUTLS_ERR_ARG, "nutls_profile_fused: n must equal nutls_fused_plan_num_ops(variant, nutls_streams_per_workgroup(h))");
  HIP_TRY(hipSetDevice(e->device));
  const int par = e->next_parity;
  int rc = run_fused(e, par, e->stream, true);
  if (rc) return rc;
  HIP_TRY(hipStreamSynchronize(e->stream));
  std::vector<unsigned long long> t(n + 1);
  HIP_TRY(hipMemcpy(t.data(), e->fz_prof, t.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  int khz = 100000;
  (void)hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, e->device);
  if (khz <= 0) khz = 100000;
  for (int i = 0; i < n; ++i) us[i] = static_cast<double>(t[i + 1] - t[i]) * 1000.0 / khz;
  e->next_parity = 1 - par;
  e->steps += 1;
  if (const char* wt = getenv("NUTLS_FUSED_WTRACE")) {       // debugging aid: the raw per-wave trace [8 waves][ops][12] of an FZ_WTRACE build (zeros otherwise)
    std::vector<unsigned long long> tr(static_cast<size_t>(n) * 8 * 12);
    HIP_TRY(hipMemcpy(tr.data(), e->fz_prof + static_cast<size_t>(n) * 9 + 1, tr.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    if (FILE* f = fopen(wt, "wb")) {
      fwrite(tr.data(), sizeof(unsigned long long), tr.size(), f);
      fclose(f);
    }
  }
  if (const char* dump = getenv("NUTLS_FUSED_PHASES")) {     // debugging aid: phase stamps of every conv op (wave 0 of workgroup 0)
    std::vector<unsigned long long> sub(static_cast<size_t>(n) * 8);
    HIP_TRY(hipMemcpy(sub.data(), e->fz_prof + n + 1, sub.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    if (FILE* f = fopen(dump, "w")) {
      for (int i = 0; i < n; ++i) {
        fprintf(f, "%-24s total %6.2f |", fused_plan_op_name(e->variant, e->fz_streams, i), us[i]);
        // stamp slots in chronological order: 0 loads issued, 5 carried weights arrived, 6 MFMA loop done (4x4 path),
        // 1 partials / parameters written, 2 past barrier 1, 3 epilogue done, 4 next image built
        const int order[7] = {0, 5, 6, 1, 2, 3, 4};
        const char* nm_conv[7] = {"issue", "wwait", "mloop", "mfma", "bar1", "epi", "build"};
        // CTFA ops: loads issued | column sums | barrier | time-attention perceptron | frequency-attention perceptron + gate | barrier; the rest (bar2) = gate applied
        const char* nm_ctfa[7] = {"issue", "colsum", "-", "bar1", "mlp_ta", "mlp_fa", "barg"};
        const char* opn = fused_plan_op_name(e->variant, e->fz_streams, i);
        const size_t ol = std::strlen(opn);
        const char* const* nm = (ol >= 4 && std::strcmp(opn + ol - 4, "ctfa") == 0) ? nm_ctfa : nm_conv;
        unsigned long long prev = t[i];
        for (int k = 0; k < 7; ++k) {
          const unsigned long long v = sub[8 * i + order[k]];
          if (v >= prev && v <= t[i + 1]) { fprintf(f, " %s %5.2f", nm[k], static_cast<double>(v - prev) * 1000.0 / khz); prev = v; }
        }
        fprintf(f, " bar2 %5.2f\n", static_cast<double>(t[i + 1] - prev) * 1000.0 / khz);
      }
      fclose(f);
    }
  }
  return NUTLS_OK;
}


}  // extern "C"
