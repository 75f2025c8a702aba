// Internal declarations shared by the host engine (engine.cpp), the weight container
// parser (weights.cpp) and the gfx950 kernels (kernels.hip).  Not part of the C ABI.
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>
#include <map>
#include <string>
#include <vector>

namespace nutls {

// ------------------------------------------------------------------ weight container ----------
struct HostTensor {
  std::vector<int> dims;
  std::vector<float> data;  // de-quantised (int8 * scale), exactly the graph's DEQUANTIZE
  // int8 tensors of the container also keep their raw form: data[k] == q[k] * scales[per-channel ? k / inner : 0]
  std::vector<int8_t> q;
  std::vector<float> scales;   // 1 entry (per tensor) or dims[0] entries (per output channel)
  size_t size() const { return data.size(); }
};
using WeightMap = std::map<std::string, HostTensor>;

// Parses a NUTLSW01 container (layout: tools/convert_tflite_weights.py).  Returns false and
// fills `err` on malformed input.
bool parse_weight_blob(const void* blob, size_t n, WeightMap* out, std::string* err);

// ------------------------------------------------------------------ kernel parameters ---------
// Memory layout: every per-stream tensor (state, scratch) lives in ONE arena laid out stream-major,
// element (stream b, row r, channel c) of a tensor at  base + b*sstride + r*ld + c.  A stream's
// whole working set (~2.6 MB) is contiguous, so the workgroup that owns the stream stays inside a
// couple of 2 MiB pages.
// Block mode with several utterances per handle (nutls_create_offline_batch): the launch's "streams" are the frames of all utterances, dense
// index b = u * n + t (utterance u, frame t of the n frames of this launch), but an utterance's frames live in ITS run of arena slots -- the
// slot in front of its first frame holds its carried state, the tap "one frame earlier" of every layer -- so consecutive utterances are
// `gap` slots further apart than their frames: slot(b) = b + (b / n) * gap.  n = 0: dense (every streaming launch, one-utterance blocks).
struct SlotMap {
  int n; unsigned mul; int gap;      // mul = floor(2^32 / n) + 1: b / n = umulhi(b, mul), exact for b * n < 2^32 (n = 1: mul = 0, the quotient is b)
};
#if defined(__HIPCC__)
__device__ __forceinline__ size_t slot_of(int b, const SlotMap& m) {
  return m.n ? static_cast<size_t>(b) + static_cast<size_t>(m.mul ? __umulhi(static_cast<unsigned>(b), m.mul) : static_cast<unsigned>(b)) * m.gap : static_cast<size_t>(b);
}
// (utterance of stream b)
__device__ __forceinline__ unsigned utt_of(int b, const SlotMap& m) { return m.n ? (m.mul ? __umulhi(static_cast<unsigned>(b), m.mul) : static_cast<unsigned>(b)) : 0u; }
#endif
inline SlotMap make_slot_map(int n, int gap) {
  if (n <= 0 || gap <= 0) return SlotMap{0, 0u, 0};
  return SlotMap{n, n == 1 ? 0u : static_cast<unsigned>((1ull << 32) / static_cast<unsigned>(n)) + 1u, gap};
}

// Generic "tap-GEMM" convolution over channels-last rows (see kernels.hip for the tiling).
struct ConvParams {
  const float* src0;   // time tap 0 (previous frame) -- or the only input when TT == 1
  const float* src1;   // time tap 1 (current frame)
  const float* wpk;    // weights in MFMA fragment order (pack_conv_weights)
  const float* wbf;    // int8 containers: the int8 weights widened to bf16 in 32x32x16 fragment order (pack_conv_weights_bf16), else null
  const float* wscale; // ... and their per-output-channel scale, packed channel order [32*NT]
  int use_bf16;        // launch the bf16-pipe kernel (block mode; needs wbf / wscale)
  const float* bias;   // [32*NT]  packed channel order
  const float* gamma;  // [32*G]   LayerNorm scale (EPI_LN only)
  const float* beta;   // [32*G]
  float* dst0;         // first destination (never null)
  float* dst1;         // optional second destination (skip-connection copy) or null
  int src_ld;          // floats between consecutive input rows
  int ld0, ld1;        // floats between consecutive output rows of dst0 / dst1
  int B, F_in, F_out;  // streams, input rows per stream, output positions per stream
  int log2_fout;
  int row_mul, row_add;  // output row of position f, group g:  f*row_mul + row_add + g
  float alpha;           // PReLU slope
  long long sstride;     // floats between consecutive streams (all per-stream tensors share one arena stride)
  SlotMap sm;            // stream index -> arena slot (block mode with several utterances; {0, 0, 0}: slot = stream)
};

enum ConvKind : int {
  CONV_EL_C32 = 0,   // (2,3) stride-2, Cin 32  -> 32, LN+PReLU      (encoder levels >= 2)
  CONV_EL_C64,       // (2,3) stride-2, Cin 64  -> 32                 (encoder level 1, decoder levels >= 2)
  CONV_EL_C128,      // (2,3) stride-2, Cin 128 -> 32                 (decoder level 1)
  CONV_DL_N64,       // (2,3) stride-1, Cin 64 -> 64 conv ch = 2 rows x 32, LN(32)+PReLU  (sub-pixel)
  CONV_DL_N128,      // (2,3) stride-1, Cin 64 -> 128 conv ch = 2 rows x 64, LN(64)+PReLU (last sub-pixel)
  CONV_IN_C64,       // 1x1, Cin 64  -> 64, LN(64)+PReLU
  CONV_IN_C128,      // 1x1, Cin 128 -> 64, LN(64)+PReLU
  CONV_DOWN,         // (1,3) stride-2, pad right, 64 -> 64, bias only
  CONV_UP_EVEN,      // transposed (1,3) stride-2: even output rows  (taps x[i-1], x[i]), 128 -> 128
  CONV_UP_ODD,       // transposed (1,3) stride-2: odd output rows   (tap x[i]),          128 -> 128
  CONV_KIND_COUNT
};

struct ConvShape {  // static description of a ConvKind
  int cin, nt, stride, tt, kf, padl, epi_ln, g;
};
ConvShape conv_shape(ConvKind k);
// bytes of dynamic LDS one workgroup of `nw` waves needs for this kind at F_out
size_t conv_lds_bytes(ConvKind k, int f_out, int nw);
int conv_pick_nw(ConvKind k, int B, int f_out);
hipError_t launch_conv(ConvKind k, const ConvParams& p, hipStream_t s);

struct LstmParams {
  const float* x; int x_ld, x_rows, x_cols;   // v[f*x_cols+c] = x[(b*x_rows+f)*x_ld + c]
  const float* wxT;   // [Din][84]
  const float* whT;   // [21][84]
  const float* bias;  // [84]
  const float* wdT;   // [21][Dout]
  const float* bd;    // [Dout]
  const float* h_in; const float* c_in; float* h_out; float* c_out;   // [B,21]
  float* dst; int dst_ld, dst_rows, dst_cols;  // y[f*dst_cols+c] -> dst[(b*dst_rows+f)*dst_ld + c]
  int Din, Dout, B;
  long long sstride;
  SlotMap sm;
};
hipError_t launch_lstm(const LstmParams& p, hipStream_t s);
// offline / block mode (offline.hip): the same cell over `frames` consecutive frames of one utterance
constexpr int kScanReadAhead = 16;   // rows the scan's prefetch reads past the frames it was given (values never used): slack rows of the zx buffer
// several utterances: `frames` frames of each of `utts` utterances (dense index u * frames + t, p.sm maps it to slots); `utt_stride` floats between the
// slot runs of consecutive utterances (h / c of utterance u start there)
hipError_t launch_lstm_block(const LstmParams& p, float* zx /* [utts * frames + kScanReadAhead][84] scratch */, int frames, hipStream_t s, int utts = 1,
                             long long utt_stride = 0);

// Dilated-dense bottleneck of the baseline variant (see ddb_device.hpp).  Weights transposed so the
// output channel is the fastest index (consecutive threads read consecutive floats).
struct DdbParams {
  const float* x; int x_ld;          // input rows [F][C]
  float* dst; int dst_ld;            // output rows [F][C]
  float* st_in;                      // [1][F][C]       previous input (updated in place)
  float* st_blk[6];                  // [d][F][k*G]     ring of the last d = 2^(k-1) frames of block k's input
  float* st_out;                     // [1][F][G]
  const float* w_in; const float* b_in; float a_in;        // [t][kw][c][g]
  const float* wg[6]; const float* bg[6];                  // [t][kw][j][g]
  const float* w1[6]; const float* b1[6];                  // [gin][gout]
  const float* gamma[6]; const float* beta[6]; float alpha[6];
  const float* w_out; const float* b_out; float a_out;     // [t][kw][g][c]
  const float* wsmall;               // wg (all blocks), w1 (all blocks), {bg,b1,gamma,beta} (all blocks) packed in the LDS order of ddb_block_wg
  const int* step;                   // device-resident frame counter (ring position)
  int F, C, B;
  long long sstride;
};
hipError_t launch_ddb(const DdbParams& p, hipStream_t s);
hipError_t launch_incr_step(int* step, hipStream_t s);
hipError_t launch_set_step(int* step, int value, hipStream_t s);   // after fused-mode steps (which take the counter by value)

// Carried partial sums of the fused kernel's two-tap convs (fused_plan.hpp OpD::ys): S = W[time tap 0] x, rebuilt on the device
// from the conv-input state tensors after those were written from outside the kernel (nutls_state_set, a step of another mode).
struct YsOp {
  int P, cin, N, stride;     // output positions, input channels, packed output channels, 2 (strided conv) or 1 (sub-pixel conv)
  int xs_off, xs_ld;         // input state tensor: float offset inside a parity block of the arena, floats per row
  int ys_off;                // the op's sums inside a block of partial sums ([pos][N])
  int w_off;                 // weights [N][3][cin] (packed channel order, int8 values as floats) inside the table's weight array
  int r32, PT, NT, PG;       // r32: the op runs on 32x32 tiles and keeps its sums in accumulator order, [wave task][pt][n][q][lane] float4
                             // (task = position group + PG * channel group); else [pos][N]
};
hipError_t launch_ysum_refresh(const float* arena, long long sstride, int x_block_off, int ys_block_off, const YsOp* ops, const float* w,
                               int n_ops, int B, hipStream_t s);

struct CtfaParams {
  const float* x; int x_ld;      // d_D  [B,F,64]
  const float* e0; int e0_ld;    // residual [B,F,64]
  float* y; int y_ld;            // out  [B,F,64]
  const float* ta_w1T; const float* ta_b1; const float* ta_w2T; const float* ta_b2;  // [64][16],[16],[16][64],[64]
  const float* fa_w1T; const float* fa_b1; const float* fa_w2T; const float* fa_b2;
  const float* ta_w2; const float* fa_w2;   // [64][16] (output channel major), as stored
  int B, F;
  long long sstride;
  SlotMap sm;
};
hipError_t launch_ctfa(const CtfaParams& p, hipStream_t s);
// offline / block mode, true 32-frame causal average of the time attention (models/proposed.py:143-147); hist [31 + p.B][64]
// `hist` = the history row of the first frame's predecessor-window start; `roll`: the block ends here (see launch_ctfa_hist_roll)
// (several utterances: p.B = utts * frames, p.sm maps frames to slots, utterance u's history hist_ustride floats after utterance u - 1's)
hipError_t launch_ctfa_causal(const CtfaParams& p, float* hist, bool roll, hipStream_t s, int utts = 1, long long hist_ustride = 0);
hipError_t launch_ctfa_hist_roll(float* hist, int frames, hipStream_t s, int utts = 1, long long hist_ustride = 0);

struct InLayerParams {   // input_layer: 1x1 conv 1->64 + LN + PReLU
  const float* x;        // [B,256]
  float* y;              // [B,256,64]
  const float* w; const float* b; const float* gamma; const float* beta; float alpha;
  int n_pos;             // B*256
  long long sstride;     // stream stride of y (x is the plain [B,256] input)
  SlotMap sm;
  SlotMap sm_io;         // frame index -> row of x (a chunk of a block of several utterances: its frames of utterance u sit n_frames rows after those of u - 1)
};
hipError_t launch_input_layer(const InLayerParams& p, hipStream_t s);

struct OutConvParams {   // 1x1 conv 64->1
  const float* x; int x_ld;   // [B,256,64]
  float* y;                   // [B,256]
  const float* w; float bias;
  int n_pos;
  long long sstride;          // stream stride of x (y is the plain [B,256] output)
  SlotMap sm;
  SlotMap sm_io;              // frame index -> row of y (see InLayerParams)
};
hipError_t launch_out_conv(const OutConvParams& p, hipStream_t s);


// ------------------------------------------------------------------ streaming loop constants / front end ---------
constexpr int NUTLS_DEV_BINS = 256;
constexpr int NUTLS_FRAME_LEN = 512;    // interpreter_proposed.py:17
constexpr int NUTLS_FRAME_STEP = 256;   // interpreter_proposed.py:18
// STFT front end / inverse-STFT + overlap-add back end of the streaming loop (stft.hip)
hipError_t launch_stft_hop(const float* pcm, float* tail, const float* win, const float* tw, float* mag, float* ph, int B, hipStream_t s);
hipError_t launch_istft_hop(const float* est, const float* ph, const float* inv_win, const float* tw, float* ola, float* pcm_out,
                            int dc_edge, int B, hipStream_t s);
// (Rounds 1-3 kept a third kernel family here: a persistent plan-interpreter kernel, megakernel.hip, execution mode 2 -- retired in
//  round 4: the fused kernel is the one-launch path, the per-layer kernels below are the cross-check and the block mode.)

// Packs OHWI conv weights [Cout][th][kw][Cin] into the order the MFMA loop streams them.
//   perm[n'] = original output channel feeding packed channel n'
//   taps     = list of (t, kw) source taps in kernel order (time-major)
std::vector<float> pack_conv_weights(const HostTensor& w, const std::vector<int>& perm,
                                     const std::vector<std::pair<int, int>>& taps_per_t,
                                     int tt, int cin, int nt);
// Same, for v_mfma_f32_16x16x4_f32 tiles: fragment (16-channel K group g, 16-row channel tile rt), lane l
// holds W[16 rt + (l & 15)][16 g + 4 (l >> 4) + j], j = 0..3.  Order [t][chunk][kf][g16][rt][lane][4].
std::vector<float> pack_conv_weights_bf16(const HostTensor& w, const std::vector<int>& perm,
                                          const std::vector<std::pair<int, int>>& taps_per_t, int tt, int cin, int nt);

// ------------------------------------------------------------------ fused (statically scheduled) kernel ----
// fused_step.hip (LSTM variant) / fused_base.hip (baseline variant) / fused_host.cpp: the frame step as one specialised
// instruction stream per op.  `prof` non-null selects the profiling build; `ddb` is the baseline's block table (else null);
// `step` = frames processed so far (ring position of the baseline's dilated-dense histories), by value: ONE launch per step.
// CTFA frequency branch of the fused kernels.  Frame mode (default; ctfa_rt with T = 1, proposed.py:162-196): `sum` is one
// row of 64 zeros and `ring` one dump row, all strides 0 -- the branch sees TA / 32.  Causal32 mode (the offline model's `ctfa`,
// proposed.py:125-160): `sum` [B][12][64] = the time attention summed over the 31 frames before this one (ta_sum_kernel, launched in
// front of the step), `ring` points at this frame's row of the history [B][12][32][64] (row = frame mod 32); strides in floats.
struct FzTa {
  const float* sum; float* ring; int sum_sstride, sum_gstride, ring_sstride, ring_gstride;
  // (rides along: start skew of the workgroups, in units of 64 clocks -- workgroup w sleeps (w mod 4) * skew before its first op, so that
  //  the HBM bursts of the op sequence, which all workgroups of a launch walk in lock step, spread out; 0 = off.  NUTLS_FUSED_SKEW.)
  int skew;
  // (rides along too: 1 = the launch also writes the state tensors nothing in the kernel reads -- the input states of the strided convs,
  //  OpD::d0_on = 2 --, 0 = it leaves them to engine.cpp states_materialize, which rebuilds them from their second copy when asked)
  int eager;
  // (and: activation trace of the PROFILING builds, nutls_debug_trace -- the outputs that never touch HBM in the fused kernel (input layer, the
  //  12 CTFA outputs, the 6 up-sampling outputs) are copied to `dbg` + stream * dbg_sstride + slot * kDbgSlotFloats; null = off.  The
  //  production kernels do not look at these fields.)
  float* dbg; long long dbg_sstride;
};
// activation trace slots (floats per slot: the largest traced tensor, 256 x 128): 0 input layer [256][64]; 1 + k: output of CTFA k [F0][64]
// (encoder stages 0..5, decoder stages 6..11); 13 + s: output of the up-sampling conv of decoder stage s [F0][128]
constexpr int kDbgSlotFloats = 256 * 128, kDbgSlots = 19;
// the lazily written state tensors of a fused plan: P rows x 32 channels, copied from the skip-connection slice the kernel does write
struct LazyCopy { int src_off, src_ld, dst_off, dst_ld, rows, width; };      // float offsets inside a parity block of the arena
hipError_t launch_lazy_states(float* arena, long long sstride, int block_off, const LazyCopy* tab, int n, int B, hipStream_t s);
hipError_t launch_ta_sum(const float* ring, float* sum, int slot, int B, hipStream_t s);
hipError_t launch_fused_step(float* arena, long long sstride, const float* blob, const float* io_in, float* io_out, int B, int par,
                             unsigned long long* prof, const DdbParams* ddb, int step, int grid, hipStream_t s, const FzTa& ta);
hipError_t launch_fused_base_step(float* arena, long long sstride, const float* blob, const float* io_in, float* io_out, int B, int par,
                                  unsigned long long* prof, const DdbParams* ddb, int step, int grid, hipStream_t s, const FzTa& ta);
hipError_t fused_step_set_attributes();
// stop twin (fused_step_stop.hip): ta.skew = the op in front of which the launch ends; `prof` must be null
hipError_t launch_fused_step_stop(float* arena, long long sstride, const float* blob, const float* io_in, float* io_out, int B, int par,
                                  unsigned long long* prof, const DdbParams* ddb, int step, int grid, hipStream_t s, const FzTa& ta);
hipError_t fused_step_stop_set_attributes();
hipError_t fused_base_step_set_attributes();
void fused_lazy_table(int variant, std::vector<LazyCopy>* tab);      // (one-stream plans; empty when the plan writes every state)
enum FusedPack : int { FZ_PACK_OK = 0, FZ_PACK_NOT_INT8 = 1, FZ_PACK_MALFORMED = 2 };
int fused_pack_blob(int variant, const WeightMap& wm, std::vector<float>* out, std::string* err, int streams = 1);      // -> FusedPack
// packed plans (several streams per workgroup, fused_plan.hpp OpD::gs): the LSTM variant has one for 2 streams (fused_step_g2.hip).  Same
// arena layout as the one-stream plan (checked in fused_setup); the blob and the layout of the carried sums are the plan's own.
bool fused_has_plan(int variant, int streams);
int fused_plan_blob_floats(int variant, int streams);
int fused_plan_arena_floats(int variant, int streams);
int fused_plan_parity_stride(int variant, int streams);
int fused_plan_ys_off(int variant, int streams);
int fused_plan_ys_block(int variant, int streams);
int fused_plan_num_ops(int variant, int streams);
const char* fused_plan_op_name(int variant, int streams, int i);
double fused_plan_op_flops(int variant, int streams, int i);
hipError_t launch_fused_step_g2(float* arena, long long sstride, const float* blob, const float* io_in, float* io_out, int B, int par,
                                unsigned long long* prof, const DdbParams* ddb, int step, int grid, hipStream_t s, const FzTa& ta);
hipError_t fused_step_g2_set_attributes();
hipError_t launch_fused_step_g4(float* arena, long long sstride, const float* blob, const float* io_in, float* io_out, int B, int par,
                                unsigned long long* prof, const DdbParams* ddb, int step, int grid, hipStream_t s, const FzTa& ta);
hipError_t fused_step_g4_set_attributes();
int fused_blob_floats(int variant);
int fused_num_ops(int variant);
const char* fused_op_name(int variant, int i);
double fused_op_flops(int variant, int i);
int fused_parity_stride(int variant);
int fused_arena_floats(int variant);
int fused_num_states(int variant);
int fused_num_pingpong(int variant);          // the first so many states are ping-pong pairs, the rest in-place history rings
const char* fused_state_name(int variant, int i);
int fused_state_off(int variant, int i);
int fused_num_scratch(int variant);
const char* fused_scratch_name(int variant, int i);
int fused_scratch_off(int variant, int i);
int fused_ys_block(int variant);          // floats of one block of carried partial sums; the arena's "ysum" scratch holds two
int fused_ys_off(int variant);            // arena offset of the first block
// the table launch_ysum_refresh needs (w: all ops' tap-0 weights, int8 values as floats); false + err if a tensor has no int8 payload
bool fused_ys_table(int variant, const WeightMap& wm, std::vector<YsOp>* ops, std::vector<float>* w, std::string* err, int streams = 1);

}  // namespace nutls
