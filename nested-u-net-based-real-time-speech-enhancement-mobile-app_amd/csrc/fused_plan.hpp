// Static schedule of the fused ("one specialised instruction stream per op") frame-step kernel.
//
// The NUNet-TLS-LSTM step has a fixed topology (reference: TFL_SIGNITURE.nutls_lstm,
// dnn_model/converter_proposed.py:188-867): every shape, every LDS address and every offset into the
// per-stream HBM arena / the weight blob is known before the kernel is compiled.  tools/gen_fused_plan.py
// writes them down as one constexpr record per op (fused_plan_lstm.inc); fused_step.hip instantiates one
// template per record, so the kernel carries no run-time plan decoding at all.
#pragma once

namespace nutls {
namespace fz {

enum OpType : int { T_INPUT = 0, T_CONV = 1, T_LSTM = 2, T_CTFA = 3, T_DDB = 4 };
enum CKind : int { K_IN = 0, K_EL = 1, K_DL = 2, K_DOWN = 3, K_UP = 4 };
// P_R32: 32x32x2 fp32 MFMA tiles on an fp32 image (the layers whose image does not fit LDS as three bf16 planes);
// P_R32B / P_X16B: 32x32x16 / 16x16x32 bf16 MFMA tiles on a three-plane bf16 image (x = hi + mid + lo exactly, see fused_step.hip)
enum Path : int { P_R32 = 0, P_R32B = 3, P_X16B = 4 };
enum Src : int { S_PREV = 0, S_CUR = 1, S_SCRATCH = 2 };   // HBM base a float offset is relative to

constexpr int LDS_BYTES = 160 * 1024;
constexpr int SCR_BYTES = 8192;                      // LSTM / CTFA scratch at the top of LDS
constexpr int SCR_B = LDS_BYTES - SCR_BYTES;
constexpr int MAX_PARTS = 6, MAX_ZERO = 12, MAX_SEG = 6;
constexpr int RING_SF_R32 = 6;                       // ... of the large layers (their K steps are 5x shorter than the fp32 MFMA's were)
constexpr int XCOPY_B = SCR_B + 7168;                // fp32 copy of the rows an LSTM / dilated-dense op reads (<= 256 floats), written by the conv op before it
constexpr int RING_SF = 3;                           // weight ring of a wave: 3 x (dwordx4 per lane = 16 int8 weights: 4 fp32-MFMA or 2 bf16-MFMA fragments), first fill by the previous op

// A rectangular block of an HBM tensor that is copied into an LDS image through registers.
struct Part {
  int src, off, ld;      // HBM: base selector, float offset of (row 0, first channel), floats between rows
  int rows, c4s;         // block size: rows x (c4s float4)
  int lds_b;             // LDS byte address of image row 0 at the block's first channel (three-plane images: in the hi plane)
  int row0;              // image row of block row 0
  int la;                // 1: loads issued by the op that builds the image, 2: one op earlier
  int round2;            // belongs to the second round of a two-round image
  // packed plans (several streams per workgroup): the block exists once per stream g0 .. g0 + ng - 1 of the workgroup -- in HBM one
  // arena slice apart, in LDS gstride_b bytes apart (the sub-images of the streams).  One-stream plans: 0, 1, 0.
  int g0, ng, gstride_b;
};
struct Zero { int lds_b, n4; };   // halo: n4 float4 of zeros

// LDS image (B operand of the MFMAs) of a conv op: rows x channels per time tap, padded pitch.
//   fmt 0 (fp32):            address(lr, c)    = tap * tap_b + row(lr) + 4 c
//   fmt 1 (3 bf16 planes):   address(lr, c, p) = tap * tap_b + row(lr) + p * plane_b + 2 c      p = 0 hi, 1 mid, 2 lo
//   row(lr) = pair ? (lr >> 1) * pitch_b + (lr & 1) * half_b : lr * pitch_b
struct Img {
  int fmt, plane_b;
  int taps, tap_b, pitch_b, pair, half_b, row0, bytes;
  int gstride_b;         // packed plans: LDS bytes between the sub-images of consecutive streams of the op (0: one stream)
  int nparts; Part parts[MAX_PARTS];
  int nzero; Zero zero[MAX_ZERO];
};

// Where an op's output rows go inside the LDS image of the op that consumes them next.
// Packed plans: the rows of the op's stream i (0 <= i < OpD::gs) go to base_b + i * gstride_b if bit i of `mask` is set; the other streams'
// rows reach their consumer through HBM (a staged part of its image).
struct Fwd { int on, base_b, pitch_b, pair, half_b, row0, fmt, plane_b, gstride_b, mask; };

struct OpD {
  int type;
  // ---- conv ------------------------------------------------------------------------------------
  int kind, P, cin, N, taps, kf, stride;   // P output positions, N conv channels (UP: per output-row parity)
  int path, PT, NT, PG, CG, KSt, KSg;      // tiling: PT x NT tiles per wave, PG x CG x (KSt x KSg) wave tasks
  int ln, R, gc;                           // LayerNorm+PReLU?, output rows per position, channels per output row
  int rounds;                              // 2: the two time taps use the same LDS region one after the other
  int nseg, seg_b[MAX_SEG];                // (tap, frequency tap) segments of K: byte offset of each inside the image
  int ex_b;                                // exchange buffer (X16 / X4 paths)
  int w_off, p_off;                        // weight blob (floats): MFMA fragments; bias | gamma | beta | alpha
  int d0_on, d0_src, d0_off, d0_ld, d1_on, d1_src, d1_off, d1_ld;   // HBM destinations (row 0, first channel)
                                           // (_on 2: a state tensor the kernel itself never reads -- written only when the launch asks for eager states,
                                           //  else rebuilt by the library from the rows' other copy when somebody looks: engine.cpp states_materialize)
  int row_mul, row_add;                    // output row of position p, sub-row r:  p * row_mul + row_add + r
  Fwd fwd;
  Img img;                                 // geometry of this op's own image + how it is staged
  int nxt;                                 // op whose image this op completes (forward + staging), -1: none
  // ---- lstm ------------------------------------------------------------------------------------
  int din, dout, x_b, x_pitch_b, x_cols, y_b, h_off, c_off, ldst_on, ldst_off, ldst_ld;
  int x_fmt, x_plane_b;                    // format of the image the op works in place on (Img::fmt / plane_b)
  int lw_off;                              // wxT | whT | bias | wdT | bd
  // ---- ctfa ------------------------------------------------------------------------------------
  int F, e0_off, e0_ld, last, cw_off;      // operates in place on fwd-described rows; cw: ta(w1T,b1,w2,b2) | fa(...)
  // ---- all -------------------------------------------------------------------------------------
  int drain;                               // every wave drains its memory counter before the op's last barrier
  int bidx;                                // T_DDB (baseline variant): which of the 13 dilated-dense bottlenecks (uses x_cols, y_b, x_pitch_b);
                                           // T_CTFA: which of the 12 CTFAs (encoder stages 0..5, decoder stages 6..11): row of FzTa::sum / ring;
                                           // up-sampling convs: which decoder stage (slot of the layer's output in the activation trace, FzTa::dbg)
  // ---- carried partial sums (two-tap convs) ------------------------------------------------------
  int ys;                                  // 1: y_t = W[tap 1] x_t + S_{t-1} with S_t = W[tap 0] x_t: the image holds x_t only, S travels through HBM as
                                           // P x N fp32 ([pos][packed channel], unscaled integer-weight sums), block kYsOff + parity * kYsBlock
  int ys_off;                              // float offset of this op's sums inside a block
  int xs_off, xs_ld;                       // the conv's input state tensor [rows][xs_ld] (written for the ABI, never read by the kernel; the
                                           // host rebuilds S from it after nutls_state_set / a step of another mode)
  // ---- packed plans: several streams per workgroup (kStreams > 1) ---------------------------------
  // The op computes streams g0 .. g0 + gs - 1 of the workgroup's kStreams: gs = kStreams -- "side by side", the streams' positions are
  // one virtual position axis of gs * P (the weights are fetched and converted once for all of them, every stream has its own
  // sub-image, Img::gstride_b apart) -- or gs = 1: one instance of the layer per stream, one after the other (the layers whose
  // images do not fit LDS gs times).  One-stream plans: gs = 1, g0 = 0.
  int gs, g0;
  int scr_b;                               // LSTM / CTFA scratch of stream i at scr_b + i * scr_gstride_b (one-stream plans: SCR_B, 0)
  int scr_gstride_b;
  int xcopy_b;                             // fp32 copy of the rows an LSTM / dilated-dense op reads, stream i at xcopy_b + i * 1024 (one-stream plans: XCOPY_B)
  int x_gstride_b;                         // LSTM: LDS bytes between the sub-images (of the conv that follows) it writes its streams' rows into
  int layer;                               // index of the layer (= op index of the one-stream plan) this op is an instance of
  // ---- small 16x16-tile conv ops (planner: tools/gen_fused_plan.py conv_op) ---------------------------------------------------------
  // epl 1: row-wise epilogue with ONE output element per lane (a row of gc channels = gc consecutive lanes) for layers with at most 512
  // outputs: the K-slice sums, LayerNorm, PReLU and the three-plane split are ~50 dependent VALU instructions per wave instead of ~110
  // on the one or two waves that hold all rows as float4 (epl 4) -- the epilogue is the critical path of a small op
  int epl;
  // nt0 / nt1: nothing in THIS launch reads destination 0 / 1 again (its rows are the next frame's previous-frame tap): stored with the non-temporal hint
  // (fused_step.hip "Cache policy"); LSTM ops: nt0 for the Dense rows written to the state tensor
  int nt0, nt1;
};
// ---- blob layout of an LSTM + Dense op (OpD::lw_off; kernel: prefetch_w / lstm_op, host: fused_host_impl.inc, planner: gen_fused_plan.py) --
// The reference's .tflite stores the LSTM kernels int8 with one scale per tensor (FULLY_CONNECTED, hybrid) and so does the blob: four
// times fewer bytes through the CU's memory pipe than fp32 (an LSTM's 51 KB of fp32 parameters were a 1 us burst in the op that prefetches
// them).  z = b + s_x (Qx x) + s_h (Qh h): the scale is applied to the K-slice sums, as in the convs.
//   gates  : [20 K slices][21 units][NRP dwords], dword j = the (i, f, g, o) int8 weights of the unit for row j of the slice -- slices 0..15: KN =
//            din / 16 rows of x each, 16..19: 6 rows of h each (rows 21..23 zero); NRP = the row count rounded up to whole float4s
//   record : [21][4] fp32 bias (i, f, g, o) | s_x, s_h, 0, 0
//   dense  : per output n 8 dwords -- 24 int8 (21 weights, 3 zero) | fp32 bias | fp32 scale (dout >= 64: the tensors TF-Lite quantises, >= 1024
//            elements), or 24 fp32 (21 weights, bias, 0, 0) for the 32 x 21 ones it leaves in fp32
constexpr int lstm_kn(int din) { return din / 16; }
constexpr int lstm_nrp(int din) { return ((lstm_kn(din) > 6 ? lstm_kn(din) : 6) + 3) / 4 * 4; }
constexpr bool lstm_dense_i8(int dout) { return dout >= 64; }
constexpr int lstm_gates_f(int din) { return 20 * 21 * lstm_nrp(din); }
constexpr int lstm_rec_f() { return 88; }
constexpr int lstm_dense_row_f(int dout) { return lstm_dense_i8(dout) ? 8 : 24; }
constexpr int lstm_blob_f(int din, int dout) { return lstm_gates_f(din) + lstm_rec_f() + lstm_dense_row_f(dout) * dout; }
constexpr int DDB_LDS_B = 64 * 1024;       // LDS scratch of a dilated-dense block op (17 920 floats), above the image it completes

}  // namespace fz
}  // namespace nutls
