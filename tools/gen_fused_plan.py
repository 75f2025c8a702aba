#!/usr/bin/env python3
"""Static schedule of the fused frame-step kernel  ->  csrc/fused_plan_lstm.inc  (+ a JSON twin).

The streaming NUNet-TLS-LSTM step (reference: TFL_SIGNITURE.nutls_lstm,
/root/reference/dnn_model/converter_proposed.py:188-867; blocks models/proposed.py:162-282) has a fixed
topology, so everything the kernel would otherwise decode per layer is decided here, once:

  * the op list (154 ops: input layer, 12 MSFE stages, central LSTM; the two phases of an up-sampling
    layer are one op, the output conv rides on the last CTFA);
  * for every conv op its LDS image (the B operand of the MFMAs: rows x channels per time tap, padded
    pitch), which part of that image the producing op forwards from registers, which parts are staged
    from HBM (previous-frame tap, skip-connection channels) and how far ahead their loads are issued;
  * the tiling of every conv op over the 8 waves of a workgroup (32x32x2 tiles with an in-register
    epilogue for the large layers, 16x16x4 tiles + K split + LDS exchange for the small ones);
  * every offset into the per-stream HBM arena (state tensors, parity-strided) and into the weight blob.

Run:  python tools/gen_fused_plan.py            (rewrites the .inc and tests/golden/fused_plan_lstm.json)
      python tools/gen_fused_plan.py --check    (exit 1 if the committed files are stale)
"""
from __future__ import annotations

import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "nested-u-net-based-real-time-speech-enhancement-mobile-app_amd")
def plan_tag(variant, G=1):
    return ("lstm" if variant == "lstm" else "base") + ("" if G == 1 else "_g%d" % G)


OUT_DIR = os.environ.get("NUTLS_PLAN_OUT")      # developer knob (tools/exp/build_plan_lib.sh): write the plans (and their dumps) here, not into the tree


def inc_path(variant, G=1):
    return os.path.join(OUT_DIR or os.path.join(PKG, "csrc"), "fused_plan_%s.inc" % plan_tag(variant, G))


def json_path(variant, G=1):
    return os.path.join(OUT_DIR or os.path.join(ROOT, "tests", "golden"), "fused_plan_%s.json" % plan_tag(variant, G))


# the plans that are built into the library: (variant, streams per workgroup)
PLANS = [("lstm", 1), ("baseline", 1), ("lstm", 2), ("lstm", 4)]
KFIRST = os.environ.get("NUTLS_PLAN_KFIRST", "1") != "0"    # K-split-first tilings of the small layers (developer knob: 0 = the round-4 tilings)
LAZY = os.environ.get("NUTLS_PLAN_LAZY", "1") != "0"        # strided convs' input states written lazily (OpD::d0_on = 2)
EPL1 = os.environ.get("NUTLS_PLAN_EPL1", "1") != "0"        # one output element per lane in the row-wise epilogue of the layers with <= 512 outputs (OpD::epl)

# (prefix, depth, f0, conv state tag, sub-pixel state tag, resample layer)   converter_proposed.py:221-727
ENC = [("msfe6_en", 6, 256, "msfe6_ee", "msfe6_ed", "msfe6_down_sampling"),
       ("msfe5_en", 5, 128, "msfe5_ee", "msfe5_ed", "msfe5_down_sampling"),
       ("msfe4_en", 4, 64, "msfe4_ee", "msfe4_ed", "msfe4_down_sampling"),
       ("msfe4_en2", 4, 32, "msfe4_ee2", "msfe4_ed2", "msfe4_down_sampling2"),
       ("msfe4_en3", 4, 16, "msfe4_ee3", "msfe4_ed3", "msfe4_down_sampling3"),
       ("msfe3_en", 3, 8, "msfe3_ee", "msfe3_ed", "msfe3_down_sampling")]
DEC = [("msfe3_de", 3, 8, "msfe3_de", "msfe3_dd", "msfe3_upsampling"),
       ("msfe4_de", 4, 16, "msfe4_de", "msfe4_dd", "msfe4_upsampling"),
       ("msfe4_de2", 4, 32, "msfe4_de2", "msfe4_dd2", "msfe4_upsampling2"),
       ("msfe4_de3", 4, 64, "msfe4_de3", "msfe4_dd3", "msfe4_upsampling3"),
       ("msfe5_de", 5, 128, "msfe5_de", "msfe5_dd", "msfe5_upsampling"),
       ("msfe6_de", 6, 256, "msfe6_de", "msfe6_dd", "msfe6_upsampling")]

T_INPUT, T_CONV, T_LSTM, T_CTFA, T_DDB = 0, 1, 2, 3, 4
DDB_LDS_B = 64 * 1024        # LDS scratch of a dilated-dense block op (70 KB), above the small image it completes
K_IN, K_EL, K_DL, K_DOWN, K_UP = 0, 1, 2, 3, 4
P_R32, P_R32B, P_X16B = 0, 3, 4
S_PREV, S_CUR, S_SCRATCH = 0, 1, 2
LDS_BYTES = 160 * 1024
SCR_BYTES = 8192
SCR_B = LDS_BYTES - SCR_BYTES
MAX_PARTS, MAX_ZERO, MAX_SEG, CARRY_FRAGS = 6, 12, 6, 12
XCOPY_B = SCR_B + 7168
# packed plans (several streams per workgroup): scratch of an LSTM op that serves several streams (20 x 21 partial float4 + h per stream),
# the fp32 rows it reads (1 KB per stream) -- in the middle of LDS: the images around an LSTM are a few KB, the exchange buffers sit at the top
PK_LSTM_SCR_B, PK_LSTM_SCR_STRIDE, PK_XCOPY_B = 64 * 1024, 7168, 96 * 1024
PK_CTFA_SCR_STRIDE = 8192


def r64(n):
    return (n + 63) // 64 * 64


# --------------------------------------------------------------------------------- arena layout
class Arena:
    """Per-stream HBM arena, mirrored by engine.cpp (allocate_states): all first buffers of the state
    tensors in signature order, then all second buffers in the same order (so `cur` and `prev` of every
    tensor differ by the constant PS), then the scratch tensors."""

    def __init__(self, variant="lstm"):
        self.states = []   # (name, rows, cols)
        for side, stages in ((0, ENC), (1, DEC)):
            for (p, D, f0, ct, st, _rs) in stages:
                for i in range(1, D + 1):
                    c = (128 if side else 64) if i == 1 else (64 if side else 32)
                    self.states.append(("%s_prev%d" % (ct, i), f0 >> (i - 1), c))
                for j in range(1, D + 1):
                    self.states.append(("%s_prev%d" % (st, j), (f0 >> D) << (j - 1), 64))
        self.rings = []    # baseline: in-place history rings of the dilated-dense blocks (single buffer, after the ping-pong block)
        if variant == "lstm":
            for (p, *_r) in ENC:
                self.states += [(p + "_h", 21, 1), (p + "_c", 21, 1)]
            self.states += [("state_h", 21, 1), ("state_c", 21, 1)]
            for (p, *_r) in DEC:
                self.states += [(p + "_h", 21, 1), (p + "_c", 21, 1)]
        else:
            # converter_nunet_tls.py:173-180 / :228-235: prev_in [1,F,C], prev_k [d,F,k*G] (d = 2^(k-1)), prev_out [1,F,G]
            bn = [(st[0], st[2] >> st[1], 32) for st in ENC] + [("", 4, 64)] + [(st[0], st[2] >> st[1], 32) for st in DEC]
            for (p, F, C) in bn:
                tag = (p + "_ddb") if p else "ddb"
                G = C // 2
                self.rings.append((tag + "_prev_in", F, C))
                for k in range(1, 7):
                    self.rings.append(("%s_prev%d" % (tag, k), (1 << (k - 1)) * F, k * G))
                self.rings.append((tag + "_prev_out", F, G))
        self.off = {}
        cur = 0
        for (n, r, c) in self.states:
            self.off[n] = cur
            cur += r64(r * c)
        self.PS = cur
        cur *= 2
        for (n, r, c) in self.rings:
            self.off[n] = cur
            cur += r64(r * c)
        self.scratch = {}
        for n, sz in [("t_inlayer", 256 * 64), ("t_y", 256 * 64), ("t_d", 256 * 64), ("t_up", 256 * 128)] + \
                     [("upcat%d" % s, (DEC[s][2] // 2) * 128) for s in range(6)]:
            self.scratch[n] = cur
            cur += r64(sz)
        # carried partial sums of the two-tap convs (P x N fp32 each): two blocks, one read (last frame's) and one written per frame
        self.ys_block = 0
        for stages in (ENC, DEC):
            for (p, D, f0, ct, st, _rs) in stages:
                for i in range(1, D + 1):
                    self.ys_block += r64((f0 >> i) * 32)
                for j in range(1, D + 1):
                    Pj, Nj = (f0 >> D) << (j - 1), (128 if j == D else 64)
                    if carries_sums(K_DL, Nj, Pj):
                        self.ys_block += r64(Pj * Nj)
        self.scratch["ysum"] = cur
        cur += 2 * self.ys_block
        self.floats = cur


# --------------------------------------------------------------------------------- weight blob
class Blob:
    def __init__(self):
        self.cur = 0
        self.items = []    # (offset, floats, what, layer key)

    def add(self, n, what, key):
        off = self.cur
        self.items.append((off, n, what, key))
        self.cur += r64(n)
        return off


def lstm_blob_floats(din, dout):
    """Blob floats of an LSTM + Dense op: int8 gate kernels [20 slices][21 units][NRP dwords] | record 88 | Dense rows (8 dwords each where
    the container holds them int8 -- dout >= 64 -- else 24 fp32): mirrors fused_plan.hpp lstm_blob_f."""
    nrp = (max(din // 16, 6) + 3) // 4 * 4
    return 20 * 21 * nrp + 88 + (8 if dout >= 64 else 24) * dout


def conv_geom(kind, side, i_or_j, D):
    """(cin, N, taps, kf, stride, ln, R, gc) of a conv op (models/proposed.py:198-265)."""
    if kind == K_IN:
        return (128 if side else 64, 64, 1, 1, 1, 1, 1, 64)
    if kind == K_EL:
        cin = (128 if side else 64) if i_or_j == 1 else (64 if side else 32)
        return (cin, 32, 2, 3, 2, 1, 1, 32)
    if kind == K_DL:
        return (64, 128, 2, 3, 1, 1, 2, 64) if i_or_j == D else (64, 64, 2, 3, 1, 1, 2, 32)
    if kind == K_DOWN:
        return (64, 64, 1, 3, 2, 0, 1, 64)
    if kind == K_UP:
        return (128, 128, 1, 2, 1, 0, 2, 128)
    raise ValueError(kind)


R32_TABLE = {   # (kind, N, P) -> (PT, NT, PG, CG)
    (K_IN, 64, 256): (1, 2, 8, 1), (K_IN, 64, 128): (1, 2, 4, 1),
    (K_EL, 32, 128): (1, 1, 4, 1),
    # (the 128-channel sub-pixel conv at 128 positions: EIGHT wave tasks of one position tile x one LayerNorm group (two channel tiles) -- with
    #  (2, 2, 2, 2), four tasks of twice the size, every SIMD had one wave to hide its own B reads and weight widening behind its own MFMAs, and
    #  four waves sat out the epilogue: 0.2934 -> 0.2890 ms per step, profiles/r06_dev_log.txt "dl128".  The same layers at 64 positions stay on
    #  16x16 tiles: on 32x32 tiles -- four tasks -- they are 0.2 % slower)
    (K_DL, 128, 128): (1, 2, 4, 2),
    (K_DOWN, 64, 128): (1, 1, 4, 2), (K_DOWN, 64, 64): (1, 1, 2, 2),
    # (msfe6_upsampling: one position tile x two channel tiles per wave task -- half the B reads of (2, 1, 2, 4), LDS time 6 144 -> 3 072 cycles
    #  against 6 144 of MFMAs: -1.0 % of the step with "sp8"; tilings with FOUR wave tasks for the 64-position up-sampling / the 128-position
    #  down-sampling conv lose 0.7 %, profiles/r06_dev_log.txt)
    (K_UP, 128, 128): (1, 2, 4, 2), (K_UP, 128, 64): (1, 1, 2, 4), (K_UP, 128, 32): (1, 1, 1, 4),
}
# packed plans only: virtual position counts (streams side by side) no single stream has
R32_TABLE_PACKED = {
    (K_DL, 64, 128): (1, 1, 4, 2), (K_EL, 32, 256): (2, 1, 4, 1),
}


def carries_sums(kind, N, P):
    """Which two-tap convs hand their previous-frame tap over as partial sums (OpD.ys): the strided convs -- P x 32 sums replace
    2P x cin input rows, 2-8x fewer bytes, and the image of the previous frame (with its staging: loads, splits, LDS stores) goes.
    Not the sub-pixel convs: their sums are as many bytes as their input rows (N = 64) or twice as many (N = 128), and the
    loads / stores of those land in the small ops around them and in front of the CTFA drain points -- measured with all of them
    carried: +18 us over the 46 small ones, +7 us in the CTFAs; with only the six on 32x32 tiles: -5 us in those ops, +13 us in the
    ops that request their sums."""
    return 1 if kind == K_EL else 0


def tiling(kind, N, P, cin, taps, rounds=1, gs=1, waves=8, fits=None, ys=0):
    """-> dict(path, PT, NT, PG, CG, KSt, KSg).  R32B: 32x32x16 bf16 tiles, PT x NT tiles per wave, PG x CG wave tasks, whole
    LayerNorm groups per wave.  X16B: 16x16x32 bf16 tiles, wave task = (position group pg, channel tile ct, K slice
    (time tap ks_t, channel-group range ks_g)), PT tiles per wave; the K slices meet in the LDS exchange buffer.
    Packed plans: `gs` streams side by side -- the tiling is that of gs * P positions (32x32 tiles: whole tiles per stream).
    None: no tiling for that many positions (the layer then runs one stream at a time)."""
    per_stream = P
    P = gs * P
    table = {**R32_TABLE, **R32_TABLE_PACKED} if gs > 1 else R32_TABLE
    # (32x32 tiles may hold several streams' positions -- except those of the two-tap convs, whose carried sums are laid out by whole tiles)
    if (kind, N, P) in table and (gs == 1 or per_stream % 32 == 0 or kind != K_EL):
        PT, NT, PG, CG = table[(kind, N, P)]
        return dict(path=P_R32B, PT=PT, NT=NT, PG=PG, CG=CG, KSt=1, KSg=1)
    if gs > 1 and P > 64:
        return None
    assert P <= 64, (kind, N, P)
    CT = N // 16
    ptiles = (P + 15) // 16
    if KFIRST and gs == 1 and kind != K_UP and rounds == 1 and ptiles == 1 and CT <= 4:
        # K split FIRST (one-stream plans): the waves of a position group that differ only in their channel tile read the SAME B
        # fragments -- eight waves x the whole K range of a 128-channel sub-pixel conv are 8 x 36 ds_read_b128 per position tile, and
        # the B reads are 5 % of the step (profiles/r05_knob_experiments.txt, `nobread`).  So: as many K slices as divide K (time tap,
        # then channel-group halves / quarters of every segment), every wave task NT channel tiles of its slice -- a B fragment then
        # serves NT MFMAs per plane.  Bounds: PT x NT <= 4 accumulator tiles per plane, the exchange buffer (one slice per K slice)
        # next to the image in LDS.  Among the tilings with eight wave tasks: least LDS + memory-pipe time, then fewest slices.
        # Only the layers of one position tile and at most four channel tiles: measured per tiling class (profiles/r05_kfirst_classes.txt),
        # the 64-channel sub-pixel convs gain 0.1-0.4 us each and the 128-input-channel layers 0.2-0.4; with more position tiles the
        # search splits the positions over the waves, every position group pulls the weights through the memory pipe again and the
        # layers LOSE 0.3-1.6 us each; the 128-channel sub-pixel convs (four tiles per task) do not move.
        nseg = taps * (3 if kind in (K_EL, K_DL, K_DOWN) else 1)
        best = None
        for kst in ((1, 2) if taps == 2 else (1,)):
            for ksg in (1, 2, 4):
                if (cin // 32) % ksg:
                    continue
                for cg in (1, 2, 4, 8):
                    if CT % cg:
                        continue
                    for pg in (1, 2, 4):
                        if ptiles % pg or kst * ksg * cg * pg != 8:
                            continue
                        pt, nt = ptiles // pg, CT // cg
                        both = 2 if (ys and kst == 1) else 1      # a two-tap conv whose waves own both taps: two accumulator sets, two slices per K slice
                        if pt * nt * both > 4 or (fits is not None and not fits(kst * ksg * both)):
                            continue
                        # cost (cycles of the two shared pipes): B reads -- 8 waves x K steps x PT x 3 planes x 8 cycles of LDS per
                        # ds_read_b128 -- and the weights, which every position group pulls through the CU's memory pipe again
                        reads = (nseg // kst) * (cin // 32 // ksg) * pt
                        cost = 8 * 3 * 8 * reads + N * nseg * cin * pg // 64
                        key = (cost, kst * ksg, -cg)
                        if best is None or key < best[0]:
                            best = (key, dict(path=P_X16B, PT=pt, NT=nt, PG=pg, CG=cg, KSt=kst, KSg=ksg))
        if best is not None:
            return best[1]
    # (one-stream instances only: on the side-by-side ops of the two-stream plan the same tilings are +0.6 % at 1 024 streams and -2.2 % at 512)
    if gs == 1 and kind == K_DL and CT == 8 and ptiles % 2 == 0 and rounds == 1:
        # the 128-channel sub-pixel convs at 32 / 64 positions (one-stream plans): two position groups x four channel groups of TWO tiles -- a wave
        # reads half the B fragments (with eight channel groups every wave read all of them: 1 152 ds_read_b128 per op at 64 positions, 3.8 us of
        # LDS time under 1 us of MFMAs) for twice the weight widening: 0.2866 -> 0.2845 ms per step (profiles/r06_dev_log.txt "nt2")
        return dict(path=P_X16B, PT=ptiles // 2, NT=2, PG=2, CG=4, KSt=1, KSg=1)
    if gs == 1 and kind == K_DL and CT == 4 and ptiles % 2 == 0 and rounds == 1 and taps == 2:
        # the same for the 64-channel sub-pixel convs at 32 / 64 positions (six ops): -1.2 % ("nt2b")
        return dict(path=P_X16B, PT=ptiles // 2, NT=2, PG=2, CG=2, KSt=2, KSg=1)
    if gs == 1 and kind == K_EL and CT == 2 and ptiles % 2 == 0 and rounds == 1 and taps == 2:
        # ... and for the strided convs at 32 / 64 positions (eight ops): both channel tiles on one wave, the positions dealt to two or four
        # groups instead: -1.0 % ("nt2c")
        ksg = 2 if (cin // 32) % 2 == 0 else 1
        pg = 8 // (2 * ksg)
        if ptiles % pg == 0:
            return dict(path=P_X16B, PT=ptiles // pg, NT=2, PG=pg, CG=1, KSt=2, KSg=ksg)
    # (the in-convs and down-sampling convs at 32 / 64 positions the same way: -0.25 %, inside the noise of a box -- left as they are)
    if gs == 1 and kind == K_DL and CT == 8 and ptiles == 1 and rounds == 1 and taps == 2:
        # the 128-channel sub-pixel convs of ONE position tile (six ops; eight channel tiles, so the K-first search above does not see them): two tiles
        # per wave task and the time taps dealt to two waves instead of eight tasks of one tile over the whole K range: -0.4 % ("sp8")
        return dict(path=P_X16B, PT=1, NT=2, PG=1, CG=4, KSt=2, KSg=1)
    rem = max(1, waves // CT)
    KSt = 1 if (rounds == 2 or kind == K_UP) else min(taps, rem)
    rem //= KSt
    KSg = 1
    while kind != K_UP and KSg * 2 <= rem and (cin // 32) % (KSg * 2) == 0:
        KSg *= 2
    rem //= KSg
    PG = 1
    while PG * 2 <= rem and ptiles % (PG * 2) == 0:
        PG *= 2
    return dict(path=P_X16B, PT=ptiles // PG, NT=1, PG=PG, CG=CT, KSt=KSt, KSg=KSg)


def make_img(kind, P, cin, rounds=1, fmt=1, csplit=1, cps=0):
    """LDS image geometry of a conv op: dict(fmt, plane_b, taps, tap_b, pitch_b, pair, half_b, row0, bytes, zero[], seg_b[], seg_tk[], seg_c0[]).
    fmt 1: every row holds three bf16 planes (hi | mid | lo of the fp32 activations, x = hi + mid + lo exactly) of `rowch` channels
    each, fmt 0: fp32 rows.  csplit 2 (1x1 layers only): the image holds one half of the input channels at a time (two rounds)."""
    esz = 2 if fmt else 4
    npl = 3 if fmt else 1
    cimg = cin // csplit

    def halo(byte0, nch):
        # nch channels starting at byte0 (plane 0): one block if the planes are adjacent, else one per plane
        return [(byte0 + p * plane_b, nch * esz // 16) for p in range(npl)] if plane_b != nch * esz else [(byte0, npl * nch * esz // 16)]

    if kind == K_IN:
        plane_b = cimg * esz
        pitch = npl * plane_b + 16
        g = dict(taps=1, pitch_b=pitch, pair=0, half_b=0, row0=0, one=P * pitch, zero=[])
        segs = [(0, 0, 0, c * cimg) for c in range(csplit)]
    elif kind == K_EL:
        plane_b = 2 * cin * esz
        pitch = npl * plane_b + 16
        half = cin * esz
        g = dict(taps=2, pitch_b=pitch, pair=1, half_b=half, row0=1, one=(P + 1) * pitch,
                 zero=halo(0, cin) + halo(P * pitch + half, cin))
        segs = [(t, k, (k >> 1) * pitch + (k & 1) * half, 0) for t in (0, 1) for k in (0, 1, 2)]
    elif kind == K_DL:
        plane_b = cin * esz
        pitch = npl * plane_b + 16
        g = dict(taps=2, pitch_b=pitch, pair=0, half_b=0, row0=1, one=(P + 2) * pitch,
                 zero=halo(0, cin) + halo((P + 1) * pitch, cin))
        segs = [(t, k, k * pitch, 0) for t in (0, 1) for k in (0, 1, 2)]
    elif kind == K_DOWN:
        plane_b = 2 * cin * esz
        pitch = npl * plane_b + 16
        half = cin * esz
        g = dict(taps=1, pitch_b=pitch, pair=1, half_b=half, row0=0, one=(P + 1) * pitch, zero=halo(P * pitch, cin))
        segs = [(0, k, (k >> 1) * pitch + (k & 1) * half, 0) for k in (0, 1, 2)]
    elif kind == K_UP:
        # Conv2DTranspose (1,3) stride 2 (proposed.py:260-265): out[2i] = W0 x[i] + W2 x[i-1], out[2i+1] = W1 x[i]
        plane_b = cin * esz
        pitch = npl * plane_b + 16
        g = dict(taps=1, pitch_b=pitch, pair=0, half_b=0, row0=1, one=(P + 1) * pitch, zero=halo(0, cin))
        segs = [(0, 2, 0, 0), (0, 0, pitch, 0), (0, 1, pitch, 0)]      # (t, kw, byte offset, first channel): two even segments, one odd
    else:
        raise ValueError(kind)
    g["fmt"] = fmt
    g["plane_b"] = plane_b if fmt else 0
    if cps:
        # Carried partial sums: y_t = W_cur x_t + W_prev x_{t-1}, and the second term is computed one frame EARLIER, by the op
        # that has x_{t-1} in LDS as its current input (the same B fragments, the other time tap's weights), and handed over as
        # P x N fp32 sums.  So the image holds the current frame only; the six K segments keep their (time tap, frequency tap)
        # tags -- they select the weights -- and the two taps of a frequency tap read the same rows.
        g["taps"] = 1
    one = (g.pop("one") + 255) // 256 * 256
    if csplit == 2:
        assert kind == K_IN and rounds == 2
        g["tap_b"] = 0
        g["bytes"] = one
        g["seg_b"] = [s[2] for s in segs]
    elif g["taps"] == 2 and rounds == 2:
        g["tap_b"] = 0
        g["bytes"] = one
        segs = [s for s in segs if s[0] == 1] + [s for s in segs if s[0] == 0]     # current-frame tap first
        g["seg_b"] = [s[2] for s in segs]
    else:
        g["tap_b"] = one if g["taps"] == 2 else 0
        g["bytes"] = one * g["taps"]
        g["seg_b"] = [s[0] * g["tap_b"] + s[2] for s in segs]
        zs = []
        for t in range(g["taps"]):
            zs += [(t * g["tap_b"] + z[0], z[1]) for z in g["zero"]]
        g["zero"] = zs
    g["seg_tk"] = [s[0] * 4 + s[1] for s in segs]
    g["seg_c0"] = [s[3] for s in segs]
    return g


def region_overlap(a, b):
    """Do two row-strided regions of floats (off, rows, ld, width) share an element?"""
    (ao, ar, al, aw), (bo, br, bl, bw) = a, b
    if ao + (ar - 1) * al + aw <= bo or bo + (br - 1) * bl + bw <= ao:
        return False
    if al != bl:
        return True          # (different pitches: the coarse answer)
    d = bo - ao              # b's first element relative to a's, in a's row grid
    r, c = d // al, d % al   # (floor: c in [0, ld))
    cols = c < aw or c + bw > al          # b's columns [c, c + bw) meet a's [0, aw) -- directly or wrapped into the next row
    rows = r < ar and r + br > 0 if c < aw else r + 1 < ar and r + 1 + br > 0
    return cols and rows


def ddb_flops(F, c):
    """Dilated-dense block (nunet_tls.py:277-359) on F positions of c channels: (2,3) conv c -> c/2, six times
    [grouped dilated (2,3) conv over k channels per filter + 1x1 conv], (2,3) conv c/2 -> c."""
    g = c // 2
    return 2 * F * (6 * c * g + sum(6 * k * g + g * g for k in range(1, 7)) + 6 * g * c)


def build(variant="lstm", G=1):
    """The plan of `variant` for workgroups of G streams.  G = 1: one stream per workgroup (the plan every handle can run).
    G > 1 ("packed"): the layers whose images fit LDS G times run the G streams SIDE BY SIDE (one virtual position axis of G * P:
    the weights are fetched and converted once, the small layers' mostly empty 16-position tiles fill up), the others run once per
    stream, one after the other -- `layers` below is the one-stream op list, `ops` the list of op INSTANCES in execution order."""
    if G == 1:
        return build_for(variant, 1, {})
    assert variant == "lstm", "packed plans exist for the LSTM variant"
    # classification: which layers run how many streams side by side -- decided on the layer list of the one-stream plan, by LDS fit
    _A, _W, layers = build_for(variant, 1, {})
    cap = {}
    for _attempt in range(64):
        cls = classify(layers, G, cap)
        try:
            return build_for(variant, G, cls)
        except NextImageDoesNotFit as e:
            # the op could not complete the image set of the op after it (its exchange buffer / the CTFA's scratch is in the way): the
            # one with the larger group runs a smaller one
            victim = e.target if cls[e.target] >= cls[e.builder] else e.builder
            assert cls[victim] > 1, str(e)
            cap[victim] = cls[victim] // 2
    raise AssertionError("no packed plan found")


def sbs_fit(o, G, nxt_bytes=0):
    """Can conv layer `o` (a record of the one-stream plan) run G streams side by side?  -> (tiling, image) or None."""
    kind, N, P, cin, taps = o["kind"], o["N"], o["P"], o["cin"], o["taps"]
    t = tiling(kind, N, P, cin, taps, 1, gs=G)
    if t is None:
        return None
    ntot = N * (2 if kind == K_UP else 1)
    lim = SCR_B
    if t["path"] == P_X16B:
        ks = t["KSt"] * t["KSg"] * (2 if (o["ys"] and t["KSt"] == 1) else 1)
        lim = (SCR_B - ks * G * P * (ntot + 4) * 4) // 256 * 256
    if t["path"] == P_R32B and o["ys"] and not (t["PG"] * t["CG"] == 4 and t["CG"] == 1):
        return None
    for fmt in (1, 0):
        g = make_img(kind, P, cin, 1, fmt=fmt, cps=o["ys"])
        if G * g["bytes"] <= lim and (fmt == 1 or t["path"] == P_R32B):
            return t, g, lim
    return None


def group_sizes(G):
    out, g = [], G
    while g >= 1:
        out.append(g)
        g //= 2
    return out


def classify(layers, G, cap=None):
    """layer name -> how many of the workgroup's G streams one instance of the layer runs side by side: G, G / 2, ... or 1 -- the
    largest group whose images (and exchange buffer) fit LDS, made equal along the links that have no HBM copy of the rows.
    `cap`: upper bounds per layer (build() lowers them where an op could not complete the image set of the op after it)."""
    cap = cap or {}
    cls = {}
    convs = [o for o in layers if o["type"] == T_CONV]
    for o in convs:
        cls[o["name"]] = next(g for g in group_sizes(G) if g <= cap.get(o["name"], G) and (g == 1 or sbs_fit(o, g)))
    changed = True
    while changed:
        changed = False
        # links without an HBM copy of the rows: both ends run the same streams.  (decoder) sub-pixel conv D -> CTFA -> down / up-sampling;
        # up-sampling -> in-conv of the decoder stage; conv D -> LSTM -> sub-pixel conv 1
        for i, o in enumerate(layers):
            if o["type"] == T_CTFA:
                grp = [layers[i - 1]] + ([layers[i + 1]] if i + 1 < len(layers) else [])
            elif o["type"] == T_CONV and o["kind"] == K_UP:
                grp = [o, layers[i + 1]]
            elif o["type"] in (T_LSTM, T_DDB):
                grp = [layers[i - 1], layers[i + 1]]
            else:
                continue
            m = min(cls[x["name"]] for x in grp)
            for x in grp:
                if cls[x["name"]] != m:
                    cls[x["name"]] = m
                    changed = True
    for i, o in enumerate(layers):
        if o["type"] == T_CTFA:
            cls[o["name"]] = cls[layers[i - 1]["name"]]
        elif o["type"] == T_LSTM:
            cls[o["name"]] = cls[layers[i + 1]["name"]]
            assert cls[layers[i - 1]["name"]] == cls[o["name"]] == G, ("an LSTM between layers that do not run all streams side by side", o["name"])
        elif o["type"] == T_INPUT:
            cls[o["name"]] = 1
            assert cls[layers[i + 1]["name"]] == 1
    return cls


class NextImageDoesNotFit(Exception):
    def __init__(self, builder, target):
        Exception.__init__(self, "%s cannot complete the image set of %s" % (builder, target))
        self.builder, self.target = builder, target


def build_for(variant, G, cls):
    A = Arena(variant)
    W = Blob()
    YS = Blob()         # carried partial sums of the two-tap convs: offsets inside one block (the arena holds two: read / write parity)
    ops = []
    base = variant != "lstm"

    def gs_of(name):
        return cls.get(name, 1)

    def new_op(**kw):
        d = dict(type=T_CONV, name="", kind=0, P=0, cin=0, N=0, taps=0, kf=0, stride=0, path=0, PT=1, NT=1, PG=1, CG=1, KSt=1, KSg=1,
                 ln=0, R=1, gc=0, rounds=1, nseg=0, seg_b=[], seg_tk=[], ex_b=0, w_off=0, p_off=0,
                 d0=None, d1=None, row_mul=1, row_add=0, fwd=None, img=None, nxt=-1, parts=[],
                 din=0, dout=0, x_b=0, x_pitch_b=0, x_cols=0, y_b=0, h_off=0, c_off=0, ldst=None, lw_off=0,
                 F=0, e0_off=0, e0_ld=0, last=0, cw_off=0, drain=0, bidx=0, wkey="", flops=0, x_fmt=0, x_plane_b=0, ys=0, ys_off=0, xs_off=0, xs_ld=0,
                 gs=1, g0=0, scr_b=SCR_B, scr_gstride_b=0, xcopy_b=XCOPY_B, x_gstride_b=0, layer=-1, epl=4, nt0=0, nt1=0)
        d.update(kw)
        d["gs"] = gs_of(d["name"])
        ops.append(d)
        return d

    def conv_op(name, wkey, kind, side, idx, D, P, d0=None, d1=None, row_mul=1, row_add=0):
        cin, N, taps, kf, stride, ln, R, gc = conv_geom(kind, side, idx, D)
        # both taps of these images do not fit LDS: one tap at a time (the previous-frame tap is staged in mid-op)
        ys = carries_sums(kind, N, P)
        rounds = 1          # (the strided convs' images hold one time tap: nothing needs two rounds any more)
        o = new_op(type=T_CONV, name=name, wkey=wkey, kind=kind, P=P, cin=cin, N=N, taps=taps, kf=kf, stride=stride, ln=ln, R=R, gc=gc,
                   rounds=rounds, d0=d0, d1=d1, row_mul=row_mul, row_add=row_add, ys=ys)
        gs = o["gs"]
        VP = gs * P         # positions of the op's streams side by side
        ntot_ = N * (2 if kind == K_UP else 1)
        img_b = make_img(kind, P, cin, rounds, fmt=1, cps=ys)["bytes"]

        def fits(ks):      # image + exchange buffer of ks slices below the scratch
            return img_b <= (SCR_B - ks * P * (ntot_ + 4) * 4) // 256 * 256
        o.update(tiling(kind, N, P, cin, taps, rounds, gs=gs, fits=fits, ys=ys))
        assert not (ys and o["path"] == P_X16B and o["KSt"] == 1 and o["KSg"] > 1), name      # (both taps per wave: one K slice)
        ntot = N * (2 if kind == K_UP else 1)
        x16 = o["path"] == P_X16B
        if x16:
            # exchange slices: K slice (time tap ks_t, channel range ks_g); a two-tap conv whose waves own both taps (KSt 1)
            # still has two slices: 0 = next frame's partial sums, 1 = this frame's
            ks = o["KSt"] * o["KSg"] * (2 if (ys and o["KSt"] == 1) else 1)
            ex = ks * VP * (ntot + 4) * 4
            o["ex_b"] = (SCR_B - ex) // 256 * 256
        if ys:
            o["ys_off"] = YS.add(P * ntot, "ysum", wkey)
        lim = o["ex_b"] if x16 else SCR_B
        # the image as three bf16 planes where that fits LDS; the two largest stay fp32 and are split when they are read
        g = make_img(kind, P, cin, rounds, fmt=1, cps=ys)
        if gs * g["bytes"] > lim:
            assert not x16, name
            g = make_img(kind, P, cin, rounds, fmt=0, cps=ys)
        g["gstride_b"] = g["bytes"] if gs > 1 else 0
        o["img"] = g
        o["nseg"] = len(g["seg_b"])
        o["seg_b"] = g["seg_b"]
        o["seg_tk"] = g["seg_tk"]
        K = (3 * cin) if kind == K_UP else taps * kf * cin
        o["flops"] = 2 * P * (K * N if kind != K_UP else 3 * cin * N)
        # int8 weights in bf16-MFMA fragment order (a fragment = the 8 K values of a lane): per wave task a run of
        # "super-fragments" (2 fragments = 16 bytes = one dwordx4 per lane)
        segw = 3 if kind == K_UP else len(g["seg_b"]) // o["KSt"]
        if x16:
            nf = segw * (cin // 32 // o["KSg"]) * o["NT"]
            wtasks = o["CG"] * o["KSt"] * o["KSg"]
        elif ys:
            # 32x32 tiles of a two-tap conv: waves 0..3 own the next frame's sums (time tap 0), waves 4..7 this frame's (tap 1)
            assert o["PG"] * o["CG"] == 4, name
            nf = 3 * (cin // 16) * o["NT"]
            wtasks = 2 * o["CG"]
        else:
            nf = segw * (cin // 16) * o["NT"]
            wtasks = o["CG"]
        if x16:
            # (fused_plan.hpp OpD::epl: one output element per lane in the row-wise epilogue where the layer has at most one element per thread)
            o["epl"] = 1 if (EPL1 and VP * ntot <= 512 and gc in (32, 64) and kind != K_UP) else 4
        o["w_off"] = W.add(wtasks * ((nf + 1) // 2) * 256, "conv_w", wkey)
        o["p_off"] = W.add(2 * ntot + 2 * gc + 1, "conv_p", wkey)      # bias | per-channel weight scale | gamma | beta | alpha
        return o

    # ---- layer list (= the op list of the one-stream plan) ------------------------------------------
    inp = new_op(type=T_INPUT, name="input_layer")
    inp["p_off"] = W.add(64 * 4 + 1, "input_p", "input_layer")

    def st_off(tag, i):
        return A.off["%s_prev%d" % (tag, i)]

    def stage(side, s):
        p, D, f0, ct, stg, rs = (DEC if side else ENC)[s]
        fd = f0 >> D
        c1 = 128 if side else 64
        pair = DEC[5 - s] if not side else None
        lst = []
        lst.append(conv_op(p + "_in", p + "_in", K_IN, side, 0, D, f0, d0=(S_CUR, st_off(ct, 1), c1)))
        for i in range(1, D + 1):
            if i < D:
                ci1 = 64 if side else 32
                d0 = (S_CUR, st_off(ct, i + 1), ci1)
                d1 = (S_CUR, st_off(stg, D - i + 1) + 32, 64)
            else:
                d0 = (S_CUR, st_off(stg, 1) + 32, 64)
                d1 = None
            lst.append(conv_op("%s_conv%d" % (p, i), "%s_conv%d" % (p, i), K_EL, side, i, D, f0 >> i, d0=d0, d1=d1))
        if base:
            # dilated-dense block (nunet_tls.py:277-359): reads e_D from the state tensor in HBM (the strided conv before it
            # drains its stores), keeps its history rings in HBM, writes d_0 to HBM and into the next image
            lst[-1]["drain"] = 1
            l = new_op(type=T_DDB, name=p + "_ddb", wkey=p, din=fd * 32, dout=fd * 32, x_cols=32, bidx=(7 + s if side else s), drain=1,
                       flops=ddb_flops(fd, 32))
        else:
            l = new_op(type=T_LSTM, name=p + "_lstm", wkey=p, din=fd * 32, dout=fd * 32, x_cols=32,
                       h_off=A.off[p + "_h"], c_off=A.off[p + "_c"], ldst=(S_CUR, st_off(stg, 1), 64), drain=1)
            l["lw_off"] = W.add(lstm_blob_floats(l["din"], l["dout"]), "lstm", p)
        lst.append(l)
        for j in range(1, D + 1):
            P = fd << (j - 1)
            if j < D:
                d0 = (S_CUR, st_off(stg, j + 1), 64)
                d1 = (S_CUR, st_off(pair[3], D - j + 1) + 32, 64) if pair else None
            else:
                d0 = (S_CUR, st_off(pair[3], 1) + 64, 128) if pair else None
                d1 = None
            lst.append(conv_op("%s_spconv%d" % (p, j), "%s_spconv%d" % (p, j), K_DL, side, j, D, P, d0=d0, d1=d1, row_mul=2))
        c = new_op(type=T_CTFA, name=p + "_ctfa", wkey=p, F=f0, e0_off=st_off(ct, 1), e0_ld=c1, drain=0, bidx=(6 + s if side else s))      # bidx: which of the 12 CTFAs (history ring of the causal32 mode)
        c["cw_off"] = W.add(2 * (64 * 16 + 16 + 64 * 16 + 64) + 65, "ctfa", p)
        lst.append(c)
        return lst

    stage_ops = {}
    downs = {}
    for s in range(6):
        stage_ops[(0, s)] = stage(0, s)
        p, D, f0, ct, stg, rs = ENC[s]
        downs[s] = conv_op(rs, rs, K_DOWN, 0, 0, D, f0 // 2, d0=(S_SCRATCH, A.scratch["upcat%d" % (5 - s)] + 64, 128))
    if base:
        downs[5]["drain"] = 1
        cl = new_op(type=T_DDB, name="ddb", wkey="", din=256, dout=256, x_cols=64, bidx=6, drain=1, flops=ddb_flops(4, 64))
    else:
        cl = new_op(type=T_LSTM, name="lstm", wkey="", din=256, dout=256, x_cols=64, h_off=A.off["state_h"], c_off=A.off["state_c"], ldst=None, drain=1)
        cl["lw_off"] = W.add(lstm_blob_floats(256, 256), "lstm", "")
    ups = {}
    for s in range(6):
        p, D, f0, ct, stg, rs = DEC[s]
        ups[s] = conv_op(rs, rs, K_UP, 1, 0, D, f0 // 2, row_mul=2)
        ups[s]["bidx"] = s          # (up-sampling convs: which decoder stage -- the slot of its output in the activation trace of the profiling build)
        stage_ops[(1, s)] = stage(1, s)
    layers = ops
    for i, o in enumerate(layers):
        o["layer"] = i
    assert YS.cur == A.ys_block, (YS.cur, A.ys_block)

    # ---- staged parts of every layer's image (per stream; the instance list below says for which streams) ------
    def part(src, off, ld, rows, c4s, lds_b, row0, la, round2=0):
        return dict(src=src, off=off, ld=ld, rows=rows, c4s=c4s, lds_b=lds_b, row0=row0, la=la, round2=round2, g0=0, ng=1, gstride_b=0)

    def esz(g):
        return 2 if g["fmt"] else 4

    def la_of(rows, c4s, ng=1):
        return 2 if (ng * rows * c4s + 511) // 512 <= 2 else 1

    for side in (0, 1):
        for s in range(6):
            p, D, f0, ct, stg, rs = (DEC if side else ENC)[s]
            lst = stage_ops[(side, s)]
            for i in range(1, D + 1):
                o = lst[i]
                g = o["img"]
                rows, cin = f0 >> (i - 1), o["cin"]
                parts = []
                o["xs_off"], o["xs_ld"] = st_off(ct, i), cin         # the conv's input state tensor [rows][cin] (nutls_state_set -> partial sums)
                if side:
                    sk = 64 if i == 1 else 32
                    src_off, src_ld = st_off(ct, i) + sk, cin
                    if G == 1 and LAZY and i >= 2:
                        # The skip rows of decoder conv i >= 2 are the paired encoder stage's d_{D-i+1}, and its sub-pixel conv D-i+1 writes them
                        # to TWO state tensors: the input state of its own next sub-pixel conv (channels 0..31 of `<ed>_prev{D-i+2}`) and the
                        # skip half of this conv's input state (converter_proposed.py:467-473).  The kernel reads the FIRST copy here too --
                        # same rows, same 32 channels at ld 64 -- so that the second is written only when somebody outside the kernel looks
                        # (OpD::d1_on = 2, marked below; engine.cpp states_materialize): 63 KB per frame and stream less to write.
                        enc_stg = ENC[5 - s][4]
                        src_off, src_ld = st_off(enc_stg, D - i + 2), 64
                    parts.append(part(S_CUR, src_off, src_ld, rows, sk // 4, g["tap_b"] + sk * esz(g), g["row0"], la_of(rows, sk // 4, o["gs"])))
                o["parts"] = parts
            for j in range(1, D + 1):
                o = lst[D + 1 + j]
                g = o["img"]
                rows = o["P"]
                parts = []
                o["xs_off"], o["xs_ld"] = st_off(stg, j), 64
                if not o["ys"]:
                    parts.append(part(S_PREV, st_off(stg, j), 64, rows, 16, 0, g["row0"], la_of(rows, 16, o["gs"])))
                if j >= 2:
                    # e_{D-j+1}, written this frame by strided conv D-j+1: visible after the LSTM's drain point,
                    # i.e. its loads may be issued by sub-pixel conv 1 at the earliest
                    la = 1 if j == 2 else la_of(rows, 8, o["gs"])
                    parts.append(part(S_CUR, st_off(stg, j) + 32, 64, rows, 8, g["tap_b"] + 32 * esz(g), g["row0"], la))
                o["parts"] = parts
            if side and s >= 1:
                o = ups[s]
                g = o["img"]
                rows = o["P"]
                o["parts"] = [part(S_SCRATCH, A.scratch["upcat%d" % s] + 64, 128, rows, 16, 64 * esz(g), g["row0"], la_of(rows, 16, o["gs"]))]

    # ---- where a layer's rows land in the image of the conv layer after it (geometry; per stream) -----------------
    conv_l = [o["layer"] for o in layers if o["type"] == T_CONV]
    nxt_layer = {a: b for a, b in zip([0] + conv_l[:-1], conv_l)}          # input / conv layer -> the conv layer whose image it completes
    prv_layer = {b: a for a, b in nxt_layer.items()}

    def fwd_geom(o, tgt, coff):
        g = tgt["img"]
        cur_tap = g["taps"] - 1
        e = 2 if g["fmt"] else 4
        return dict(on=1, base_b=cur_tap * g["tap_b"] + coff * e, pitch_b=g["pitch_b"], pair=g["pair"], half_b=g["half_b"], row0=g["row0"],
                    fmt=g["fmt"], plane_b=g["plane_b"], gstride_b=0, mask=1)

    for o in layers:
        if o["type"] not in (T_CONV, T_INPUT) or o["layer"] not in nxt_layer:
            continue
        tgt = layers[nxt_layer[o["layer"]]]
        coff = 0
        if o["type"] == T_CONV and o["kind"] == K_EL and tgt["kind"] == K_DL:
            coff = 32          # e_D -> channels [32,64) of sub-pixel conv 1's input; the LSTM writes [0,32)
        if o["type"] == T_CONV and o["kind"] == K_DOWN and tgt["kind"] == K_UP:
            coff = 64          # z -> channels [64,128) of the first up-sampling input; the central LSTM writes [0,64)
        o["fwd"] = fwd_geom(o, tgt, coff)
    # the last sub-pixel conv of the network feeds the last CTFA (+ output conv): plain [256][64+4] rows at LDS 0
    layers[conv_l[-1]]["fwd"] = dict(on=1, base_b=0, pitch_b=68 * 4, pair=0, half_b=0, row0=0, fmt=0, plane_b=0, gstride_b=0, mask=1)

    # ---- op instances in execution order ------------------------------------------------------------------------
    # a maximal run of consecutive one-stream-at-a-time layers is walked once per stream; side-by-side layers once for all
    import copy
    if G == 1:
        inst = layers
    else:
        inst = []

        def walk(lo, hi, group):
            """layers[lo:hi] for the streams of `group`: a layer that runs that many side by side becomes one instance; a maximal run of
            layers with smaller groups is walked once per half of the group (and so on down to single streams)."""
            i = lo
            while i < hi:
                L = layers[i]
                if L["gs"] >= len(group):
                    assert L["gs"] == len(group), (L["name"], L["gs"], group)
                    o = copy.deepcopy(L)
                    o["g0"] = group[0]
                    inst.append(o)
                    i += 1
                    continue
                j = i
                while j < hi and layers[j]["gs"] < len(group):
                    j += 1
                half = len(group) // 2
                walk(i, j, group[:half])
                walk(i, j, group[half:])
                i = j

        walk(0, len(layers), list(range(G)))
    ops = inst
    for i, o in enumerate(ops):
        o["idx"] = i
    n_ops = len(ops)

    def streams(o):
        return set(range(o["g0"], o["g0"] + o["gs"]))

    # ---- who completes whose image: op I (input / conv) builds the image of the next conv instance J -- stages its parts, zeroes its halos
    #      and forwards the rows of the streams both have, if J's layer is the one I's layer feeds ------------------------------------
    conv_idx = [o["idx"] for o in ops if o["type"] == T_CONV]
    # (packed plans: an instance of the network's LAST CTFA that is followed by another stream's ops builds the image after it -- the
    #  sub-pixel conv before it forwards into the plain rows that CTFA works on, which share LDS with that image)
    last_layer = len(layers) - 1
    chain = [o["idx"] for o in ops if o["type"] in (T_CONV, T_INPUT) or (o["type"] == T_CTFA and o["layer"] == last_layer)]
    for a, b in zip(chain, chain[1:]):
        if ops[b]["type"] == T_CONV and not (ops[a]["type"] == T_CONV and ops[a]["layer"] == conv_l[-1]):
            ops[a]["nxt"] = b          # (an input-layer instance in mid-list has no image: the op before it builds nothing)
    last_conv = ops[conv_idx[-1]]
    for o in ops:
        if o["type"] not in (T_CONV, T_INPUT, T_CTFA) or o["nxt"] < 0:
            if o["type"] in (T_CONV, T_INPUT) and o is not last_conv:
                o["fwd"] = None
            continue
        J = ops[o["nxt"]]
        gj = J["img"]
        feeds = nxt_layer.get(o["layer"]) == J["layer"]
        common = streams(o) & streams(J) if feeds else set()
        f = o["fwd"]
        if not common:
            o["fwd"] = None
        else:
            f = dict(f)
            f["gstride_b"] = gj["gstride_b"]
            f["mask"] = sum(1 << (s - o["g0"]) for s in common)
            f["base_b"] += (o["g0"] - J["g0"]) * gj["gstride_b"]
            o["fwd"] = f
        # the image of J: its layer's parts for all of J's streams, plus -- for the streams whose rows nobody forwards -- the rows of the
        # layer before it, from the HBM tensor that layer writes them to
        parts = []
        for p in J["parts"]:
            q = dict(p)
            q["g0"], q["ng"], q["gstride_b"] = J["g0"], J["gs"], gj["gstride_b"]
            parts.append(q)
        missing = sorted(streams(J) - common)
        if missing and J["layer"] in prv_layer:
            src_l = layers[prv_layer[J["layer"]]]
            assert missing == list(range(missing[0], missing[-1] + 1))
            assert src_l["type"] == T_CONV and src_l["d0"] is not None, ("no HBM copy of the rows of", src_l["name"], "for", J["name"])
            fg = src_l["fwd"] if src_l["fwd"] else None
            # (the producer LAYER's forward geometry: recompute -- an instance's may have been cleared)
            tgt_l = layers[J["layer"]]
            coff = 32 if (src_l["kind"] == K_EL and tgt_l["kind"] == K_DL) else (64 if (src_l["kind"] == K_DOWN and tgt_l["kind"] == K_UP) else 0)
            fg = fwd_geom(src_l, tgt_l, coff)
            d0 = src_l["d0"]
            rows, c4s = src_l["P"] * src_l["R"], src_l["gc"] // 4
            q = part(d0[0], d0[1], d0[2], rows, c4s, fg["base_b"] + (missing[0] - J["g0"]) * gj["gstride_b"], fg["row0"], 1)
            q["g0"], q["ng"], q["gstride_b"] = missing[0], len(missing), gj["gstride_b"]
            q["fwdsub"] = 1
            parts.append(q)
        # Packed plans: where the builder is not the op the one-stream plan stages this image behind (an instance boundary: another
        # stream's short down-sampling / CTFA op, an in-conv that also hands rows over through HBM), the loads go out one op EARLIER
        # (la 2, up to 8 carried float4 per thread) -- measured with la 1: msfe6_down_sampling 3.4 -> 8.9 us per instance, msfe6_de_ctfa
        # 3.2 -> 8.0, msfe5_down_sampling 2.1 -> 7.5 (profiles/r04_g4_v1_timeline.json): the builder waits for HBM between its barriers
        boundary = G > 1 and (bool(missing) or not feeds)
        if boundary:
            for q in parts:
                if q["la"] == 1 and (q["ng"] * q["rows"] * q["c4s"] + 511) // 512 <= 8:
                    q["la"] = 2          # (the drain rule below takes it back to 1 where the rows are not visible that early)
        J["parts"] = parts
    # the last sub-pixel conv instance(s) of the network feed the last CTFA
    for o in ops:
        if o["type"] == T_CONV and o["layer"] == conv_l[-1]:
            o["fwd"] = dict(layers[conv_l[-1]]["fwd"])
            o["nxt"] = o["nxt"] if o["nxt"] >= 0 else -1

    # LSTM / CTFA work in place on the image of the conv op that follows them
    for o in ops:
        if o["type"] in (T_LSTM, T_DDB):
            tgt = ops[o["idx"] + 1]
            g = tgt["img"]
            assert tgt["type"] == T_CONV and streams(tgt) == streams(o) and streams(ops[o["idx"] - 1]) == streams(o), o["name"]
            b0 = (g["taps"] - 1) * g["tap_b"] + g["row0"] * g["pitch_b"]
            assert not g["pair"]
            o["x_b"] = b0 + o["x_cols"] * (2 if g["fmt"] else 4)
            o["y_b"] = b0
            o["x_pitch_b"] = g["pitch_b"]
            o["x_fmt"] = g["fmt"]
            o["x_plane_b"] = g["plane_b"]
            o["x_gstride_b"] = g["gstride_b"]
            if o["gs"] > 1:
                o["scr_b"], o["scr_gstride_b"], o["xcopy_b"] = PK_LSTM_SCR_B, PK_LSTM_SCR_STRIDE, PK_XCOPY_B
                ops[o["idx"] - 1]["xcopy_b"] = PK_XCOPY_B
                assert o["gs"] * g["bytes"] <= PK_LSTM_SCR_B and PK_LSTM_SCR_B + o["gs"] * PK_LSTM_SCR_STRIDE <= PK_XCOPY_B
                prev = ops[o["idx"] - 1]
                assert prev["gs"] * prev["img"]["bytes"] <= PK_LSTM_SCR_B and prev["ex_b"] >= PK_XCOPY_B + o["gs"] * 1024, prev["name"]
            if o["type"] == T_DDB:
                assert g["bytes"] <= DDB_LDS_B and DDB_LDS_B + 17920 * 4 <= SCR_B
        if o["type"] == T_CTFA:
            prev = ops[o["idx"] - 1]
            assert prev["type"] == T_CONV and streams(prev) == streams(o) and prev["fwd"] and prev["fwd"]["mask"] == (1 << o["gs"]) - 1, o["name"]
            o["fwd"] = dict(prev["fwd"])
            o["last"] = 1 if o["layer"] == len(layers) - 1 else 0
            if o["gs"] > 1:
                o["scr_b"], o["scr_gstride_b"] = SCR_B - (o["gs"] - 1) * PK_CTFA_SCR_STRIDE, PK_CTFA_SCR_STRIDE
    assert ops[-1]["type"] == T_CTFA and ops[-1]["last"]

    # ---- checks -----------------------------------------------------------------------------------
    for o in ops:
        if o["type"] != T_CONV:
            continue
        g = o["img"]
        lim = o["ex_b"] if o["path"] == P_X16B else SCR_B
        assert o["gs"] * g["bytes"] <= lim, (o["name"], g["bytes"], lim)
        if o["nxt"] >= 0:
            J = ops[o["nxt"]]
            lim_n = lim
            if o["idx"] + 1 < n_ops and ops[o["idx"] + 1]["type"] == T_CTFA:
                lim_n = min(lim_n, ops[o["idx"] + 1]["scr_b"])          # (the CTFA between them works in place on J's image set)
            if J["gs"] * J["img"]["bytes"] > lim_n:
                raise NextImageDoesNotFit(o["name"], J["name"])
        assert len(o["parts"]) <= MAX_PARTS and len(g["zero"]) <= MAX_ZERO and o["nseg"] <= MAX_SEG, o["name"]
        assert sum(z[1] for z in g["zero"]) <= 512, o["name"]
        tasks = o["PG"] * o["CG"] * o["KSt"] * o["KSg"]
        assert tasks in (1, 2, 4, 8), (o["name"], tasks)
        assert g["fmt"] == 1 or o["path"] == P_R32B, o["name"]
        VP = o["gs"] * o["P"]
        if o["path"] == P_X16B:
            assert o["PT"] * o["PG"] * 16 >= VP and (o["cin"] // 32) % o["KSg"] == 0 and o["nseg"] % o["KSt"] == 0
        if o["path"] == P_R32B:
            assert VP % (32 * o["PT"] * o["PG"]) == 0 and (not o["ln"] or o["NT"] * 32 == o["gc"])
            assert o["gs"] == 1 or o["P"] % 32 == 0 or not o["ys"]
    # the last CTFA's plain rows
    # Same-frame HBM hand-offs (skip connections, and in packed plans the rows nobody forwards): the loads of a staged part may only be
    # issued after a drain point (every wave has waited for its own stores; LSTM / CTFA ops drain when they START, conv ops flagged
    # `drain` when they END) that follows the producing op.  Where a packed plan has none, the op before the issuing one is flagged.
    builder = {o["nxt"]: o["idx"] for o in ops if o["nxt"] >= 0}

    def drained_between(prod, issue):
        for k in range(prod, issue):
            if ops[k]["drain"] and (k > prod or ops[k]["type"] == T_CONV):
                return True
        return False

    def ensure_drained(prod, issue):
        """A drain point in [prod, issue)?  CTFA ops drain (at their start) only where a hand-off needs it -- none does in the one-stream
        plans: the LSTM of the stage has drained by then, and a CTFA's drain is a full round trip of the stores the sub-pixel conv before it
        just issued -- so: use what is there, else switch on the last CTFA in between, else (packed plans) let a conv op drain at its end."""
        if drained_between(prod, issue):
            return True
        cand = [k for k in range(prod + 1, issue) if ops[k]["type"] == T_CTFA]
        if cand:
            ops[cand[-1]]["drain"] = 1
            return True
        return False

    def producers(src, off, ld, strm):
        return [q["idx"] for q in ops for d in (q["d0"], q["d1"]) if d and d[0] == src and d[1] == off and d[2] == ld and strm in streams(q)]

    for o in ops:
        for p in o["parts"]:
            if p["src"] == S_PREV:
                continue
            b_idx = builder[o["idx"]]
            prods = []
            for strm in range(p["g0"], p["g0"] + p["ng"]):
                pr = [k for k in producers(p["src"], p["off"], p["ld"], strm) if k < o["idx"]]
                assert len(pr) == 1, (o["name"], p, pr)
                prods.append(pr[0])
            prod = max(prods)
            assert prod < b_idx or p.get("fwdsub"), (o["name"], p, prod, b_idx)
            if p["la"] == 2 and not (prod < b_idx - 1 and ensure_drained(prod, b_idx - 1)):
                p["la"] = 1
            issue = b_idx - (p["la"] - 1)
            assert prod < issue, (o["name"], "a staged part's rows are produced by the op that would load them", p, prod, issue)
            if not ensure_drained(prod, issue):
                assert G > 1, (o["name"], p, prod, issue)
                cand = [k for k in range(prod, issue) if ops[k]["type"] == T_CONV]          # (a conv op drains at its END: its own stores included)
                assert cand, (o["name"], "no op to drain in between", ops[prod]["name"], ops[issue]["name"])
                ops[cand[-1]]["drain"] = 1
            p["producer"] = prod
    # The same rule for what the kernel's weight / operand prefetch reads from THIS frame's state: a CTFA's residual rows e0
    # (fused_step.hip prefetch_w, issued TWO ops ahead of the CTFA) were written by the stage's in-conv earlier in the frame.
    for o in ops:
        if o["type"] != T_CTFA:
            continue
        for strm in streams(o):
            prod = [k for k in producers(S_CUR, o["e0_off"], o["e0_ld"], strm) if k < o["idx"]]
            assert len(prod) == 1, (o["name"], "producer of the residual rows")
            issue = o["idx"] - 2
            assert ensure_drained(prod[0], issue), (o["name"], "no drain point between", prod[0], "and the prefetch in", issue)
    # ---- lazily written states (one-stream plans): the output rows of a strided conv go to TWO state tensors -- the input state of the next
    # strided conv (d0: `*_prev{i+1}`, the signature's echo of that conv's input) and the skip-connection slice of the sub-pixel conv's input
    # state (d1).  The kernel reads the second one (a staged part of the decoder side); the first one it never reads: the next strided conv
    # takes the rows from LDS.  Such a d0 is marked `lazy` (OpD::d0_on = 2): a launch writes it only when the handle asks for eager states;
    # otherwise the library rebuilds it from the d1 copy when somebody outside the kernel looks (engine.cpp states_materialize).
    for o in ops:
        o["lazy0"] = 0
        o["lazy1"] = 0
    if G == 1 and LAZY:
        overlap = region_overlap
        readers = []
        for q in ops:
            for pp in q.get("parts", []):
                if pp["src"] in (S_CUR, S_PREV):
                    readers.append((pp["off"], pp["rows"], pp["ld"], 4 * pp["c4s"]))
            if q["type"] == T_CTFA:
                readers.append((q["e0_off"], q["F"], q["e0_ld"], 64))
        for o in ops:
            if o["type"] == T_CONV and o["kind"] == K_EL and o["d0"] and o["d1"] and o["d0"][0] == S_CUR and o["d1"][0] == S_CUR and o["R"] == 1:
                a = (o["d0"][1], o["P"], o["d0"][2], o["N"])
                if not any(overlap(a, b) for b in readers):
                    o["lazy0"] = 1
        # ... and the second copy of an encoder sub-pixel conv's rows (d1: the skip half of the paired decoder conv's input state): nothing in
        # the kernel reads it any more (the decoder's staged part above takes the rows from d0)
        for o in ops:
            if o["type"] == T_CONV and o["kind"] == K_DL and o["d0"] and o["d1"] and o["d0"][0] == S_CUR and o["d1"][0] == S_CUR:
                a = (o["d1"][1], o["P"] * o["R"], o["d1"][2], o["gc"])
                if not any(overlap(a, b) for b in readers):
                    o["lazy1"] = 1
    # ---- cache policy of the state stores (OpD::nt0 / nt1): a destination nothing in THIS launch reads again -- no same-frame staged part (skip rows,
    # up-sampling inputs, packed plans' hand-offs), no CTFA residual -- holds rows that only the next frame needs (its previous-frame taps): such stores get
    # the non-temporal hint, so that they do not push the rows that ARE re-read, and the weights, out of the L2
    overlap2 = region_overlap
    same_frame = {}
    for q in ops:
        for pp in q.get("parts", []):
            if pp["src"] in (S_CUR, S_SCRATCH):
                same_frame.setdefault(pp["src"], []).append((pp["off"], pp["rows"], pp["ld"], 4 * pp["c4s"]))
        if q["type"] == T_CTFA:
            same_frame.setdefault(S_CUR, []).append((q["e0_off"], q["F"], q["e0_ld"], 64))
    for o in ops:
        if o["type"] == T_CONV:
            for k, dk in (("nt0", o["d0"]), ("nt1", o["d1"])):
                if dk:
                    reg = (dk[1], o["P"] * o["R"], dk[2], o["gc"])
                    o[k] = 0 if any(overlap2(reg, b) for b in same_frame.get(dk[0], [])) else 1
        elif o["type"] == T_LSTM and o["ldst"]:
            reg = (o["ldst"][1], o["dout"] // o["x_cols"], o["ldst"][2], o["x_cols"])
            o["nt0"] = 0 if any(overlap2(reg, b) for b in same_frame.get(o["ldst"][0], [])) else 1
        elif o["type"] == T_CTFA and o["d0"]:
            reg = (o["d0"][1], o["F"], o["d0"][2], 64)
            o["nt0"] = 0 if any(overlap2(reg, b) for b in same_frame.get(o["d0"][0], [])) else 1
    return A, W, ops


# --------------------------------------------------------------------------------- emit
def c_part(p):
    return "{%d,%d,%d,%d,%d,%d,%d,%d,%d,%d,%d,%d}" % (p["src"], p["off"], p["ld"], p["rows"], p["c4s"], p["lds_b"], p["row0"], p["la"], p["round2"],
                                                 p["g0"], p["ng"], p["gstride_b"])


NO_PART = "{0,0,0,0,0,0,0,0,0,0,1,0}"


def c_img(o):
    g = o["img"]
    if g is None:
        return "{0,0,0,0,0,0,0,0,0,0,0,{%s},0,{%s}}" % (",".join([NO_PART] * MAX_PARTS), ",".join(["{0,0}"] * MAX_ZERO))
    parts = [c_part(p) for p in o["parts"]] + [NO_PART] * (MAX_PARTS - len(o["parts"]))
    zs = ["{%d,%d}" % z for z in g["zero"]] + ["{0,0}"] * (MAX_ZERO - len(g["zero"]))
    return "{%d,%d,%d,%d,%d,%d,%d,%d,%d,%d,%d,{%s},%d,{%s}}" % (g["fmt"], g["plane_b"], g["taps"], g["tap_b"], g["pitch_b"], g["pair"], g["half_b"], g["row0"],
                                                                g["bytes"], g["gstride_b"], len(o["parts"]), ",".join(parts), len(g["zero"]), ",".join(zs))


def c_fwd(f):
    if not f:
        return "{0,0,0,0,0,0,0,0,0,0}"
    return "{%d,%d,%d,%d,%d,%d,%d,%d,%d,%d}" % (f["on"], f["base_b"], f["pitch_b"], f["pair"], f["half_b"], f["row0"], f["fmt"], f["plane_b"],
                                           f["gstride_b"], f["mask"])


def c_dst(d, lazy=0):
    return "0,0,0,0" if d is None else "%d,%d,%d,%d" % ((2 if lazy else 1,) + tuple(d))


def pad(lst, n):
    return list(lst) + [0] * (n - len(lst))


def op_label(o):
    return o["name"] + ("#s%d" % o["g0"] if o["g0"] else "")


def emit(A, W, ops, G=1):
    L = []
    L.append("// GENERATED by tools/gen_fused_plan.py -- do not edit (tests/test_fused_plan.py checks it is current).")
    L.append("constexpr int kStreams = %d;              // streams per workgroup (1: the plan every handle can run; > 1: packed plan, OpD::gs / g0)" % G)
    L.append("constexpr int kNumOps = %d;" % len(ops))
    L.append("constexpr int kParityStride = %d;      // floats between the two buffers of every state tensor" % A.PS)
    L.append("constexpr int kArenaFloats = %d;       // per-stream arena the plan addresses (engine.cpp lays it out identically)" % A.floats)
    L.append("constexpr int kBlobFloats = %d;        // weight blob in plan order" % W.cur)
    L.append("constexpr int kYsOff = %d;             // arena offset of the two blocks of carried partial sums (block b at kYsOff + b * kYsBlock)" % A.scratch["ysum"])
    L.append("constexpr int kYsBlock = %d;" % A.ys_block)
    L.append("constexpr OpD kOps[kNumOps] = {")
    for o in ops:
        seg_b = ",".join(str(x) for x in pad(o["seg_b"], MAX_SEG))
        ldst = o["ldst"]
        row = ("{%d, /*conv*/ %d,%d,%d,%d,%d,%d,%d, %d,%d,%d,%d,%d,%d,%d, %d,%d,%d, %d, %d,{%s}, %d, %d,%d, %s, %s, %d,%d, %s, %s, %d, "
               "/*lstm*/ %d,%d,%d,%d,%d,%d,%d,%d, %d,%d,%d, %d,%d, %d, /*ctfa*/ %d,%d,%d,%d,%d, %d, %d, /*ys*/ %d,%d,%d,%d, /*streams*/ %d,%d, %d,%d, %d, %d, %d, /*epl, nt0, nt1*/ %d, %d,%d},   // %d %s") % (
            o["type"], o["kind"], o["P"], o["cin"], o["N"], o["taps"], o["kf"], o["stride"],
            o["path"], o["PT"], o["NT"], o["PG"], o["CG"], o["KSt"], o["KSg"], o["ln"], o["R"], o["gc"], o["rounds"],
            o["nseg"], seg_b, o["ex_b"], o["w_off"], o["p_off"], c_dst(o["d0"], o["lazy0"]), c_dst(o["d1"], o["lazy1"]), o["row_mul"], o["row_add"],
            c_fwd(o["fwd"]), c_img(o), o["nxt"],
            o["din"], o["dout"], o["x_b"], o["x_pitch_b"], o["x_cols"], o["y_b"], o["h_off"], o["c_off"],
            1 if ldst else 0, ldst[1] if ldst else 0, ldst[2] if ldst else 0, o["x_fmt"], o["x_plane_b"], o["lw_off"],
            o["F"], o["e0_off"], o["e0_ld"], o["last"], o["cw_off"], o["drain"], o["bidx"], o["ys"], o["ys_off"], o["xs_off"], o["xs_ld"],
            o["gs"], o["g0"], o["scr_b"], o["scr_gstride_b"], o["xcopy_b"], o["x_gstride_b"], o["layer"], o["epl"], o["nt0"], o["nt1"], o["idx"], op_label(o))
        L.append("  " + row)
    L.append("};")
    # what the host needs to pack the blob / check the arena
    L.append("struct BlobItem { int off, floats, what, op; const char* key; };   // what: 0 conv fragments, 1 conv params, 2 lstm, 3 ctfa, 4 input layer")
    what_code = {"conv_w": 0, "conv_p": 1, "lstm": 2, "ctfa": 3, "input_p": 4}
    item_op = {}
    for o in ops:
        for k in ("w_off", "p_off", "lw_off", "cw_off"):
            if o.get(k) or (k == "p_off" and o["type"] == T_INPUT):
                item_op.setdefault(o[k], o["idx"])
    L.append("constexpr int kNumBlobItems = %d;" % len(W.items))
    L.append("static const BlobItem kBlobItems[kNumBlobItems] = {")
    for (off, n, what, key) in W.items:
        L.append('  {%d, %d, %d, %d, "%s"},' % (off, n, what_code[what], item_op[off], key))
    L.append("};")
    L.append("struct StateOff { const char* name; int off; };")
    L.append("constexpr int kNumStateOffs = %d;      // ping-pong states first, then (baseline) the in-place history rings" % (len(A.states) + len(A.rings)))
    L.append("constexpr int kNumPingPong = %d;" % len(A.states))
    L.append("static const StateOff kStateOffs[kNumStateOffs] = {")
    for (n, r, c) in A.states + A.rings:
        L.append('  {"%s", %d},' % (n, A.off[n]))
    L.append("};")
    L.append("static const StateOff kScratchOffs[%d] = {" % len(A.scratch))
    for n, off in A.scratch.items():
        L.append('  {"%s", %d},' % (n, off))
    L.append("};")
    L.append("// (time tap, frequency tap) of every K segment, in the order the kernel walks them: t * 4 + kw")
    L.append("static const int kSegTk[kNumOps][%d] = {" % MAX_SEG)
    for o in ops:
        L.append("  {%s}," % ",".join(str(x) for x in pad(o["seg_tk"], MAX_SEG)))
    L.append("};")
    L.append("static const char* const kOpNames[kNumOps] = {%s};" % ", ".join('"%s"' % op_label(o) for o in ops))
    L.append("static const double kOpFlops[kNumOps] = {%s};" % ", ".join("%d" % o["flops"] for o in ops))
    return "\n".join(L) + "\n"


def plan_json(variant, G, A, W, ops):
    return json.dumps(dict(variant=variant, streams_per_workgroup=G, parity_stride=A.PS, arena_floats=A.floats, blob_floats=W.cur, state_off=A.off,
                           scratch=A.scratch, blob=[list(x) for x in W.items], ops=ops), indent=1, sort_keys=True)


def main():
    stale = False
    for variant, G in PLANS:
        A, W, ops = build(variant, G)
        inc = emit(A, W, ops, G)
        js = plan_json(variant, G, A, W, ops)
        if "--check" in sys.argv:
            ok = os.path.exists(inc_path(variant, G)) and open(inc_path(variant, G)).read() == inc and open(json_path(variant, G)).read() == js
            print("fused plan (%s) is %s" % (plan_tag(variant, G), "current" if ok else "STALE"))
            stale = stale or not ok
            continue
        for path, text in ((inc_path(variant, G), inc), (json_path(variant, G), js)):
            if not (os.path.exists(path) and open(path).read() == text):          # (an unchanged plan keeps its mtime: no rebuild of its kernel)
                open(path, "w").write(text)
        r32 = [o for o in ops if o["type"] == T_CONV and o["path"] == P_R32B]
        sbs = [o for o in ops if o["gs"] > 1]
        print("%-8s ops %d (conv %d, of which %d on 32x32 tiles; %d side by side), arena %d floats (parity stride %d), blob %d floats" % (
            plan_tag(variant, G), len(ops), sum(o["type"] == T_CONV for o in ops), len(r32), len(sbs), A.floats, A.PS, W.cur))
    sys.exit(1 if stale else 0)


if __name__ == "__main__":
    main()
