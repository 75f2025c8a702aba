"""CPU tests of the fused kernel's host side: the committed static schedule is what tools/gen_fused_plan.py writes, and the
weight blob the library packs (conv kernels int8 in MFMA fragment order) decodes -- with the walk the kernel uses -- back
to exactly the int8 payload / scales / biases of the container."""
import ctypes
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import GOLDEN

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_committed_plan_is_current():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_fused_plan.py"), "--check"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr


def _perm(o):
    n = o["N"]
    if o["kind"] == 2 and n == 64:       # sub-pixel shuffle folded into the channel order
        return np.array([2 * c + r for r in range(2) for c in range(32)])
    if o["kind"] == 2 and n == 128:
        return np.array([r * 64 + (c2 % 32) * 2 + c2 // 32 for r in range(2) for c2 in range(64)])
    return np.arange(n)


def _container(variant):
    """lstm: the shipped container (the reference's trained .tflite, conv kernels int8); baseline: random-init weights
    (no trained ones exist) written the same way."""
    from nunet_amd.weights import DEFAULT_WEIGHTS, read_blob, synthetic_weights, write_blob
    if variant == "lstm":
        return read_blob(DEFAULT_WEIGHTS)
    return write_blob(synthetic_weights("baseline", seed=11, bias_std=0.1, affine_jitter=0.1), int8_convs=True)


@pytest.mark.parametrize("variant,streams", [("lstm", 1), ("baseline", 1), ("lstm", 2), ("lstm", 4)])
def test_blob_packing_round_trips_the_container(variant, streams):
    """(streams > 1: the packed plans -- other tilings, hence another fragment order; one blob item per LAYER, shared by its instances)"""
    import nunet_amd  # noqa: F401
    from nunet_amd.runner import load_library, _fptr
    from nunet_amd.weights import parse_blob
    lib = load_library()
    blob = _container(variant)
    v = {"lstm": 0, "baseline": 1}[variant]
    n = lib.nutls_fused_plan_blob_floats(v, streams)
    assert n > 0 and (streams > 1 or n == lib.nutls_fused_blob_floats(v))
    out = np.zeros(n, np.float32)
    buf = ctypes.create_string_buffer(blob, len(blob))
    assert lib.nutls_fused_pack_blob_plan(buf, len(blob), v, streams, _fptr(out), n) == 0, lib.nutls_last_error()
    tag = ("lstm" if variant == "lstm" else "base") + ("" if streams == 1 else "_g%d" % streams)
    plan = json.load(open(os.path.join(GOLDEN, "fused_plan_%s.json" % tag)))
    assert plan["blob_floats"] == n and plan["streams_per_workgroup"] == streams
    W, Wq = parse_blob(blob), parse_blob(blob, dequantize=False)
    lane = np.arange(64)
    n_conv = 0
    for o in plan["ops"]:
        if o["type"] != 1 or o["g0"] != 0:
            continue
        n_conv += 1
        r32, up = o["path"] == 3, o["kind"] == 4     # 3: 32x32x16 tiles, 4: 16x16x32 tiles (bf16 MFMA, 8 K values per lane and fragment)
        assert o["path"] in (3, 4)
        cin, N = o["cin"], o["N"]
        q, sc = Wq[o["wkey"] + ".w"]
        perm = _perm(o)
        G = cin // (16 if r32 else 32)
        r32two = r32 and o["ys"]          # two-tap conv on 32x32 tiles: waves 0..3 own time tap 0, waves 4..7 tap 1
        wt = (2 * o["CG"] if r32two else o["CG"]) if r32 else o["CG"] * o["KSt"] * o["KSg"]
        GW = G // o["KSg"]
        segw = 3 if (up or r32two) else o["nseg"] // o["KSt"]
        nf = segw * GW * o["NT"]
        nsf = (nf + 1) // 2
        raw = out[o["w_off"]:o["w_off"] + wt * nsf * 256].view(np.int8).reshape(wt, nsf, 64, 16)
        rec = np.full(q.shape, 99, np.int32)
        for task in range(wt):
            ct, ks = task % o["CG"], task // o["CG"]
            ks_g, ks_t = ks % o["KSg"], ks // o["KSg"]
            for f in range(nf):          # the walk of conv_x16b / conv_r32b (fused_step.hip)
                if r32:
                    nt, sgi = f % o["NT"], f // o["NT"]
                    sg, g, T = (3 * ks if r32two else 0) + sgi // G, sgi % G, ct * o["NT"] + nt
                    npk, c0 = 32 * T + (lane & 31), 16 * g + 8 * (lane >> 5)
                else:
                    nt, r = f % o["NT"], f // o["NT"]          # (NT channel tiles per task, the tile index fastest)
                    sg, g, T = ks_t * segw + r // GW, ks_g * GW + r % GW, ct * o["NT"] + nt
                    npk, c0 = 16 * T + (lane & 15), 32 * g + 8 * (lane >> 4)
                t, kw = o["seg_tk"][sg] >> 2, o["seg_tk"][sg] & 3
                for qq in range(8):
                    rec[perm[npk], t, kw, c0 + qq] = raw[task, f // 2, :, (f % 2) * 8 + qq]
        assert np.array_equal(rec, q.astype(np.int32)), o["name"]          # every weight covered, every byte right
        ntot, gc = N * (2 if up else 1), o["gc"]
        blk = out[o["p_off"]:o["p_off"] + 2 * ntot + 2 * gc + 1]
        assert np.array_equal(blk[:N], W[o["wkey"] + ".b"][perm]), o["name"]
        assert np.array_equal(blk[ntot:ntot + N], sc[perm] if sc.size > 1 else np.full(N, sc[0], np.float32)), o["name"]
        if o["ln"]:
            assert np.array_equal(blk[2 * ntot:2 * ntot + gc], W[o["wkey"] + ".gamma"].reshape(-1))
            assert np.array_equal(blk[2 * ntot + gc:2 * ntot + 2 * gc], W[o["wkey"] + ".beta"].reshape(-1))
            assert blk[2 * ntot + 2 * gc] == W[o["wkey"] + ".alpha"].reshape(-1)[0]
    assert n_conv == 128
    # LSTM + Dense ops: the int8 gate kernels [K slice][unit][row] with the four gates of a unit in one dword, the record (bias | s_x, s_h), the
    # Dense rows (int8 + bias + scale where the container holds them int8, fp32 otherwise) -- csrc/fused_plan.hpp "blob layout of an LSTM op"
    n_lstm = 0
    for o in plan["ops"]:
        if o["type"] != 2 or o["g0"] != 0:
            continue
        n_lstm += 1
        ln, dn = ("lstm", "dense") if o["wkey"] == "" else (o["wkey"] + "_lstm", o["wkey"] + "_dense")
        din, dout = o["din"], o["dout"]
        kn = din // 16
        nrp = (max(kn, 6) + 3) // 4 * 4
        gates = out[o["lw_off"]:o["lw_off"] + 420 * nrp].view(np.int8).reshape(20, 21, nrp, 4)
        (qx, sx), (qh, sh) = Wq[ln + ".wx"], Wq[ln + ".wh"]
        for g in range(4):          # Keras gate order i, f, g, o: rows g * 21 + u of the [84, K] kernels
            assert np.array_equal(gates[:16, :, :kn, g].transpose(1, 0, 2).reshape(21, din), qx[g * 21:(g + 1) * 21]), ln
            hpart = gates[16:, :, :6, g].transpose(1, 0, 2).reshape(21, 24)
            assert np.array_equal(hpart[:, :21], qh[g * 21:(g + 1) * 21]) and not hpart[:, 21:].any(), ln
        assert not gates[:16, :, kn:].any() and not gates[16:, :, 6:].any()
        rec = out[o["lw_off"] + 420 * nrp:o["lw_off"] + 420 * nrp + 88]
        assert np.array_equal(rec[:84].reshape(21, 4).T.reshape(-1), W[ln + ".b"]) and rec[84] == sx[0] and rec[85] == sh[0]
        drow = out[o["lw_off"] + 420 * nrp + 88:]
        if dout >= 64:
            qd, sd = Wq[dn + ".w"]
            rows = drow[:8 * dout].reshape(dout, 8)
            assert np.array_equal(rows[:, :6].copy().view(np.int8).reshape(dout, 24)[:, :21], qd) and np.array_equal(rows[:, 6], W[dn + ".b"])
            assert np.all(rows[:, 7] == sd[0])
        else:
            rows = drow[:24 * dout].reshape(dout, 24)
            assert np.array_equal(rows[:, :21], W[dn + ".w"]) and np.array_equal(rows[:, 21], W[dn + ".b"])
    assert n_lstm == (13 if variant == "lstm" else 0)


@pytest.mark.parametrize("name,bottleneck", [("lstm", 2), ("base", 4)])
def test_plan_images_fit_lds_and_ops_cover_the_network(name, bottleneck):
    plan = json.load(open(os.path.join(GOLDEN, "fused_plan_%s.json" % name)))
    ops = plan["ops"]
    assert len(ops) == 154 and sum(o["type"] == bottleneck for o in ops) == 13 and sum(o["type"] == 3 for o in ops) == 12
    assert sum(o["type"] in (2, 4) for o in ops) == 13
    flops = sum(o["flops"] for o in ops)
    # 147.9 MFLOP per frame and stream; the plan counts the conv ops only (CTFA 1x1s as lowered by TFLite: 4.2 M, LSTM + Dense 0.3 M)
    assert 0.96 < flops / (2 * 73_967_252) < 1.0
    for o in ops:
        if o["type"] == 1:
            lim = o["ex_b"] if o["path"] == 4 else 160 * 1024 - 8192
            assert o["img"]["bytes"] <= lim, o["name"]
            for p in o["parts"]:
                assert p["la"] in (1, 2)


@pytest.mark.parametrize("name", ["lstm", "base"])
def test_carried_partial_sums_layout(name):
    """The strided convs hand W[tap 0] x_t to the next frame as P x 32 fp32 sums (OpD.ys): their image holds one time tap, no part
    of it comes from the previous frame's state, the sums of all of them tile one block exactly, the arena holds two blocks after
    the scratch tensors, and every such op knows the state tensor the host rebuilds its sums from."""
    plan = json.load(open(os.path.join(GOLDEN, "fused_plan_%s.json" % name)))
    ops = [o for o in plan["ops"] if o["type"] == 1]
    ys = [o for o in ops if o["ys"]]
    assert len(ys) == 52 and all(o["kind"] == 1 for o in ys) and all(o["ys"] == (o["kind"] == 1) for o in ops)
    spans = sorted((o["ys_off"], o["ys_off"] + (o["P"] * o["N"] + 63) // 64 * 64) for o in ys)
    assert spans[0][0] == 0 and all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
    block = spans[-1][1]
    assert plan["arena_floats"] == plan["scratch"]["ysum"] + 2 * block
    state_off = set(plan["state_off"].values())
    for o in ys:
        assert o["img"]["taps"] == 1 and o["rounds"] == 1 and o["nseg"] == 6 and o["seg_b"][:3] == o["seg_b"][3:]
        assert all(p["src"] != 0 for p in o["parts"])            # nothing staged from the `prev` parity
        assert o["xs_off"] in state_off and o["xs_ld"] == o["cin"]
    for o in ops:
        if o["kind"] == 2:                                        # sub-pixel convs keep the two-tap image
            assert o["img"]["taps"] == 2 and any(p["src"] == 0 for p in o["parts"])


def test_malformed_containers_are_error_codes_not_crashes():
    """The container parser never trusts header fields (huge / zero dims, truncated payloads, wrapping products):
    every malformed blob comes back as NUTLS_ERR_WEIGHTS through the C ABI (host-only entry point, no GPU needed)."""
    import struct
    import nunet_amd  # noqa: F401
    from nunet_amd.runner import load_library, _fptr
    from nunet_amd.weights import DEFAULT_WEIGHTS, read_blob
    lib = load_library()
    n = lib.nutls_fused_blob_floats(0)
    out = np.zeros(n, np.float32)
    good = read_blob(DEFAULT_WEIGHTS)

    def rc_of(b):
        buf = ctypes.create_string_buffer(bytes(b), len(b))
        return lib.nutls_fused_pack_blob(buf, len(b), 0, _fptr(out), n)

    def tensor(name, dtype, dims, ns=0, payload=b""):
        nb = name.encode()
        return struct.pack("<H", len(nb)) + nb + struct.pack("<BB", dtype, len(dims)) + struct.pack("<%dI" % len(dims), *dims) + struct.pack("<I", ns) + payload

    head = b"NUTLSW01" + struct.pack("<I", 1)
    bad = [
        good[:1000],                                                        # truncated
        b"XXXXXXXX" + good[8:],                                             # wrong magic
        head + tensor("a.w", 0, (0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF)),      # product overflows / huge resize
        head + tensor("a.w", 1, (0, 4)),                                    # zero dim (division by dims[0])
        head + tensor("a.w", 0, (1 << 30,)),                                # 4 GiB of payload that is not there
        head + tensor("a.w", 1, (4, 4), ns=0x40000000),                     # scale count beyond the file
        good + b"\0",                                                       # trailing bytes
    ]
    for b in bad:
        assert rc_of(b) == -2, lib.nutls_last_error()                      # NUTLS_ERR_WEIGHTS
    assert rc_of(good) == 0


@pytest.mark.parametrize("G", [2, 4])
def test_packed_plans_cover_every_layer_for_every_stream(G):
    """Packed plans (G streams per workgroup): the op instances are the layers of the one-stream plan -- each exactly once per stream,
    gs = G, G / 2, .. or 1 streams side by side per instance --, a stream walks its layers in the order of the
    one-stream plan, everything an op touches fits LDS, rows are forwarded only to the op right after, and every same-frame HBM
    hand-off has a drain point between the producer and the op that issues the loads."""
    one = json.load(open(os.path.join(GOLDEN, "fused_plan_lstm.json")))
    plan = json.load(open(os.path.join(GOLDEN, "fused_plan_lstm_g%d.json" % G)))
    ops = plan["ops"]
    assert plan["streams_per_workgroup"] == G and plan["arena_floats"] == one["arena_floats"] and plan["state_off"] == one["state_off"]
    names = [o["name"] for o in one["ops"]]
    for s in range(G):
        mine = [o for o in ops if o["g0"] <= s < o["g0"] + o["gs"]]
        assert [o["name"] for o in mine] == names, "stream %d does not walk the network in order" % s
        assert [o["layer"] for o in mine] == list(range(len(names)))
    assert all(o["gs"] in (1, 2, 4) and o["gs"] <= G and o["g0"] % o["gs"] == 0 for o in ops)          # groups: G, G / 2, .. 1 streams, aligned
    assert sum(o["gs"] == G for o in ops) > len(names) // 2          # most layers run side by side
    SCR_B = 160 * 1024 - 8192
    for o in ops:
        if o["type"] == 1:
            lim = o["ex_b"] if o["path"] == 4 else SCR_B
            assert o["gs"] * o["img"]["bytes"] <= lim, o["name"]
            assert o["img"]["gstride_b"] == (o["img"]["bytes"] if o["gs"] > 1 else 0)
            if o["nxt"] >= 0:
                J = ops[o["nxt"]]
                assert J["type"] == 1 and J["idx"] > o["idx"] and J["gs"] * J["img"]["bytes"] <= lim, (o["name"], J["name"])
            f = o["fwd"]
            if f and o["nxt"] >= 0:          # forwarded rows land in the image this op completes, for streams both ops have
                J = ops[o["nxt"]]
                for i in range(o["gs"]):
                    if (f["mask"] >> i) & 1:
                        assert J["g0"] <= o["g0"] + i < J["g0"] + J["gs"]
        if o["type"] == 1 or (o["type"] == 3 and o["nxt"] >= 0):
            for p in (ops[o["nxt"]]["parts"] if o["nxt"] >= 0 else []):
                assert p["la"] in (1, 2) and p["ng"] >= 1 and ops[o["nxt"]]["g0"] <= p["g0"] and p["g0"] + p["ng"] <= ops[o["nxt"]]["g0"] + ops[o["nxt"]]["gs"]
                if p["src"] != 0:
                    issue = o["idx"] - (p["la"] - 1)
                    prod = p["producer"]
                    assert prod < issue
                    assert any(ops[k]["drain"] and (k > prod or ops[k]["type"] == 1) for k in range(prod, issue)), (ops[o["nxt"]]["name"], p)
        if o["type"] == 2:          # LSTM: side by side, its scratch in the middle of LDS
            assert o["gs"] == G and o["scr_b"] + G * o["scr_gstride_b"] <= o["xcopy_b"] and o["xcopy_b"] + G * 1024 <= ops[o["idx"] - 1]["ex_b"]


def _input_rows(o):
    """rows x channels of one time tap of a conv op's input (per stream): what its LDS image must hold before its MFMA loop starts"""
    kind, P = o["kind"], o["P"]
    return {0: P, 1: 2 * P, 2: P, 3: 2 * P, 4: P}[kind], o["cin"]


@pytest.mark.parametrize("tag", ["lstm", "base", "lstm_g2", "lstm_g4"])
def test_every_image_is_completed_exactly_once(tag):
    """Data flow of a plan, re-derived from its records alone: for every conv op instance, every stream it runs and every time tap, each
    (input row, 32-channel block) of its LDS image is written exactly once before the op starts -- by the rows the op before it forwards
    from registers (Fwd, for the streams in its mask), by the LSTM / dilated-dense op in between, or by a staged part (HBM -> LDS) -- with
    no gaps and no overlaps, inside the stream's own sub-image; and a part that reads this frame's rows reads what an EARLIER op of the
    same stream wrote to exactly that tensor."""
    plan = json.load(open(os.path.join(GOLDEN, "fused_plan_%s.json" % tag)))
    ops = plan["ops"]
    builder = {o["nxt"]: o for o in ops if o["nxt"] >= 0}
    n_checked = 0
    for J in ops:
        if J["type"] != 1:
            continue
        g = J["img"]
        esz = 2 if g["fmt"] else 4
        rows, cin = _input_rows(J)
        cover = {}          # (stream, tap, row, channel block of 32) -> how often written

        def put(stream, lds_b, row0, nrows, nch, what):
            sub = lds_b - (stream - J["g0"]) * g["gstride_b"]
            tap = 1 if (g["taps"] == 2 and sub >= g["tap_b"]) else 0
            ch0 = (sub - tap * g["tap_b"]) // esz
            assert 0 <= ch0 and ch0 + nch <= cin and ch0 % 32 == 0 and nch % 32 == 0, (J["name"], what, ch0, nch)
            for r in range(nrows):
                lr = row0 + r - g["row0"]          # input row (the image's first rows are the halo)
                assert 0 <= lr < rows, (J["name"], what, lr, rows)
                for cb in range(ch0 // 32, (ch0 + nch) // 32):
                    cover[(stream, tap, lr, cb)] = cover.get((stream, tap, lr, cb), 0) + 1

        B = builder.get(J["idx"])
        if B is not None and B["fwd"] and B["type"] in (0, 1):
            f = B["fwd"]
            out_rows, out_ch = (256, 64) if B["type"] == 0 else (B["P"] * B["R"], B["gc"])
            for i in range(B["gs"]):
                if (f["mask"] >> i) & 1:
                    s = B["g0"] + i
                    assert J["g0"] <= s < J["g0"] + J["gs"]
                    put(s, f["base_b"] + i * f["gstride_b"] + (B["g0"] - J["g0"]) * 0, f["row0"], out_rows, out_ch, "fwd of " + B["name"])
        prev = ops[J["idx"] - 1]
        if prev["type"] in (2, 4):          # LSTM / dilated-dense op: writes [F][x_cols] at channels 0 .. x_cols - 1 of the current tap
            F = prev["dout"] // prev["x_cols"]
            for i in range(prev["gs"]):
                put(prev["g0"] + i, prev["y_b"] - g["row0"] * g["pitch_b"] + i * prev["x_gstride_b"] + (prev["g0"] - J["g0"]) * g["gstride_b"],
                    g["row0"], F, prev["x_cols"], prev["name"])
        for p in J["parts"]:
            for i in range(p["ng"]):
                s = p["g0"] + i
                put(s, p["lds_b"] + i * p["gstride_b"], p["row0"], p["rows"], 4 * p["c4s"], "part")
                if p["src"] != 0:          # this frame's rows: an earlier op of this stream wrote exactly that tensor
                    prod = ops[p["producer"]]
                    assert prod["idx"] < J["idx"] and any(d and d[0] == p["src"] and d[1] == p["off"] and d[2] == p["ld"] for d in (prod["d0"], prod["d1"]))
        want = {(s, t, r, cb) for s in range(J["g0"], J["g0"] + J["gs"]) for t in range(g["taps"]) for r in range(rows) for cb in range(cin // 32)}
        if J["cin"] < 32:
            continue
        missing = want - set(cover)
        extra = {k: v for k, v in cover.items() if v != 1 or k not in want}
        assert not missing, (tag, J["name"], J["g0"], sorted(missing)[:4], len(missing))
        assert not extra, (tag, J["name"], J["g0"], list(extra.items())[:4])
        n_checked += 1
    assert n_checked >= 128
