"""CPU tests of the oracle itself (no GPU): oracle B (stage-structured restatement) against
the committed golden vectors produced by oracle A (op-by-op execution of the reference's
shipped graph), the reference-derived known answers, and -- in the build container only --
oracle A re-run live against the same goldens."""
import os

import numpy as np
import pytest

from nunet_amd import stream_enhance as SE, topology as T
from nunet_amd.weights import load_weights
from oracle.nutls_ref import NutlsRef

from conftest import GOLDEN, REFERENCE_TFLITE


def rms(a, b):
    return float(np.sqrt(np.mean((np.asarray(a, np.float64) - np.asarray(b, np.float64)) ** 2)))


def test_weight_container_inventory():
    w = load_weights()
    layers = {k.split(".")[0] for k in w}
    assert len(layers) == 180            # 178 conv-like + lstm pairs share prefixes: see topology
    n_ln = sum(1 for k in w if k.endswith(".gamma"))
    assert n_ln == 117                   # 13 in-convs + 52 strided + 52 sub-pixel convs
    assert sum(1 for k in w if k.endswith(".wx")) == 13
    assert w["msfe6_en_spconv6.w"].shape == (128, 2, 3, 64)
    assert w["msfe3_upsampling.w"].shape == (128, 1, 3, 128)
    assert w["lstm.wx"].shape == (84, 256) and w["dense.w"].shape == (256, 21)
    assert sum(v.size for v in w.values()) == 2832910


def test_topology_matches_reference_signature():
    # converter_proposed.py:26-187: 1 input + 104 conv states + 26 LSTM states; 205 090 floats
    assert len(T.input_names()) == 131 and len(T.output_names()) == 131
    assert T.state_floats_per_stream() == 205090
    assert "msfe4_ee2_prev3" in T.input_names() and "msfe4_dd3_cur1" in T.output_names()
    shapes = {b.format("prev"): s for b, s in T.state_specs()}
    assert shapes["msfe6_ee_prev1"] == (1, 256, 64) and shapes["msfe6_de_prev1"] == (1, 256, 128)
    assert shapes["msfe4_ee3_prev4"] == (1, 2, 32) and shapes["msfe3_dd_prev1"] == (1, 1, 64)


def test_kat_b1_closed_form():
    """x[k] = 0.5+0.5 sin(0.1k) three frames from zero state."""
    kat = np.load(os.path.join(GOLDEN, "kat_b1.npz"))
    # the numbers printed in SURVEY.md B.1 (oracle A, survey session) pin the fixture itself
    np.testing.assert_allclose(kat["out1"][:6], [0.841054, 0.276825, 0.561412, 0.719467, 0.657214, 0.764735], atol=2e-6)
    np.testing.assert_allclose(kat["out3"][250:], [0.452088, 0.508922, 0.547724, 0.601137, 0.641985, 0.709822], atol=2e-6)
    assert abs(float(kat["out2"].sum()) - 130.80884) < 2e-3
    assert abs(float(kat["state_h3"][1]) - 0.995055) < 2e-6
    ref = NutlsRef(batch=1)
    for fr in (1, 2, 3):
        out = ref.step(kat["x"].reshape(1, 256)).numpy().reshape(-1)
        assert rms(out, kat["out%d" % fr]) < 1e-6
        np.testing.assert_allclose(ref.state["state_h"].numpy().reshape(-1), kat["state_h%d" % fr], atol=1e-5)
        assert abs(float(ref.state["msfe6_ee_prev2"].sum()) - float(kat["ee_cur2_sum%d" % fr])) < 1e-2


def test_oracle_b_matches_golden_clip_and_states():
    clip = np.load(os.path.join(GOLDEN, "clip_4s.npz"))
    ref = NutlsRef(batch=1)
    outs = []
    for i in range(64):
        outs.append(ref.step(clip["mags_in"][i:i + 1]).numpy().reshape(-1))
        if i + 1 in (3, 64):
            st = np.load(os.path.join(GOLDEN, "state_f%d.npz" % (i + 1)))
            for base, shp in T.state_specs():
                k_ref = base if len(shp) == 1 else base.format("prev")
                k_gold = base if len(shp) == 1 else base.format("cur")
                a, b = ref.state[k_ref].numpy().reshape(-1), st[k_gold].reshape(-1)
                # LSTM cell states reach |c| ~ 64, so the tolerance is relative
                np.testing.assert_allclose(a, b, rtol=2e-5, atol=2e-5, err_msg=k_gold)
    assert rms(np.stack(outs), clip["mags_out"][:64]) < 1e-6


def test_oracle_b_batch_invariance():
    """Streams are independent: a stream's output does not depend on batch size/position."""
    clip = np.load(os.path.join(GOLDEN, "clip_4s.npz"))
    a = NutlsRef(batch=1)
    b = NutlsRef(batch=3)
    for i in range(4):
        x = clip["mags_in"][i + 10]
        xb = np.stack([clip["mags_in"][i + 50], x, clip["mags_in"][i + 100]])
        oa = a.step(x[None]).numpy()[0]
        ob = b.step(xb).numpy()[1]
        assert rms(oa, ob) < 1e-6


def test_oracle_b_signature_call_round_trip():
    """The named-tensor surface (interpreter_proposed.py:215-350) equals the batched step."""
    clip = np.load(os.path.join(GOLDEN, "clip_4s.npz"))
    sig, eng = NutlsRef(batch=1), NutlsRef(batch=1)
    out = SE.zero_state()
    for i in range(3):
        m = clip["mags_in"][i]
        out = sig.signature_call(**SE.feeds_from_outputs(out, m.reshape(1, 1, 256, 1)))
        o2 = eng.step(m[None]).numpy().reshape(-1)
        assert out["model_out"].shape == (1, 1, 256, 1) and out["msfe6_en_h"].shape == (1, 21)
        assert rms(out["model_out"].reshape(-1), o2) == 0.0
    with pytest.raises(ValueError):
        sig.signature_call(input=np.zeros((1, 1, 256, 1), np.float32))


@pytest.mark.reference
def test_oracle_a_live_reproduces_goldens(has_reference):
    """Build container only: re-execute the shipped flatbuffer and compare with the fixtures."""
    if not has_reference:
        pytest.skip("/root/reference not present")
    from oracle.graph_exec import GraphOracle
    g = GraphOracle(REFERENCE_TFLITE)
    assert g.key == "nutls_lstm_sm" and len(g.model.ops) == 3066 and len(g.model.tensors) == 4054
    clip = np.load(os.path.join(GOLDEN, "clip_4s.npz"))
    out = SE.zero_state()
    for i in range(3):
        out = g(**SE.feeds_from_outputs(out, clip["mags_in"][i].reshape(1, 1, 256, 1)))
        assert rms(out["model_out"].reshape(-1), clip["mags_out"][i]) < 1e-7
    with pytest.raises(ValueError):
        g(input=np.zeros((1, 1, 256, 1), np.float32))


@pytest.mark.reference
def test_hybrid_quantisation_gap_of_the_tflite_runtime(has_reference):
    """Build container only.  "Parity unpinned at the TF-Lite-runtime level" as a NUMBER: the reference's runtime quantises
    the activations entering every int8-weight CONV_2D / FULLY_CONNECTED to int8 per call (hybrid kernels); `GraphOracle(hybrid=True)` emulates
    that.  Over the first 40 frames of the golden clip its output is 3.0e-3 RMS away from the float execution of the same graph
    (median per frame 1.3e-3, worst frame 1.5e-2; SURVEY measured 4.5e-3 over 100 frames) -- 15 000 x the distance between this repo's HIP kernels and
    the float oracle (2e-7), and above the north-star's 1e-3: no float implementation can be within 1e-3 of that runtime, whatever it
    does.  A maintainer with TensorFlow installed checks the emulation in one line:
        tf.lite.Interpreter("nutls_lstm.tflite").get_signature_runner("nutls_lstm_sm")(**feeds)  vs  GraphOracle(path, hybrid=True)(**feeds)"""
    if not has_reference:
        pytest.skip("/root/reference not present")
    from oracle.graph_exec import GraphOracle, _fake_quant_rows
    # the quantiser itself: codes are int8, zero is exact, the error is at most half a step
    x = np.array([[-1.0, 0.0, 0.3, 2.0], [0.5, 0.25, 0.0, 0.125]], np.float32)
    xq = _fake_quant_rows(x)
    assert xq[0, 1] == 0.0 and np.all(np.abs(xq - x) <= np.array([[3.0 / 255 / 2], [0.5 / 255 / 2]]) + 1e-7)
    gf, gh = GraphOracle(REFERENCE_TFLITE, backend="torch"), GraphOracle(REFERENCE_TFLITE, backend="torch", hybrid=True)
    clip = np.load(os.path.join(GOLDEN, "clip_4s.npz"))
    of, oh = SE.zero_state(), SE.zero_state()
    se, n, per_frame = 0.0, 40, []
    for i in range(n):
        m = clip["mags_in"][i].reshape(1, 1, 256, 1)
        of = gf(**SE.feeds_from_outputs(of, m))
        oh = gh(**SE.feeds_from_outputs(oh, m))
        d = (of["model_out"] - oh["model_out"]).reshape(-1).astype(np.float64)
        per_frame.append(float(np.sqrt(np.mean(d * d))))
        se += float(np.mean(d * d))
    gap = float(np.sqrt(se / n))
    print("hybrid-quantisation gap over %d frames: %.3e RMS (median per frame %.3e, max %.3e)" % (n, gap, float(np.median(per_frame)), max(per_frame)))
    assert 1e-3 < gap < 2e-2
    # and the float execution is the one the goldens hold (the emulation did not leak into it)
    assert rms(of["model_out"].reshape(-1), clip["mags_out"][n - 1]) < 1e-6


# ------------------------------------------------------------------------------------------------
#  Baseline variant (dilated-dense bottleneck, BASELINE config 3).  No trained weights / goldens
#  exist anywhere, so the STREAMING restatement is pinned against an independent
#  OFFLINE formulation of the same block: the non-"valid" Keras layers of nunet_tls.py:190-272
#  (causal ZeroPadding2D((d,0),(d,d)) + dilated, grouped Conv2D) applied to a whole sequence.
# ------------------------------------------------------------------------------------------------
def _offline_ddb(w, tag, x_seq):
    """x_seq [T,F,C] -> [T,F,C] with torch conv2d on NCHW tensors (H = time, W = frequency)."""
    import torch
    import torch.nn.functional as Fn
    T_, F, C = x_seq.shape
    G = C // 2

    def prelu(y, name):
        a = float(w[name + ".alpha"].reshape(()))
        return torch.clamp(y, min=0) + a * torch.clamp(y, max=0)

    def ohwi(name):    # [O,kh,kw,I] -> OIHW
        return torch.from_numpy(w[name + ".w"]).permute(0, 3, 1, 2).contiguous()

    x = torch.from_numpy(x_seq).permute(2, 0, 1).unsqueeze(0)                 # [1,C,T,F]
    o = [prelu(Fn.conv2d(Fn.pad(x, (1, 1, 1, 0)), ohwi(tag + "_in"), torch.from_numpy(w[tag + "_in.b"])), tag + "_in")]
    for k in range(1, 7):
        d = 1 << (k - 1)
        n = "%s_%d" % (tag, k)
        inp = torch.cat(o[::-1], dim=1)                                        # newest first, k*G channels
        wg = torch.from_numpy(w[n + ".wg"]).permute(0, 3, 1, 2).contiguous()  # [G,k,2,3]: groups=G, k in-ch per group
        y = Fn.conv2d(Fn.pad(inp, (d, d, d, 0)), wg, torch.from_numpy(w[n + ".bg"]), dilation=d, groups=G)
        z = Fn.conv2d(y, torch.from_numpy(w[n + ".w1"]).reshape(G, G, 1, 1), torch.from_numpy(w[n + ".b1"]))
        z = z.permute(0, 2, 3, 1)                                              # channels last for LN
        mu = z.mean(-1, keepdim=True)
        var = ((z - mu) ** 2).mean(-1, keepdim=True)
        z = (z - mu) * torch.rsqrt(var + 1e-8) * torch.from_numpy(w[n + ".gamma"]) + torch.from_numpy(w[n + ".beta"])
        o.append(prelu(z.permute(0, 3, 1, 2), n))
    out = prelu(Fn.conv2d(Fn.pad(o[-1], (1, 1, 1, 0)), ohwi(tag + "_out"), torch.from_numpy(w[tag + "_out.b"])), tag + "_out")
    return out[0].permute(1, 2, 0).numpy()


def test_baseline_streaming_ddb_equals_offline_dilated_conv():
    """Every one of the 13 bottlenecks, WHOLE (in conv + six grouped dilated blocks + out conv, nunet_tls.py:190-272):
    oracle B's streaming form (rings, converter_nunet_tls.py:374-411) against the offline Keras formulation evaluated
    with torch.nn.functional.conv2d(groups=G, dilation=d) over a 70-frame sequence."""
    import torch
    from nunet_amd.weights import synthetic_weights
    w = synthetic_weights("baseline", seed=7, bias_std=0.2, affine_jitter=0.2)
    ref = NutlsRef(w, batch=1, variant="baseline")
    rng = np.random.default_rng(3)
    for prefix, F, C in T.bottlenecks():
        tag = (prefix + "_ddb") if prefix else "ddb"
        T_ = 70                                                  # > 2 * 32 frames: every ring wraps
        xs = rng.standard_normal((T_, F, C)).astype(np.float32)
        want = _offline_ddb(w, tag, xs)
        got = []
        for t in range(T_):
            ref._new = {}
            got.append(ref._ddb(torch.from_numpy(xs[t:t + 1]), tag).numpy()[0])
            ref.state.update(ref._new)
        assert rms(np.stack(got), want) < 2e-6, tag


def test_baseline_topology_matches_reference_signature():
    # converter_nunet_tls.py:41-249: 104 conv states + 13 x 8 dilated-dense states = 208; 411 904 floats
    specs = T.state_specs("baseline")
    assert len(specs) == 208 and T.state_floats_per_stream("baseline") == 411904
    shapes = {b.format("prev"): s for b, s in specs}
    assert shapes["msfe6_en_ddb_prev4"] == (8, 4, 64) and shapes["ddb_prev6"] == (32, 4, 192)
    assert shapes["ddb_prev_in"] == (1, 4, 64) and shapes["msfe3_de_ddb_prev_out"] == (1, 1, 16)


def test_oracle_b_lstm_cell_equals_torch_lstmcell():
    """Oracle B's hand-written Keras LSTM cell (gates i, f, g, o; proposed.py:70-119) against torch.nn.LSTMCell loaded
    with the same weights (torch's gate order is i, f, g, o as well), for all 13 LSTMs, 5 steps of carried state."""
    import torch
    w = load_weights()
    ref = NutlsRef(batch=2)
    rng = np.random.default_rng(11)
    for prefix, F, C in T.bottlenecks():
        lstm = (prefix + "_lstm") if prefix else "lstm"
        dense = (prefix + "_dense") if prefix else "dense"
        hn, cn = ((prefix + "_h", prefix + "_c") if prefix else ("state_h", "state_c"))
        din = w[lstm + ".wx"].shape[1]
        cell = torch.nn.LSTMCell(din, T.LSTM_UNITS)
        with torch.no_grad():
            cell.weight_ih.copy_(torch.from_numpy(w[lstm + ".wx"]))
            cell.weight_hh.copy_(torch.from_numpy(w[lstm + ".wh"]))
            cell.bias_ih.copy_(torch.from_numpy(w[lstm + ".b"]))
            cell.bias_hh.zero_()
        h = torch.zeros(2, T.LSTM_UNITS)
        c = torch.zeros(2, T.LSTM_UNITS)
        ref.state[hn], ref.state[cn] = h.clone(), c.clone()
        for _ in range(5):
            v = torch.from_numpy(rng.standard_normal((2, din)).astype(np.float32))
            ref._new = {}
            got = ref._lstm_dense(v, lstm, dense, hn, cn)
            ref.state.update(ref._new)
            with torch.no_grad():
                h, c = cell(v, (h, c))
                want = torch.nn.functional.linear(h, torch.from_numpy(w[dense + ".w"]), torch.from_numpy(w[dense + ".b"]))
            assert rms(got.numpy(), want.numpy()) < 1e-6, lstm
            assert rms(ref.state[cn].numpy(), c.numpy()) < 1e-6, lstm


@pytest.mark.skipif(not os.path.exists(REFERENCE_TFLITE), reason="needs /root/reference (build container only)")
def test_oracle_a_numpy_ops_agree_with_torch_functional():
    """The goldens come from oracle A's hand-written numpy operators.  Here the same flatbuffer runs with every
    FLOP-carrying operator (CONV_2D, TRANSPOSE_CONV, FULLY_CONNECTED, AVERAGE_POOL_2D, LOGISTIC, TANH, PRELU) evaluated
    by torch.nn.functional instead: third-party arithmetic must reproduce the committed goldens too.  (Still not the
    TF-Lite runtime -- that stays unpinned, SURVEY F6 -- but no longer pinned to hand-written numpy alone.)"""
    from oracle.graph_exec import GraphOracle
    clip = np.load(os.path.join(GOLDEN, "clip_4s.npz"))
    a, b = GraphOracle(REFERENCE_TFLITE), GraphOracle(REFERENCE_TFLITE, backend="torch")
    fa, fb = a.zero_feeds(), b.zero_feeds()
    for i in range(4):
        x = clip["mags_in"][i].reshape(1, 1, 256, 1)
        fa["input"], fb["input"] = x, x
        oa, ob = a(**fa), b(**fb)
        assert rms(ob["model_out"].reshape(-1), clip["mags_out"][i]) < 1e-6
        for k in oa:
            # (saturated LSTM cell states accumulate the last-bit difference between the two sigmoid / tanh implementations)
            assert rms(oa[k], ob[k]) < 5e-6 * max(1.0, float(np.abs(oa[k]).max())), k
        if i == 2:
            fb_prev3 = dict(ob)
        fa = {k.replace("_cur", "_prev"): v for k, v in oa.items() if k != "model_out"}
        fb = {k.replace("_cur", "_prev"): v for k, v in ob.items() if k != "model_out"}
    st = np.load(os.path.join(GOLDEN, "state_f3.npz"))         # the committed state goldens after frame 3 = feeds of frame 4
    for k in ("msfe6_ee_cur1", "msfe4_dd3_cur2", "msfe3_de_cur1", "msfe6_dd_cur6"):
        np.testing.assert_allclose(fb_prev3[k].reshape(-1), st[k].reshape(-1), rtol=1e-4, atol=1e-4, err_msg=k)   # same tolerance as the GPU golden test


def test_oracle_b_causal32_ctfa_equals_the_offline_pooling_formulation():
    """ctfa_mode="causal32": oracle B's frame-by-frame history against the offline model's layers evaluated on a whole
    sequence (models/proposed.py:143-147: ZeroPadding2D((31,0)) + AveragePooling1D(32, strides=1) over time)."""
    import torch
    import torch.nn.functional as Fn
    ref = NutlsRef(batch=1, ctfa_mode="causal32")
    rng = np.random.default_rng(5)
    T_, F = 75, 8
    xs = torch.from_numpy(rng.standard_normal((T_, F, 64)).astype(np.float32))
    e0 = torch.from_numpy(rng.standard_normal((T_, F, 64)).astype(np.float32))
    prefix = "msfe3_en"
    got = torch.stack([ref._ctfa(xs[t:t + 1], e0[t:t + 1], prefix)[0] for t in range(T_)])
    ta = ref._mlp_gate(xs.mean(dim=1), prefix + "_ta")                                   # [T,64]
    pooled = Fn.avg_pool1d(Fn.pad(ta.t().unsqueeze(0), (31, 0)), 32, stride=1)[0].t()     # [T,64] causal mean of 32
    fa = ref._mlp_gate(pooled, prefix + "_fa")
    want = xs * (ta * fa).unsqueeze(1) + e0
    assert rms(got.numpy(), want.numpy()) < 1e-6
    frame = NutlsRef(batch=1)                       # the streaming form differs from it after the first frame only
    f0 = frame._ctfa(xs[0:1], e0[0:1], prefix)[0]
    assert rms(f0.numpy(), want[0].numpy()) < 1e-7
