"""GPU: the causal 32-frame CTFA (the offline / training model's attention, /root/reference/dnn_model/models/proposed.py:125-160,
:143-147) in the STREAMING fused kernel -- a non-default mode (the reference's streaming graph feeds the frequency branch TA / 32,
proposed.py:179-183 with T = 1; that stays the default).  Checked against oracle B in the same mode, against the offline
handle in the same mode, on the one-stream plan and on the packed plans; history semantics (reset, mode switch)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN
from nunet_amd import NutlsEngine, NutlsOffline
from oracle.nutls_ref import NutlsRef

pytestmark = pytest.mark.gpu


def rms(a, b):
    return float(np.sqrt(np.mean((np.asarray(a, np.float64) - np.asarray(b, np.float64)) ** 2)))


@pytest.fixture(scope="module")
def clip():
    return np.load(os.path.join(GOLDEN, "clip_4s.npz"))


@pytest.fixture(scope="module")
def causal32_oracle_run(clip):
    """The oracle's side of the test below, computed once for the three plans: 80 frames of four streams in causal32 mode."""
    frames = clip["mags_in"]
    B, n = 4, 80
    ref = NutlsRef(batch=B, ctfa_mode="causal32")
    xs = [np.stack([frames[(i + 40 * s) % 249] for s in range(B)]) for i in range(n)]
    return xs, [ref.step(x).numpy() for x in xs]


@pytest.mark.parametrize("spw", [1, 2, 4])
def test_streaming_causal32_matches_oracle_and_offline(clip, causal32_oracle_run, spw):
    """80 frames (the history ring wraps at 32 and 64): four streams of the real clip at different offsets vs oracle B with
    ctfa_mode="causal32" (<= 2e-5 RMS), stream 0 vs the offline handle in the same mode; and the mode matters (the default frame mode
    gives a different output on the same input)."""
    frames = clip["mags_in"]
    xs, wants = causal32_oracle_run
    B, n = 4, len(xs)
    eng = NutlsEngine(batch=B, streams_per_workgroup=spw)
    assert eng.streams_per_workgroup == spw
    eng.set_ctfa_mode("causal32")
    frame_mode = NutlsEngine(batch=B, streams_per_workgroup=spw)
    outs, diff_to_frame_mode = [], 0.0
    for i in range(n):
        out = eng.step(xs[i])
        assert rms(out, wants[i]) < 2e-5, i
        diff_to_frame_mode = max(diff_to_frame_mode, rms(out, frame_mode.step(xs[i])))
        outs.append(out[0])
    assert diff_to_frame_mode > 1e-4          # not the same function as the default mode
    off = NutlsOffline(max_frames=32, ctfa_mode="causal32")
    got = off.process(np.stack([frames[i % 249] for i in range(n)]))
    assert rms(np.stack(outs), got) < 2e-5
    for e in (eng, frame_mode, off):
        e.close()


def test_causal32_history_follows_reset_and_mode_switches(clip):
    """nutls_reset(stream) clears that stream's history only; switching the mode clears everyone's; the default is the frame mode;
    the mode needs the fused kernel."""
    frames = clip["mags_in"]
    eng = NutlsEngine(batch=2)
    eng.set_ctfa_mode("causal32")
    for i in range(40):
        eng.step(np.stack([frames[i], frames[i + 50]]))
    eng.reset(1)
    fresh = NutlsEngine(batch=1)
    fresh.set_ctfa_mode("causal32")
    for i in range(12):          # stream 1 restarted from nothing: equal to a new handle fed the same frames
        a = eng.step(np.stack([frames[40 + i], frames[i]]))
        b = fresh.step(frames[i:i + 1])
        assert rms(a[1], b[0]) < 1e-6, i
    # leaving and re-entering the mode starts a new history (state tensors stay): same as a handle whose history was cleared
    eng.set_ctfa_mode("frame")
    eng.set_ctfa_mode("causal32")
    with pytest.raises(ValueError):
        eng.set_mode("graph")          # the per-layer kernels have no causal32 CTFA for streaming handles
    eng.set_ctfa_mode("frame")
    eng.set_mode("graph")
    with pytest.raises(ValueError):
        eng.set_ctfa_mode("causal32")
    eng.close()
    fresh.close()


def test_causal32_state_set_leaves_the_other_streams_history_alone(clip):
    """The per-stream workflow of INTEGRATION.md -- state_get, change ONE stream's row, state_set -- must not disturb the live streams
    beside it: nutls_state_set takes [B, ...] buffers and does not touch the time-attention history (include/nutls.h, nutls_state_set).
    Stream 0 keeps running bit-identically to an untouched twin; stream 1, reset and re-loaded, equals a fresh handle."""
    frames = clip["mags_in"]
    eng, twin = NutlsEngine(batch=2), NutlsEngine(batch=2)
    for e in (eng, twin):
        e.set_ctfa_mode("causal32")
    for i in range(40):
        x = np.stack([frames[i], frames[i + 50]])
        eng.step(x)
        twin.step(x)
    # a new utterance moves into stream 1: reset that stream, then write its (zero) rows through the [B, ...] accessor
    eng.reset(1)
    for name in ("msfe6_ee_prev1", "msfe4_de_h", "state_c"):
        a = eng.state_get(name)
        a[1] = 0.0
        eng.state_set(name, a)
    fresh = NutlsEngine(batch=1)
    fresh.set_ctfa_mode("causal32")
    for i in range(34):          # (past one wrap of stream 0's history ring)
        a = eng.step(np.stack([frames[40 + i], frames[i]]))
        b = twin.step(np.stack([frames[40 + i], frames[i]]))
        c = fresh.step(frames[i:i + 1])
        assert rms(a[0], b[0]) < 2e-6, i          # (the carried sums are rebuilt after a state_set: 1e-7, not bit for bit)
        assert rms(a[1], c[0]) < 2e-6, i
    for e in (eng, twin, fresh):
        e.close()
