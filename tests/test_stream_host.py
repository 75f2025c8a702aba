"""Host-loop plumbing (CPU): windows against the phone app's hard-coded tables, framing,
and the end-to-end SNR of config 1 with the oracle standing in for the model."""
import json
import os

import numpy as np
import pytest

from nunet_amd import stream_enhance as SE
from oracle.nutls_ref import NutlsRef

from conftest import GOLDEN


def test_windows_match_java_tables():
    # known answers: mobile_app/.../RTSE_NUTLS_LSTM.java:62-63 (tf.signal windows, float32)
    tabs = json.load(open(os.path.join(GOLDEN, "windows.json")))
    w, inv = SE.analysis_window(), SE.inverse_window()
    np.testing.assert_allclose(w[tabs["window"]["index"]], tabs["window"]["value"], atol=2e-7)
    np.testing.assert_allclose(inv[tabs["inverse_window"]["index"]], tabs["inverse_window"]["value"], atol=3e-7, rtol=3e-6)
    assert abs(float(inv.max()) - tabs["inverse_window"]["max"]) < 1e-6
    assert inv[128] == 1.0 and w[0] == np.float32(1e-7) and w[511] == np.float32(1e-7)


def test_framing_matches_golden_magnitudes():
    clip = np.load(os.path.join(GOLDEN, "clip_4s.npz"))
    audio = clip["noisy_i16"].astype(np.float64) / 32768.0
    mags, phases = SE.frame_magnitudes(audio)
    assert mags.shape == (249, 257)     # (64000-256)//256, interpreter_proposed.py:32
    np.testing.assert_allclose(mags[:, 1:].astype(np.float32), clip["mags_in"], atol=1e-6)


def test_config1_clip_snr_with_oracle_runner():
    """BASELINE config 1: 4 s clip, batch 1, CPU plumbing.  SNR 0.76 -> 11.63 dB."""
    clip = np.load(os.path.join(GOLDEN, "clip_4s.npz"))
    audio = clip["noisy_i16"].astype(np.float64) / 32768.0
    clean = clip["clean_i16"].astype(np.float64) / 32768.0
    ref = NutlsRef(batch=1)
    enh, times = SE.real_time_speech_enhancer(audio, ref.signature_call)
    assert len(times) == 249 and len(enh) == 64000
    n = 248 * 256
    assert abs(SE.snr_db(clean[:n], audio[:n]) - 0.76) < 0.05
    assert abs(SE.snr_db(clean[:n], enh[:n]) - 11.63) < 0.05
    assert abs(SE.si_snr_db(clean[:n], enh[:n]) - 13.28) < 0.05
    assert float(np.max(np.abs(enh[:n] - clip["enhanced"][:n]))) < 1e-5


def test_device_loop_validates_its_arguments_without_a_gpu():
    """enhance_batch_on_device is host plumbing around engine.enhance_hop: shape checks and the output alignment
    can be exercised with a stand-in engine (identity on every hop)."""
    class Identity:
        batch = 2

        def enhance_hop(self, hop, dc_mode="edge"):
            assert hop.shape == (2, SE.FRAME_STEP) and hop.dtype == np.float32 and hop.flags["C_CONTIGUOUS"]
            return hop

    audio = np.arange(2 * 1280, dtype=np.float32).reshape(2, 1280)
    out = SE.enhance_batch_on_device(audio, Identity())
    assert out.shape == (2, 1280)
    # 4 hops are processed ((1280 - 256) // 256); the first 256 output samples are dropped like the reference does
    np.testing.assert_array_equal(out[:, :768], audio[:, 256:1024])
    assert not out[:, 768:].any()
    with pytest.raises(ValueError):
        SE.enhance_batch_on_device(audio[:1], Identity())
