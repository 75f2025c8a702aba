"""``-m gpu``: STFT front end / inverse-STFT + overlap-add back end on the device,
called through the C ABI, against the numpy restatement of the reference's host loop
(``nunet_amd.stream_enhance`` <- ``dnn_model/interpreter_proposed.py:15-370``) and the committed golden
clip (magnitudes and waveform produced by that loop around oracle A)."""
import os

import numpy as np
import pytest

from nunet_amd import NutlsEngine, stream_enhance as SE

from conftest import GOLDEN

pytestmark = pytest.mark.gpu

HOP = SE.FRAME_STEP


@pytest.fixture(scope="module")
def clip():
    return np.load(os.path.join(GOLDEN, "clip_4s.npz"))


def hops_of(audio):
    n = (len(audio) - (SE.FRAME_LEN - HOP)) // HOP
    return [np.ascontiguousarray(audio[i * HOP:(i + 1) * HOP], dtype=np.float32) for i in range(n)]


def test_stft_hop_matches_host_loop_and_golden_magnitudes(clip):
    """|X| bins 1..256 and the phase of every frame of the clip: device FFT vs np.fft.rfft of the same
    windowed buffer, and vs the golden model inputs."""
    import torch
    audio = (clip["noisy_i16"].astype(np.float64) / 32768.0).astype(np.float32)
    mags, phases = SE.frame_magnitudes(audio)
    eng = NutlsEngine(batch=1)
    scale = float(np.abs(mags).max())
    for i, hop in enumerate(hops_of(audio)):
        eng.stft_hop(torch.from_numpy(hop[None]).cuda())
        got = eng.debug_get("mag_in", (256,))[0]
        assert np.abs(got - mags[i, 1:]).max() < 2e-6 * scale, i
        assert np.abs(got - clip["mags_in"][i]).max() < 2e-6 * scale, i
        ph = eng.debug_get("phasor", (257, 2))[0]
        ref = np.exp(1j * phases[i])
        strong = mags[i] > 1e-3 * scale          # the phase of a numerically empty bin is noise on both sides
        assert np.abs((ph[:, 0] + 1j * ph[:, 1]) - ref)[strong].max() < 2e-4, i
    eng.close()


def _hip():
    import ctypes
    for name in ("libamdhip64.so", "/opt/rocm/lib/libamdhip64.so"):
        try:
            return ctypes.CDLL(name)
        except OSError:
            continue
    pytest.skip("libamdhip64.so not loadable through ctypes")


def test_identity_model_reconstructs_the_input_one_hop_late():
    """Size-independent property: window * inverse window overlap-adds to one, so stft -> (mag_out := mag_in)
    -> istft returns the input delayed by one hop.  The signal is a sum of bin-centred sinusoids far from DC
    (bins 16..200), so the bin the 256-bin model never sees carries nothing (Hann side lobes < -90 dB there)."""
    import ctypes
    import torch
    rng = np.random.default_rng(7)
    B, n_hops = 5, 12
    n = np.arange(HOP * n_hops)
    x = np.zeros((B, n.size))
    for b in range(B):
        for k in rng.choice(np.arange(16, 201), size=24, replace=False):
            x[b] += rng.uniform(0.2, 1.0) * np.cos(2 * np.pi * k * n / SE.FRAME_LEN + rng.uniform(0, 2 * np.pi))
    x = x.astype(np.float32)
    eng = NutlsEngine(batch=B)
    hip = _hip()
    out = []
    y = torch.empty(B, HOP, device="cuda")
    for i in range(n_hops):
        eng.stft_hop(torch.from_numpy(np.ascontiguousarray(x[:, i * HOP:(i + 1) * HOP])).cuda())
        torch.cuda.synchronize()
        # identity "model": copy the analysis magnitudes to the model-output buffer (hipMemcpyDeviceToDevice = 3)
        assert hip.hipMemcpy(ctypes.c_void_p(eng.io_out_ptr), ctypes.c_void_p(eng.io_in_ptr), ctypes.c_size_t(B * 256 * 4), 3) == 0
        eng.istft_hop(y, "zero")
        out.append(y.cpu().numpy().copy())
    got = np.concatenate(out, axis=1)
    # output hop i = input hop i-1; hop 0 of the input only ever sees the rising half window (its partner frame
    # holds zeros), so compare from input hop 1 on
    ref = x[:, HOP:-HOP]
    err = got[:, 2 * HOP:] - ref
    assert np.sqrt(np.mean(err ** 2)) < 2e-5 * np.sqrt(np.mean(ref ** 2))
    eng.close()


def test_device_pipeline_matches_host_loop_waveform(clip):
    """Whole pipeline on the device (one hop per host call) vs the numpy loop around the same engine,
    vs the golden enhanced waveform, and the config-1 quality numbers (SNR 0.76 -> 11.63 dB)."""
    audio = (clip["noisy_i16"].astype(np.float64) / 32768.0).astype(np.float32)
    eng = NutlsEngine(batch=1)
    dev = SE.enhance_batch_on_device(audio[None], eng)[0]
    eng.close()
    gold = clip["enhanced"].astype(np.float64)
    n = min(len(dev), len(gold))
    scale = np.sqrt(np.mean(gold[:n] ** 2))
    assert np.sqrt(np.mean((dev[:n] - gold[:n]) ** 2)) < 1e-4 * scale
    clean = clip["clean_i16"].astype(np.float64) / 32768.0
    n = 248 * 256                      # the samples the 249 frames fully cover (make_golden.py)
    assert abs(SE.snr_db(clean[:n], dev[:n]) - float(clip["snr_after"])) < 0.05
    assert abs(SE.si_snr_db(clean[:n], dev[:n]) - float(clip["sisnr_after"])) < 0.05


def test_batched_streams_are_independent_and_reset_clears_the_tails(clip):
    audio = (clip["noisy_i16"].astype(np.float64) / 32768.0).astype(np.float32)[:HOP * 20]
    B = 3
    batch = np.stack([audio, 0.5 * audio, audio[::-1].copy()])
    eng = NutlsEngine(batch=B)
    ref = SE.enhance_batch_on_device(batch, eng)
    eng.reset()
    again = SE.enhance_batch_on_device(batch, eng)
    np.testing.assert_array_equal(ref, again)                 # reset restores the all-zero start exactly
    solo = NutlsEngine(batch=1)
    one = SE.enhance_batch_on_device(batch[2:3], solo)[0]
    solo.close()
    eng.close()
    assert np.sqrt(np.mean((one - ref[2]) ** 2)) < 1e-5 * np.sqrt(np.mean(one ** 2))


def test_dc_modes_and_bad_arguments(clip):
    audio = (clip["noisy_i16"].astype(np.float64) / 32768.0).astype(np.float32)[:HOP * 6]
    eng = NutlsEngine(batch=1)
    edge = SE.enhance_batch_on_device(audio[None], eng, "edge")
    eng.reset()
    zero = SE.enhance_batch_on_device(audio[None], eng, "zero")
    assert np.abs(edge - zero).max() > 0            # the DC bin differs
    with pytest.raises(ValueError):
        eng.enhance_hop(np.zeros((1, 255), np.float32))
    with pytest.raises(ValueError):
        eng.enhance_hop(np.zeros((1, 256), np.float32), "mirror")
    eng.close()


def test_quality_harness_on_a_directory_of_wav_pairs(clip, tmp_path):
    """the directory harness (wav pairs in, SNR / SI-SNR out, whole loop on the GPU) reproduces the
    golden clip's numbers; a second, shorter pair rides in the same batch."""
    from scipy.io import wavfile
    from nunet_amd.evaluate import evaluate_directory, find_pairs
    wavfile.write(str(tmp_path / "40hc020i_0.wav"), 16000, clip["noisy_i16"])
    wavfile.write(str(tmp_path / "40hc020i.wav"), 16000, clip["clean_i16"])
    wavfile.write(str(tmp_path / "short_0.wav"), 16000, clip["noisy_i16"][:20000])
    wavfile.write(str(tmp_path / "short.wav"), 16000, clip["clean_i16"][:20000])
    wavfile.write(str(tmp_path / "orphan_0.wav"), 16000, clip["noisy_i16"][:4000])        # no clean partner: ignored
    assert [p["name"] for p in find_pairs(str(tmp_path))] == ["40hc020i", "short"]
    rows = {r["name"]: r for r in evaluate_directory(str(tmp_path), out_dir=str(tmp_path / "out"))}
    full = rows["40hc020i"]
    assert abs(full["snr_before"] - float(clip["snr_before"])) < 0.05 and abs(full["snr_after"] - float(clip["snr_after"])) < 0.05
    assert abs(full["sisnr_after"] - float(clip["sisnr_after"])) < 0.05
    assert rows["short"]["snr_after"] > rows["short"]["snr_before"] + 5.0
    assert (tmp_path / "out" / "short_enhanced.wav").exists()
