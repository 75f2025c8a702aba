"""Streaming host loop: ring buffer -> window -> rFFT -> |X|, angle -> MODEL -> DC pad ->
irFFT -> inverse window -> overlap-add.

Counterpart of ``real_time_speech_enhancer`` in
``/root/reference/dnn_model/interpreter_proposed.py:15-370`` (same framing, same
windows, same DC handling), written against a *runner* object with the
reference's signature-runner call surface (``runner(input=..., <130 state
kwargs>) -> dict``), e.g. :class:`nunet_amd.runner.NutlsRunner`.  The model call
is the only heavy part and runs on the GPU; this loop is plumbing (numpy).
"""
from __future__ import annotations

import time
from typing import Callable, Dict, List, Tuple

import numpy as np

from . import topology as T

FRAME_LEN = 512    # interpreter_proposed.py:17
FRAME_STEP = 256   # interpreter_proposed.py:18
SAMPLE_RATE = 16000
HOP_SECONDS = FRAME_STEP / SAMPLE_RATE   # 0.016 s, the RTF denominator (interpreter_proposed.py:412)


def hann_periodic(n: int = FRAME_LEN) -> np.ndarray:
    """``tf.signal.hann_window(n)`` (periodic=True).  Evaluated in float32 like TF does, so
    the taps agree with the tables the reference's phone app hard-codes
    (mobile_app/.../RTSE_NUTLS_LSTM.java:62) to 1e-7 (e.g. w[1] = 3.76403e-05, where the
    float64 value would be 3.76491e-05)."""
    k = np.arange(n, dtype=np.float32)
    arg = np.float32(2.0 * np.pi) * k / np.float32(n)
    return (np.float32(0.5) - np.float32(0.5) * np.cos(arg)).astype(np.float32)


def analysis_window() -> np.ndarray:
    """Hann with both end taps forced to 1e-7 (interpreter_proposed.py:20-22)."""
    w = hann_periodic()
    w[0], w[-1] = 1e-7, 1e-7
    return w


def inverse_window() -> np.ndarray:
    """``tf.signal.inverse_stft_window_fn(256, hann_window)(512)``
    (interpreter_proposed.py:24-26): w / (w^2 + w_shifted_by_hop^2)."""
    w = hann_periodic().astype(np.float32)
    den = np.square(w).reshape(FRAME_LEN // FRAME_STEP, FRAME_STEP).sum(axis=0, keepdims=True)
    den = np.tile(den, (FRAME_LEN // FRAME_STEP, 1)).reshape(-1)
    return (w / den).astype(np.float32)


def zero_state(variant: str = "lstm") -> Dict[str, np.ndarray]:
    """The all-zero ``tflite_out`` seed of interpreter_proposed.py:36-198 / interpreter_nunet_tls.py:36-370 (keys are
    the *output* names: ``*_cur{i}``, ``*_h``, ``*_c``, ``model_out``; the baseline's dilated-dense histories are
    ``[1, d, F, C]``)."""
    st = {"model_out": np.zeros((1, 1, T.N_BINS, 1), np.float32)}
    for base, shp in T.state_specs(variant):
        if len(shp) == 1:
            st[base] = np.zeros((1, shp[0]), np.float32)
        else:
            st[base.format("cur")] = np.zeros((1,) + shp, np.float32)
    return st


def feeds_from_outputs(prev_out: Dict[str, np.ndarray], sliced_mag: np.ndarray, variant: str = "lstm") -> Dict[str, np.ndarray]:
    """cur -> prev echo the caller performs every frame (interpreter_proposed.py:215-350,
    interpreter_nunet_tls.py:372-540 for the 'nutls' signature)."""
    feeds = {"input": sliced_mag}
    for base, shp in T.state_specs(variant):
        if len(shp) == 1:
            feeds[base] = prev_out[base]
        else:
            feeds[base.format("prev")] = prev_out[base.format("cur")]
    return feeds


def hop_frames(noisy_speech: np.ndarray) -> np.ndarray:
    """The loop's analysis buffers, all at once: frame ``i`` = hop ``i - 1`` (zeros for the first) followed by hop ``i``
    -- what the reference's shifting ``in_buffer`` holds at step ``i`` (interpreter_proposed.py:203-205).  ``[n, 512]``
    float32 view-copy; ``n = (len - 256) // 256`` (:32)."""
    audio = np.asarray(noisy_speech, np.float32)
    n = (audio.shape[0] - (FRAME_LEN - FRAME_STEP)) // FRAME_STEP
    padded = np.concatenate([np.zeros(FRAME_LEN - FRAME_STEP, np.float32), audio[:n * FRAME_STEP]])
    if n <= 0:
        return np.zeros((0, FRAME_LEN), np.float32)
    return np.lib.stride_tricks.sliding_window_view(padded, FRAME_LEN)[::FRAME_STEP][:n]


def frame_magnitudes(noisy_speech: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """(|X| [n,257], angle [n,257]) of every analysis frame: float32 windowed buffer, float64 transform like
    ``np.fft.rfft`` in the reference loop (interpreter_proposed.py:206-210)."""
    spec = np.fft.rfft(hop_frames(noisy_speech) * analysis_window(), axis=-1)
    return np.abs(spec), np.angle(spec)


def overlap_add(blocks: np.ndarray, total: int) -> np.ndarray:
    """Synthesis blocks ``[n, 512]`` (float32, already inverse-windowed) -> the loop's ``out_file`` of ``total`` samples:
    output hop ``i`` = first half of block ``i`` + second half of block ``i - 1``, summed in float32 as the reference's
    ``out_buffer`` does (interpreter_proposed.py:359-365)."""
    n = blocks.shape[0]
    out = np.zeros((total,))
    if n:
        seg = blocks[:, :FRAME_STEP].copy()
        seg[1:] += blocks[:-1, FRAME_STEP:]
        out[:n * FRAME_STEP] = seg.reshape(-1)
    return out


def real_time_speech_enhancer(noisy_speech: np.ndarray, runner: Callable[..., Dict[str, np.ndarray]],
                              dc_mode: str = "edge", variant: str = None) -> Tuple[np.ndarray, List[float]]:
    """Frame-by-frame enhancement of one utterance; returns (waveform, seconds per frame).  The seconds cover what the
    reference's ``time_array`` covers (interpreter_proposed.py:201-366: the whole loop body -- buffer shift and rfft, model call,
    irfft, overlap-add): the frame's model call + synthesis as measured, plus its share of the analysis and of the overlap-add,
    which run vectorised outside the loop here and are timed as a whole and dealt evenly to the frames.  Same function as the reference's loop (interpreter_proposed.py:15-370), organised around the one
    thing that has to be sequential -- the model call, whose state feeds the next frame: analysis of all frames up front
    (:func:`frame_magnitudes`), per frame the runner call + inverse transform, overlap-add at the end (:func:`overlap_add`).

    ``variant``: which signature the runner implements -- ``"lstm"`` ('nutls_lstm_sm', interpreter_proposed.py:380)
    or ``"baseline"`` ('nutls', interpreter_nunet_tls.py:549); default: what the runner says (``signature_key``).

    ``dc_mode``: how bin 0 is re-created after the 256-bin model: ``"edge"`` (the PC loop,
    interpreter_proposed.py:352-353: bin 0 = bin 1) or ``"zero"`` (the phone, mobile_app/.../RTSE_NUTLS_LSTM.java:677)."""
    if dc_mode not in ("edge", "zero"):
        raise ValueError("dc_mode must be 'edge' or 'zero'")
    if variant is None:
        variant = "baseline" if getattr(runner, "signature_key", "nutls_lstm_sm") == "nutls" else "lstm"
    audio = np.asarray(noisy_speech)
    t_pre = time.time()
    mags, phases = frame_magnitudes(audio)
    rotors = np.exp(1j * phases)
    t_pre = time.time() - t_pre
    inv_win = inverse_window()
    blocks = np.zeros((mags.shape[0], FRAME_LEN), np.float32)
    seconds: List[float] = []
    outputs = zero_state(variant)
    for i in range(mags.shape[0]):
        t0 = time.time()
        model_in = mags[i, 1:].astype(np.float32).reshape(1, 1, T.N_BINS, 1)
        outputs = runner(**feeds_from_outputs(outputs, model_in, variant))
        est = np.empty(FRAME_LEN // 2 + 1)
        est[1:] = outputs["model_out"].reshape(-1)
        est[0] = est[1] if dc_mode == "edge" else 0.0
        blocks[i] = np.fft.irfft(est * rotors[i]).astype(np.float32) * inv_win
        seconds.append(time.time() - t0)
    total = len(audio) + (FRAME_LEN - FRAME_STEP)
    t_post = time.time()
    wave = overlap_add(blocks, total)[FRAME_LEN - FRAME_STEP:]
    t_post = time.time() - t_post
    if seconds:
        share = (t_pre + t_post) / len(seconds)
        seconds = [t + share for t in seconds]
    return wave, seconds


def enhance_batch_on_device(noisy: np.ndarray, engine, dc_mode: str = "edge") -> np.ndarray:
    """The same loop with the STFT / inverse STFT on the GPU too: ``noisy [B, N]`` (one utterance per stream,
    equal lengths) -> enhanced ``[B, N]`` in the alignment of :func:`real_time_speech_enhancer` (whose first
    256 output samples are dropped, interpreter_proposed.py:368).  Only PCM hops cross the host boundary:
    ``engine.enhance_hop`` = ``nutls_enhance_hop_host`` of the C ABI."""
    audio = np.asarray(noisy, np.float32)
    if audio.ndim == 1:
        audio = audio[None]
    if audio.shape[0] != engine.batch:
        raise ValueError("need one utterance per stream: %d != %d" % (audio.shape[0], engine.batch))
    n = audio.shape[1]
    num_blocks = (n - (FRAME_LEN - FRAME_STEP)) // FRAME_STEP
    out = np.zeros((audio.shape[0], n + (FRAME_LEN - FRAME_STEP)), np.float64)
    for idx in range(num_blocks):
        hop = np.ascontiguousarray(audio[:, idx * FRAME_STEP:(idx + 1) * FRAME_STEP])
        out[:, idx * FRAME_STEP:(idx + 1) * FRAME_STEP] = engine.enhance_hop(hop, dc_mode)
    return out[:, FRAME_LEN - FRAME_STEP:]


def snr_db(clean: np.ndarray, est: np.ndarray) -> float:
    n = min(len(clean), len(est))
    c, e = np.asarray(clean[:n], np.float64), np.asarray(est[:n], np.float64)
    return float(10.0 * np.log10(np.sum(c * c) / np.sum((c - e) ** 2)))


def si_snr_db(clean: np.ndarray, est: np.ndarray) -> float:
    n = min(len(clean), len(est))
    c, e = np.asarray(clean[:n], np.float64), np.asarray(est[:n], np.float64)
    c, e = c - c.mean(), e - e.mean()
    s = (np.dot(e, c) / np.dot(c, c)) * c
    return float(10.0 * np.log10(np.sum(s * s) / np.sum((e - s) ** 2)))
