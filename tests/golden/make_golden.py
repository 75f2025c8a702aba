#!/usr/bin/env python3
"""Generate the committed golden vectors from the reference's own shipped artefacts.

Run in the build container only (needs /root/reference):

    python tests/golden/make_golden.py

Everything here is produced by **oracle A** (``oracle/graph_exec.py``): a float32
op-by-op execution of ``/root/reference/dnn_model/tflite/nutls_lstm.tflite`` driven
through the reference's host loop semantics
(``/root/reference/dnn_model/interpreter_proposed.py:200-367``) on the reference's
sample ``/root/reference/dnn_model/data/40hc020i_0.wav`` (clean: ``40hc020i.wav``).
TensorFlow / TF-Lite cannot be installed here, so these are the strongest
reference-derived vectors available.

Outputs (tests/golden/):
  kat_b1.npz        closed-form input x[k]=0.5+0.5 sin(0.1k), 3 frames from zero state:
                    model_out per frame + selected state sums
  clip_4s.npz       first 64 000 samples of the noisy/clean pair (int16), the 249 input
                    magnitude frames [249,256], oracle-A outputs [249,256], the enhanced
                    waveform and its SNR / SI-SNR
  state_f3.npz      all 130 state tensors after frame 3 of the clip  (per-layer intermediates:
  state_f64.npz     ... and after frame 64                            every conv input is a state)
  windows.json      a few taps of the Java window tables (known answers for the STFT step)
"""
from __future__ import annotations

import json
import os
import re
import sys

import numpy as np
from scipy.io import wavfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import nunet_amd  # noqa: E402,F401
from nunet_amd import stream_enhance as SE, topology as T  # noqa: E402
from oracle.graph_exec import GraphOracle  # noqa: E402

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def state_dict(out):
    d = {}
    for base, shp in T.state_specs():
        k = base if len(shp) == 1 else base.format("cur")
        d[k] = np.asarray(out[k], np.float32)
    return d


def main():
    g = GraphOracle(os.path.join(REF, "dnn_model/tflite/nutls_lstm.tflite"))

    # ---- B.1 closed-form known-answer test -------------------------------------------------
    x = (0.5 + 0.5 * np.sin(0.1 * np.arange(256))).astype(np.float32)
    out = SE.zero_state()
    kat = {"x": x}
    for fr in range(3):
        out = g(**SE.feeds_from_outputs(out, x.reshape(1, 1, 256, 1)))
        kat["out%d" % (fr + 1)] = out["model_out"].reshape(256).astype(np.float32)
        kat["state_h%d" % (fr + 1)] = out["state_h"].reshape(-1).astype(np.float32)
        kat["ee_cur2_sum%d" % (fr + 1)] = np.float32(out["msfe6_ee_cur2"].sum())
    np.savez_compressed(os.path.join(HERE, "kat_b1.npz"), **kat)

    # ---- real clip ----------------------------------------------------------------------------
    fs, noisy = wavfile.read(os.path.join(REF, "dnn_model/data/40hc020i_0.wav"))
    fs2, clean = wavfile.read(os.path.join(REF, "dnn_model/data/40hc020i.wav"))
    assert fs == fs2 == 16000 and noisy.dtype == np.int16
    noisy, clean = noisy[:64000], clean[:64000]
    audio = noisy.astype(np.float64) / 32768.0   # what soundfile.read returns
    mags, _ = SE.frame_magnitudes(audio)
    frames = {}

    n_frames = [0]
    outs = []

    def runner(**feeds):
        o = g(**feeds)
        n_frames[0] += 1
        outs.append(o["model_out"].reshape(256).astype(np.float32))
        if n_frames[0] in (3, 64):
            frames[n_frames[0]] = state_dict(o)
        return o

    enh, _ = SE.real_time_speech_enhancer(audio, runner)
    outs = np.stack(outs)
    c = clean.astype(np.float64) / 32768.0
    n = (len(outs) - 1) * 256   # samples the loop actually produced (249 hops, first one trimmed)
    res = dict(
        noisy_i16=noisy, clean_i16=clean,
        mags_in=mags[:, 1:].astype(np.float32), mags_out=outs,
        enhanced=enh.astype(np.float32),
        snr_before=np.float64(SE.snr_db(c[:n], audio[:n])), snr_after=np.float64(SE.snr_db(c[:n], enh[:n])),
        sisnr_before=np.float64(SE.si_snr_db(c[:n], audio[:n])), sisnr_after=np.float64(SE.si_snr_db(c[:n], enh[:n])),
    )
    np.savez_compressed(os.path.join(HERE, "clip_4s.npz"), **res)
    for k, st in frames.items():
        np.savez_compressed(os.path.join(HERE, "state_f%d.npz" % k), **st)
    print("clip: frames", outs.shape, "SNR %.2f -> %.2f dB, SI-SNR %.2f -> %.2f dB" % (
        res["snr_before"], res["snr_after"], res["sisnr_before"], res["sisnr_after"]))
    print("in rms %.4f max %.3f | out rms %.4f max %.3f min %.4f neg %.2f%%" % (
        np.sqrt((res["mags_in"] ** 2).mean()), res["mags_in"].max(), np.sqrt((outs ** 2).mean()),
        outs.max(), outs.min(), 100 * (outs < 0).mean()))

    # ---- window tables (known answers from the phone app) -------------------------------------
    java = None
    for dirpath, _, files in os.walk(os.path.join(REF, "mobile_app")):
        if "RTSE_NUTLS_LSTM.java" in files:
            java = os.path.join(dirpath, "RTSE_NUTLS_LSTM.java")
    lines = open(java).read().split("\n")
    tabs = {}
    for key, ln in (("window", 61), ("inverse_window", 62)):
        nums = re.findall(r"[-+]?\d+\.\d+(?:[eE][-+]?\d+)?|[-+]?\d+(?:[eE][-+]?\d+)", lines[ln].split("{", 1)[1])
        vals = [float(v) for v in nums]
        assert len(vals) == 512
        idx = [0, 1, 2, 3, 64, 127, 128, 129, 255, 256, 257, 300, 384, 509, 510, 511]
        tabs[key] = {"index": idx, "value": [vals[i] for i in idx], "max": max(vals), "sum": float(np.sum(vals))}
    json.dump(tabs, open(os.path.join(HERE, "windows.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
