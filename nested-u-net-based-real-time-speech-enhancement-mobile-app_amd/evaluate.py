"""Quality harness: enhance every ``<name>_0.wav`` (noisy) of a directory that has a clean
partner ``<name>.wav`` -- the layout of the reference's ``dnn_model/data`` -- with the whole loop on the GPU
(:func:`nunet_amd.stream_enhance.enhance_batch_on_device`) and report SNR / SI-SNR before and after.

The reference's own harness (``dnn_model/test_interface.py:52-110``) reports PESQ / STOI through the ``pesq`` /
``pystoi`` packages, which are not available here; SNR and SI-SNR are the measures this repository pins its
golden clip with (0.76 -> 11.63 dB, 0.74 -> 13.28 dB on ``40hc020i``).
"""
from __future__ import annotations

import glob
import os
from typing import Dict, List, Optional

import numpy as np

from . import stream_enhance as SE


def find_pairs(directory: str) -> List[Dict[str, str]]:
    pairs = []
    for noisy in sorted(glob.glob(os.path.join(directory, "*_0.wav"))):
        clean = noisy[:-len("_0.wav")] + ".wav"
        if os.path.exists(clean):
            pairs.append({"name": os.path.basename(clean)[:-4], "noisy": noisy, "clean": clean})
    return pairs


def _read(path: str) -> np.ndarray:
    from scipy.io import wavfile
    fs, x = wavfile.read(path)
    if fs != SE.SAMPLE_RATE:
        raise ValueError("%s: %d Hz, the model runs at %d Hz" % (path, fs, SE.SAMPLE_RATE))
    if x.ndim > 1:
        x = x[:, 0]
    if np.issubdtype(x.dtype, np.integer):
        return x.astype(np.float64) / float(np.iinfo(x.dtype).max + 1)
    return x.astype(np.float64)


def evaluate_directory(directory: str, out_dir: Optional[str] = None, dc_mode: str = "edge", device: int = 0) -> List[Dict[str, float]]:
    """All pairs of ``directory`` in ONE batched engine (one stream per clip; shorter clips are zero-padded to the
    longest and scored on their own length).  Returns one dict per clip; optionally writes ``<name>_enhanced.wav``."""
    from .runner import NutlsEngine
    pairs = find_pairs(directory)
    if not pairs:
        raise ValueError("no <name>_0.wav / <name>.wav pairs in %s" % directory)
    noisy = [_read(p["noisy"]) for p in pairs]
    clean = [_read(p["clean"]) for p in pairs]
    n_max = max(len(x) for x in noisy)
    batch = np.zeros((len(pairs), n_max), np.float32)
    for i, x in enumerate(noisy):
        batch[i, :len(x)] = x
    eng = NutlsEngine(batch=len(pairs), device=device)
    try:
        enhanced = SE.enhance_batch_on_device(batch, eng, dc_mode)
    finally:
        eng.close()
    rows = []
    for i, p in enumerate(pairs):
        n = min(len(noisy[i]), len(clean[i]))
        n = ((n - (SE.FRAME_LEN - SE.FRAME_STEP)) // SE.FRAME_STEP - 1) * SE.FRAME_STEP      # samples the frames fully cover
        c, x, y = clean[i][:n], noisy[i][:n], enhanced[i][:n]
        rows.append({"name": p["name"], "seconds": n / SE.SAMPLE_RATE,
                     "snr_before": SE.snr_db(c, x), "snr_after": SE.snr_db(c, y),
                     "sisnr_before": SE.si_snr_db(c, x), "sisnr_after": SE.si_snr_db(c, y)})
        if out_dir:
            from scipy.io import wavfile
            os.makedirs(out_dir, exist_ok=True)
            wavfile.write(os.path.join(out_dir, p["name"] + "_enhanced.wav"), SE.SAMPLE_RATE,
                          np.clip(enhanced[i][:len(noisy[i])] * 32768.0, -32768, 32767).astype(np.int16))
    return rows
