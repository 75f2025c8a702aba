#!/usr/bin/env python3
"""SNR / SI-SNR of the GPU pipeline on a directory of <name>_0.wav (noisy) / <name>.wav (clean) pairs, e.g. the
reference's dnn_model/data:   python tools/evaluate_clips.py /root/reference/dnn_model/data [--out enhanced/]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nunet_amd  # noqa: E402,F401
from nunet_amd.evaluate import evaluate_directory  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("directory")
ap.add_argument("--out", default=None)
ap.add_argument("--dc-mode", default="edge", choices=["edge", "zero"])
a = ap.parse_args()
rows = evaluate_directory(a.directory, a.out, a.dc_mode)
print("%-16s %7s %12s %12s %12s %12s" % ("clip", "sec", "SNR before", "SNR after", "SI-SNR bef.", "SI-SNR aft."))
for r in rows:
    print("%-16s %7.2f %12.2f %12.2f %12.2f %12.2f" % (r["name"], r["seconds"], r["snr_before"], r["snr_after"], r["sisnr_before"], r["sisnr_after"]))
