#!/usr/bin/env python3
"""GPU-box profiling aid: phase breakdown of every conv layer inside the persistent kernel
(workgroup 0, wall-clock stamps).  Writes gpurun_out/substamps.txt."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import nunet_amd  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
eng = nunet_amd.NutlsEngine(batch=B)
x = (0.25 * np.abs(np.random.default_rng(0).standard_normal((B, 256)))).astype(np.float32)
for _ in range(5):
    eng.step(x)
for _ in range(3):
    eng.profile_persistent()
os.makedirs("gpurun_out", exist_ok=True)
os.environ["NUTLS_SUBSTAMPS"] = "gpurun_out/substamps.txt"
us = eng.profile_persistent()
print("step us %.1f" % us.sum())
