#!/usr/bin/env python3
"""GPU-box debugging aid for the baseline (dilated-dense) variant: HIP engine vs oracle B."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nunet_amd  # noqa: E402
from nunet_amd import NutlsEngine, topology as T  # noqa: E402
from nunet_amd.weights import parse_blob, synthetic_weights, write_blob  # noqa: E402
from oracle.nutls_ref import NutlsRef  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "launches"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
blob = write_blob(synthetic_weights("baseline", seed=4321, bias_std=0.1, affine_jitter=0.1), int8_convs=True)
w = parse_blob(blob)
eng = NutlsEngine(blob, batch=2, variant="baseline", mode=mode)
ref = NutlsRef(w, batch=2, variant="baseline")
print("engine created, ops per step:", eng.launches_per_step, flush=True)
rng = np.random.default_rng(1)
for s in range(steps):
    x = (0.25 * np.abs(rng.standard_normal((2, 256)))).astype(np.float32)
    o = eng.step(x)
    r = ref.step(x).numpy()
    print("step %d rms err %.3e (ref rms %.3e)" % (s, np.sqrt(np.mean((o - r) ** 2)), np.sqrt(np.mean(r ** 2))), flush=True)
bad = 0
for base, shp in T.state_specs("baseline"):
    n = base.format("prev")
    a = eng.state_get(n).reshape(2, -1)
    b = ref.state[n].numpy().reshape(2, -1)
    err = np.abs(a - b).max()
    if err > 1e-3 * max(1.0, np.abs(b).max()):
        bad += 1
        print("%-26s maxerr %.3e max|ref| %.3e  <<<<" % (n, err, np.abs(b).max()))
print("states bad:", bad)
