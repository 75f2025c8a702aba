#!/usr/bin/env python3
"""GPU-box profiling aid: cycle stamps inside a dilated-dense block of the baseline variant."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import nunet_amd
from nunet_amd.weights import synthetic_weights, write_blob
B = 256
eng = nunet_amd.NutlsEngine(write_blob(synthetic_weights("baseline", seed=4321)), batch=B, variant="baseline")
x = (0.25 * np.abs(np.random.default_rng(0).standard_normal((B, 256)))).astype(np.float32)
for _ in range(5):
    eng.step(x)
plan = [p["layer"] for p in eng.launch_plan()]
for nm in sys.argv[1:]:
    os.environ["NUTLS_DBG_OP"] = str(plan.index(nm))
    os.environ["NUTLS_SUBSTAMPS"] = "/tmp/ss.txt"
    eng.profile_persistent()
    for ln in open("/tmp/ss.txt"):
        if ln.startswith("# op") or ln.startswith("#   wave"):
            print(ln.rstrip()[:200])
