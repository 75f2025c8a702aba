#!/usr/bin/env python3
"""GPU-box profiling aid: shader-cycle stamps inside chosen conv layers of the persistent kernel.
usage: gpu_opcycles.py B name [name ...]   -> gpurun_out/opcycles.txt"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import nunet_amd  # noqa: E402

B = int(sys.argv[1])
names = sys.argv[2:]
eng = nunet_amd.NutlsEngine(batch=B)
x = (0.25 * np.abs(np.random.default_rng(0).standard_normal((B, 256)))).astype(np.float32)
for _ in range(5):
    eng.step(x)
plan = [p["layer"] for p in eng.launch_plan()]
os.makedirs("gpurun_out", exist_ok=True)
out = open("gpurun_out/opcycles.txt", "w")
for nm in names:
    idx = plan.index(nm)
    os.environ["NUTLS_DBG_OP"] = str(idx)
    for rep in range(2):
        os.environ["NUTLS_SUBSTAMPS"] = "/tmp/ss.txt"
        eng.profile_persistent()
        for ln in open("/tmp/ss.txt"):
            if ln.startswith("# op") or ln.startswith("#   wave"):
                out.write(ln)
out.close()
print(open("gpurun_out/opcycles.txt").read())
