#!/usr/bin/env python3
"""GPU box: config 5 (B streams, host buffers in and out every call) by kind of host memory.
    python tools/exp/host_io.py [B]        (NUTLS_HOST_ZEROCOPY=0: pinned buffers through copy commands instead of direct access)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import nunet_amd
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
eng = nunet_amd.NutlsEngine(batch=B, mode="fused")
rng = np.random.default_rng(7)
pool = [(0.25 * np.abs(rng.standard_normal((B, 256)))).astype(np.float32) for _ in range(4)]
pin_in = [nunet_amd.host_alloc((B, 256)) for _ in range(4)]
for a, b in zip(pin_in, pool): a[...] = b
pin_out = nunet_amd.host_alloc((B, 256))
page_out = np.empty((B, 256), np.float32)

def run(step, n=300, reps=5):
    for i in range(60): step(i)
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        for i in range(n): step(i)
        best = min(best, (time.perf_counter() - t0) / n)
    return best

# correctness first: the same frames through both kinds of memory from the same state
eng.reset()
a = [eng.step(pool[i % 4]).copy() for i in range(6)]
eng.reset()
b = [eng.step(pin_in[i % 4], out=pin_out).copy() for i in range(6)]
print("pinned vs pageable, 6 frames: max abs diff %.3g" % max(float(np.abs(x - y).max()) for x, y in zip(a, b)))
for name, step in (("pageable, new output array per call", lambda i: eng.step(pool[i % 4])),
                   ("pageable, output array reused", lambda i: eng.step(pool[i % 4], out=page_out)),
                   ("pinned (nutls_host_alloc)", lambda i: eng.step(pin_in[i % 4], out=pin_out))):
    t = run(step)
    print("%-40s %.4f ms/step  %9.0f frames/s" % (name, t * 1e3, B / t))
eng.close()
