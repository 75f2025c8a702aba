#!/usr/bin/env python3
"""GPU box: duration of the first steps of a fresh handle (HIP events around every launch) -- how long the cold start of the step kernel lasts
(code fetch, TLBs, clocks), i.e. what a short timed window right after creation (the driver's --warmup 5 --steps 20) still contains."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, nunet_amd
B = int(os.environ.get("B", "256"))
eng = nunet_amd.NutlsEngine(batch=B)
pool = torch.from_numpy((0.25 * np.abs(np.random.default_rng(0).standard_normal((8, B, 256)))).astype(np.float32)).cuda()
out = torch.empty(B, 256, device="cuda")
n = 96
ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
torch.cuda.synchronize()
ev[0].record()
for s in range(n):
    eng.step(pool[s % 8], out)
    ev[s + 1].record()
torch.cuda.synchronize()
ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(n)]
print("first steps (ms):", " ".join("%.3f" % m for m in ms[:12]))
for a, b in ((0, 5), (5, 25), (25, 45), (45, 96)):
    print("steps %2d..%2d mean %.4f ms" % (a, b, sum(ms[a:b]) / (b - a)))
# a second and third burst after idle gaps: is the slow start a property of a fresh handle / process, or of a GPU that was idle?
for gap in (0.2, 1.0, 0.002):
    time.sleep(gap)
    ev[0].record()
    for s in range(n):
        eng.step(pool[s % 8], out)
        ev[s + 1].record()
    torch.cuda.synchronize()
    ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(n)]
    print("after %.3f s idle: steps 0..5 %.4f  5..25 %.4f  25..45 %.4f  45..96 %.4f ms" % (gap, sum(ms[0:5]) / 5, sum(ms[5:25]) / 20, sum(ms[25:45]) / 20, sum(ms[45:96]) / 51))
