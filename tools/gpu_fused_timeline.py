#!/usr/bin/env python3
"""GPU box: fused-mode step time at batch B and the in-kernel per-op timeline (workgroup 0, wall clock)."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import nunet_amd

B = int(os.environ.get("B", "256"))
steps = int(os.environ.get("STEPS", "200"))
mode = os.environ.get("MODE", "fused")
variant = os.environ.get("VARIANT", "lstm")
weights = None
if variant == "baseline":
    from nunet_amd.weights import synthetic_weights, write_blob
    weights = write_blob(synthetic_weights("baseline", seed=4321), int8_convs=True)
eng = nunet_amd.NutlsEngine(weights, batch=B, mode=mode, variant=variant)
rng = np.random.default_rng(1234)
pool = torch.from_numpy((0.25 * np.abs(rng.standard_normal((8, B, 256)))).astype(np.float32)).cuda()
out = torch.empty(B, 256, device="cuda")
for s in range(32):
    eng.step(pool[s % 8], out)
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
ev[0].record()
for s in range(steps):
    eng.step(pool[s % 8], out)
ev[1].record()
torch.cuda.synchronize()
ms = ev[0].elapsed_time(ev[1]) / steps
print("mode %s B %d: %.4f ms/step  %.1f frames/s  frac of fp32 MFMA peak %.4f" % (mode, B, ms, B / ms * 1e3, B * 2 * 73967252 / (ms * 1e-3) / 157.3e12))
if mode == "fused":
    plan = eng.fused_plan()
    for _ in range(3):
        eng.profile_fused()
    us = np.zeros(len(plan))
    reps = 10
    for _ in range(reps):
        us += eng.profile_fused()
    us /= reps
    print("timeline total %.1f us" % us.sum())
    tot_ideal = 0
    rows = []
    for p, t in zip(plan, us):
        ideal = p["flops"] / (256 * 2.4e9) * 1e6
        tot_ideal += ideal
        rows.append({"layer": p["layer"], "us": float(t), "ideal_us": ideal})
        print("%-24s us=%7.2f ideal_us=%6.2f" % (p["layer"], t, ideal))
    print("sum ideal %.1f us" % tot_ideal)
    if len(sys.argv) > 1:
        json.dump({"batch": B, "mode": mode, "ms_per_step": ms, "ops": rows}, open(sys.argv[1], "w"), indent=1)
eng.close()
