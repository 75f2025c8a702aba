#!/usr/bin/env python3
"""GPU box: step time of the three fused plans of the LSTM variant, each at one full round of workgroups (256 workgroups: B = 256 / 512 / 1024 for
1 / 2 / 4 streams per workgroup) -> profiles/plan_cost_model.json.  engine.cpp's plan choice (fused_setup: rounds of workgroups x step time of
the plan) carries the RATIOS of these times as constants; tests/test_abi.py::test_plan_cost_constants_match_the_measurement fails when the two
drift more than 10 % apart.

    python tools/gpu_plan_cost.py [out.json]"""
import json, os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import nunet_amd

out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "plan_cost_model.json")
times = {}
for g, B in ((1, 256), (2, 512), (4, 1024)):
    eng = nunet_amd.NutlsEngine(batch=B, mode="fused", streams_per_workgroup=g)
    assert eng.streams_per_workgroup == g
    rng = np.random.default_rng(1234)
    pool = torch.from_numpy((0.25 * np.abs(rng.standard_normal((8, B, 256)))).astype(np.float32)).cuda()
    out = torch.empty(B, 256, device="cuda")
    for s in range(64):
        eng.step(pool[s % 8], out)
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(5):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ev[0].record()
        for s in range(300):
            eng.step(pool[s % 8], out)
        ev[1].record()
        torch.cuda.synchronize()
        best = min(best, ev[0].elapsed_time(ev[1]) / 300)
    times[g] = best
    eng.close()
rec = {"what": "ms per step of one round of 256 workgroups, best of 5 windows of 300 steps, one box, back to back",
       "ms_per_step": {str(g): round(t, 4) for g, t in times.items()},
       "ratio_to_one_stream": {str(g): round(t / times[1], 3) for g, t in times.items()},
       "kernel_source_sha16": None}
try:
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    rec["kernel_source_sha16"] = bench.kernel_source_sha16("fused")
except Exception:
    pass
json.dump(rec, open(out_path, "w"), indent=1)
print(json.dumps(rec))
