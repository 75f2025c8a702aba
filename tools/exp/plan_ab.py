#!/usr/bin/env python3
"""GPU box: step time of a handle of B streams on each fused plan (1 / 2 / 4 streams per workgroup), same box back to back.
    python tools/exp/plan_ab.py 512 768 1024 1536 2048"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import nunet_amd
for B in [int(a) for a in sys.argv[1:]]:
    row = []
    for g in (1, 2, 4):
        if B % g:
            row.append("   -   "); continue
        eng = nunet_amd.NutlsEngine(batch=B, mode="fused", streams_per_workgroup=g)
        rng = np.random.default_rng(1234)
        pool = torch.from_numpy((0.25 * np.abs(rng.standard_normal((4, B, 256)))).astype(np.float32)).cuda()
        out = torch.empty(B, 256, device="cuda")
        for s in range(100): eng.step(pool[s % 4], out)
        torch.cuda.synchronize()
        best = 1e9
        for rep in range(5):
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            ev[0].record()
            for s in range(100): eng.step(pool[s % 4], out)
            ev[1].record(); torch.cuda.synchronize()
            best = min(best, ev[0].elapsed_time(ev[1]) / 100)
        row.append("%.4f ms (%4.0f k f/s)" % (best, B / best))
        eng.close()
    default = nunet_amd.NutlsEngine(batch=B)
    print("B = %4d | 1 per workgroup %s | 2: %s | 4: %s | library's choice: %d" % (B, row[0], row[1], row[2], default.streams_per_workgroup))
    default.close()
