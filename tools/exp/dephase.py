#!/usr/bin/env python3
"""GPU box experiment: does de-phasing two half-batches avoid the chip-wide HBM bursts of the one-launch step?  Two handles of B/2 streams on
two HIP streams, each stepping back to back on its own (no join between them); aggregate frames/s with the second chain started `offset_us`
after the first, against one handle of B streams.     python tools/exp/dephase.py [B]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import nunet_amd

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
rng = np.random.default_rng(1234)
N = 600


def run_one():
    eng = nunet_amd.NutlsEngine(batch=B, mode="fused", streams_per_workgroup=1)
    pool = torch.from_numpy((0.25 * np.abs(rng.standard_normal((8, B, 256)))).astype(np.float32)).cuda()
    out = torch.empty(B, 256, device="cuda")
    for s in range(400): eng.step(pool[s % 8], out)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        for s in range(N): eng.step(pool[s % 8], out)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    eng.close()
    return B * N / best


def run_two(offset_us, parts=2):
    h = B // parts
    engs = [nunet_amd.NutlsEngine(batch=h, mode="fused", streams_per_workgroup=1) for _ in range(parts)]
    pools = [torch.from_numpy((0.25 * np.abs(rng.standard_normal((8, h, 256)))).astype(np.float32)).cuda() for _ in range(parts)]
    outs = [torch.empty(h, 256, device="cuda") for _ in range(parts)]
    streams = [torch.cuda.Stream() for _ in range(parts)]
    for s in range(400):
        for e, p, o, st in zip(engs, pools, outs, streams):
            with torch.cuda.stream(st): e.step(p[s % 8], o)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k, st in enumerate(streams):          # chain k starts k * offset_us late
            if k and offset_us > 0:
                with torch.cuda.stream(st): torch.cuda._sleep(int(offset_us * k * 2100))      # (cycles at ~2.1 GHz)
        for s in range(N):
            for e, p, o, st in zip(engs, pools, outs, streams):
                with torch.cuda.stream(st): e.step(p[s % 8], o)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    for e in engs: e.close()
    return B * N / best


print("one handle of %d streams:            %8.0f frames/s" % (B, run_one()))
for parts, off in ((2, 0), (2, 70), (2, 140), (2, 200), (4, 70), (4, 35)):
    print("%d handles of %d streams, offset %3d us: %8.0f frames/s" % (parts, B // parts, off, run_two(off, parts)))
