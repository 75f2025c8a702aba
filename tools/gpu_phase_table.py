#!/usr/bin/env python3
"""GPU box: in-op phase table of the fused step kernel (profiling twin: wave 0 of workgroup 0 stamps the wall clock at every op boundary and
at the phase boundaries inside an op), averaged over many profiled steps -- the wall clock ticks every 10-40 ns, the average resolves less.

    python tools/gpu_phase_table.py <out.txt> [reps]           env: B (256), VARIANT (lstm | baseline), NUTLS_LIB / NUTLS_DEV (library variant)

Writes per-op rows (same format as NUTLS_FUSED_PHASES) followed by the sums per op class (tools/phase_sums.py's classes) and the un-profiled
step time of the same library."""
import os, re, sys, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import nunet_amd

out_path = sys.argv[1]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
B = int(os.environ.get("B", "256"))
variant = os.environ.get("VARIANT", "lstm")
weights = None
if variant == "baseline":
    from nunet_amd.weights import synthetic_weights, write_blob
    weights = write_blob(synthetic_weights("baseline", seed=4321), int8_convs=True)
eng = nunet_amd.NutlsEngine(weights, batch=B, mode="fused", variant=variant, streams_per_workgroup=int(os.environ.get("SPW", "0")) or None)
rng = np.random.default_rng(1234)
pool = torch.from_numpy((0.25 * np.abs(rng.standard_normal((8, B, 256)))).astype(np.float32)).cuda()
out = torch.empty(B, 256, device="cuda")
for s in range(64):
    eng.step(pool[s % 8], out)
torch.cuda.synchronize()
best = 1e9
for rep in range(3):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for s in range(300):
        eng.step(pool[s % 8], out)
    ev[1].record()
    torch.cuda.synchronize()
    best = min(best, ev[0].elapsed_time(ev[1]) / 300)

plan = eng.fused_plan()
names = [p["layer"] for p in plan]
tmp = tempfile.mktemp(suffix=".txt")
os.environ["NUTLS_FUSED_PHASES"] = tmp
for _ in range(5):
    eng.profile_fused()
order, acc, tot = {}, {}, np.zeros(len(names))
for _ in range(reps):
    eng.profile_fused()
    for i, l in enumerate(open(tmp)):
        m = re.match(r"(\S+)\s+total\s+([\d.]+) \|(.*)", l)
        tot[i] += float(m.group(2))
        kv = re.findall(r"([\w-]+)\s+([\d.]+)", m.group(3))
        order.setdefault(i, [k for k, _ in kv])
        for k, v in kv:
            acc.setdefault(i, {})
            acc[i][k] = acc[i].get(k, 0.0) + float(v)
os.unlink(tmp)
del os.environ["NUTLS_FUSED_PHASES"]
eng.close()

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import json
pj = json.load(open(os.path.join(ROOT, "tests", "golden", "fused_plan_%s.json" % ("lstm" if variant == "lstm" else "base"))))
cls = {}
for o in pj["ops"]:
    if o["type"] == 1:
        kind = {0: "in", 1: "el", 2: "dl", 3: "down", 4: "up"}[o["kind"]]
        cls[o["name"]] = ("r32b", kind) if o["path"] == 3 else (("x16b P>=16" if o["P"] >= 16 else "x16b P<16"), kind)
    else:
        cls[o["name"]] = ({0: "input", 2: "lstm", 3: "ctfa", 4: "ddb"}[o["type"]], "")
with open(out_path, "w") as f:
    f.write("# %s B=%d: un-profiled step %.4f ms; profiled timeline %.1f us (mean of %d profiled steps, wave 0 of workgroup 0)\n" % (variant, B, best, tot.sum() / reps, reps))
    csum, ccnt, ctot = {}, {}, {}
    ksum = {}
    for i, n in enumerate(names):
        row = "%-24s total %6.3f |" % (n, tot[i] / reps)
        c, kind = cls.get(n, ("?", ""))
        ctot[c] = ctot.get(c, 0.0) + tot[i] / reps
        ccnt[c] = ccnt.get(c, 0) + 1
        if kind:
            ksum[kind] = ksum.get(kind, 0.0) + tot[i] / reps
        for k in order.get(i, []):
            v = acc[i][k] / reps
            row += " %s %6.3f" % (k, v)
            csum.setdefault(c, {})
            csum[c][k] = csum[c].get(k, 0.0) + v
        f.write(row + "\n")
    f.write("# sums per op class (us per step)\n")
    for c in sorted(ctot):
        f.write("%-12s n %3d total %7.2f us (%5.2f each) | %s\n" % (c, ccnt[c], ctot[c], ctot[c] / ccnt[c], "  ".join("%s %6.2f" % kv for kv in csum.get(c, {}).items())))
    f.write("# conv families: " + "  ".join("%s %.1f" % kv for kv in sorted(ksum.items())) + "\n")
    f.write("sum %.2f us\n" % tot.sum() * 1 if False else "sum %.2f us\n" % (tot.sum() / reps))
print(open(out_path).read().split("# sums per op class")[1])
print("step %.4f ms" % best)
