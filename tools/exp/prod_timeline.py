#!/usr/bin/env python3
"""GPU box: op-by-op timeline of the UN-instrumented step kernel.  A library built with -DFZ_STOPAT=1 (tools/exp/build_plan_lib.sh) ends a launch
in front of op N (nutls_debug_knob "skew" carries N); the step time as a function of N, differenced, is what each op costs the production
instruction stream -- no stamps, no extra waits (the profiling twin's phase tables over-credit anything that delays a prefetch: DESIGN.md).

    python tools/exp/prod_timeline.py <out.txt> <lib name> [<lib name> ...]        ("main" is not a STOPAT build: names of build/exp libraries)
Writes one row per op: cumulative and per-op microseconds for every library, side by side."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = os.path.join(ROOT, "nested-u-net-based-real-time-speech-enhancement-mobile-app_amd")
CODE = r'''
import os, sys, json, numpy as np, torch
sys.path.insert(0, %r)
import nunet_amd
B = int(os.environ.get("B", "256"))
eng = nunet_amd.NutlsEngine(batch=B, mode="fused", streams_per_workgroup=1)
names = [p["layer"] for p in eng.fused_plan()]
rng = np.random.default_rng(1234)
pool = torch.from_numpy((0.25 * np.abs(rng.standard_normal((8, B, 256)))).astype(np.float32)).cuda()
out = torch.empty(B, 256, device="cuda")
eng.debug_knob("skew", len(names) + 1)          # (never reached: the whole step)
for s in range(400): eng.step(pool[s %% 8], out)
torch.cuda.synchronize()
def timed(n, reps=3, k=60):
    eng.debug_knob("skew", n)
    best = 1e9
    for _ in range(reps):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ev[0].record()
        for s in range(k): eng.step(pool[s %% 8], out)
        ev[1].record(); torch.cuda.synchronize()
        best = min(best, ev[0].elapsed_time(ev[1]) / k * 1e3)
    return best
cum = [timed(n) for n in range(0, len(names) + 1)]
json.dump({"names": names, "cum_us": cum}, open(os.environ["OUT_JSON"], "w"))
eng.close()
''' % ROOT
out_path, libs = sys.argv[1], sys.argv[2:]
import json, tempfile
res = {}
for name in libs:
    tmp = tempfile.mktemp(suffix=".json")
    env = dict(os.environ, OUT_JSON=tmp, NUTLS_DEV="1", NUTLS_LIB=os.path.join(PKG, "build", "exp", "libnutls_%s.so" % name))
    r = subprocess.run([sys.executable, "-c", CODE], env=env, capture_output=True, text=True, timeout=1200)
    if r.returncode != 0:
        sys.stderr.write("%s FAILED: %s\n" % (name, r.stderr[-600:]))
        continue
    res[name] = json.load(open(tmp))
with open(out_path, "w") as f:
    f.write("# un-instrumented step kernel, launches that end in front of op N (FZ_STOPAT builds): us per op = T(N+1) - T(N); B = %s\n" % os.environ.get("B", "256"))
    f.write("%-24s" % "op" + "".join("  %10s cum" % n for n in res) + "\n")
    names = next(iter(res.values()))["names"]
    for i, nm in enumerate(names):
        f.write("%-24s" % nm + "".join("  %6.2f %7.1f" % (res[n]["cum_us"][i + 1] - res[n]["cum_us"][i], res[n]["cum_us"][i + 1]) for n in res) + "\n")
    f.write("%-24s" % "launch floor (N = 0)" + "".join("  %6.2f        " % res[n]["cum_us"][0] for n in res) + "\n")
print(open(out_path).read()[-1500:])
