#!/usr/bin/env python3
"""GPU box: per-wave shader-clock trace of the fused step kernel (library built with -DFZ_WTRACE=1, tools/exp/build_plan_lib.sh): every wave
of workgroup 0 stamps s_memtime at the phase boundaries of every op.  Shows, per op and wave, WHEN each wave reaches each boundary relative
to the first wave entering the op -- i.e. who waits for whom at the two barriers of a small conv op.

    NUTLS_DEV=1 NUTLS_LIB=.../libnutls_wtrace.so python tools/gpu_wave_trace.py <out.txt> [reps] [op-name-substring ...]

Slots: 0 op entry | 1 loads issued | 2 weights in registers | 3 MFMA loop done | 4 partial tiles + parameters in LDS (in front of barrier 1) |
5 past barrier 1 | 6 row-wise epilogue done | 7 next image built | 9 in front of the op's last barrier | 10 past it.
(16x16-tile conv ops carry all of them; the other op kinds 0, 1, 9, 10.)"""
import os, sys, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import nunet_amd

out_path = sys.argv[1]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
pick = sys.argv[3:]
B = int(os.environ.get("B", "256"))
eng = nunet_amd.NutlsEngine(batch=B, mode="fused")
rng = np.random.default_rng(1234)
pool = torch.from_numpy((0.25 * np.abs(rng.standard_normal((8, B, 256)))).astype(np.float32)).cuda()
out = torch.empty(B, 256, device="cuda")
for s in range(64):
    eng.step(pool[s % 8], out)
torch.cuda.synchronize()
names = [p["layer"] for p in eng.fused_plan()]
n = len(names)
tmp = tempfile.mktemp(suffix=".bin")
os.environ["NUTLS_FUSED_WTRACE"] = tmp
for _ in range(5):
    eng.profile_fused()
acc = np.zeros((8, n, 12))
cnt = np.zeros((8, n, 12))
for _ in range(reps):
    eng.profile_fused()
    tr = np.fromfile(tmp, dtype=np.uint64).reshape(8, n, 12).astype(np.float64)
    t0 = tr[:, :, 0].min(axis=0)                      # first wave into each op
    ok = tr > 0
    rel = np.where(ok, tr - t0[None, :, None], 0.0)
    acc += rel
    cnt += ok
os.unlink(tmp)
eng.close()
mean = np.where(cnt > 0, acc / np.maximum(cnt, 1), np.nan)
if np.all(cnt == 0):
    sys.exit("no trace: the library was not built with -DFZ_WTRACE=1")

lab = ["entry", "loads", "wts", "mfma", "part", "bar1", "epi", "build", "-", "pre-b2", "bar2", "-"]
with open(out_path, "w") as f:
    f.write("# cycles after the first wave entered the op (mean of %d profiled steps, workgroup 0, B = %d); slots: %s\n" % (reps, B, " ".join("%d=%s" % (i, l) for i, l in enumerate(lab) if l != "-")))
    dur = mean[:, :, 10].max(axis=0)
    f.write("# op duration (last wave past the last barrier): sum %.0f cycles\n" % np.nansum(dur))
    for i, nm in enumerate(names):
        if pick and not any(p in nm for p in pick):
            continue
        f.write("%s  (op %d, %.0f cycles)\n" % (nm, i, dur[i]))
        for w in range(8):
            f.write("  w%d " % w + " ".join("%s %5.0f" % (lab[k], mean[w, i, k]) for k in (0, 1, 2, 3, 4, 5, 6, 7, 9, 10) if cnt[w, i, k] > 0) + "\n")
    # class view: for the 16x16-tile conv ops, the mean over ops of (a) each wave's arrival at barrier 1 / barrier 2 and (b) the release times
    conv = [i for i in range(n) if cnt[0, i, 4] > 0]
    f.write("# 16x16-tile conv ops (%d): mean over ops, cycles after op entry\n" % len(conv))
    for w in range(8):
        m = mean[w][conv]
        f.write("  w%d " % w + " ".join("%s %5.0f" % (lab[k], np.nanmean(m[:, k])) for k in (0, 1, 2, 3, 4, 5, 6, 7, 9, 10)) + "\n")
    small = [i for i in conv if dur[i] < 5000]
    f.write("# ... of which shorter than 5000 cycles (%d)\n" % len(small))
    for w in range(8):
        m = mean[w][small]
        f.write("  w%d " % w + " ".join("%s %5.0f" % (lab[k], np.nanmean(m[:, k])) for k in (0, 1, 2, 3, 4, 5, 6, 7, 9, 10)) + "\n")
print(open(out_path).read()[-3000:])
