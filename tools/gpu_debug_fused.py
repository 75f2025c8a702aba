#!/usr/bin/env python3
"""GPU box: fused mode against the plan-interpreter mode, tensor by tensor (first broken op = first line)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nunet_amd
from nunet_amd import NutlsEngine, topology as T

B = int(os.environ.get("B", "2"))
frames = int(os.environ.get("FRAMES", "3"))
clip = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "clip_4s.npz"))
a = NutlsEngine(batch=B, mode="graph")
b = NutlsEngine(batch=B, mode="fused")
names = []
for st in T.STAGES:
    order = [("%s_prev1" % st.conv_tag)]
    for i in range(2, st.depth + 1):
        order.append("%s_prev%d" % (st.conv_tag, i))
    order.append("%s_prev1" % st.spconv_tag)
    order += [st.prefix + "_h", st.prefix + "_c"]
    for j in range(2, st.depth + 1):
        order.append("%s_prev%d" % (st.spconv_tag, j))
    names += order
    if st.prefix == "msfe3_en":
        names += ["state_h", "state_c"]
for f in range(frames):
    x = np.stack([clip["mags_in"][(f + 31 * s) % 249] for s in range(B)])
    oa, ob = a.step(x), b.step(x)
    err = float(np.sqrt(np.mean((oa - ob) ** 2)))
    print("frame %d: out rms diff %.3e  (|out| rms %.3e, nan %d)" % (f, err, float(np.sqrt(np.mean(oa ** 2))), int(np.isnan(ob).sum())))
    bad = 0
    for n in names:
        va, vb = a.state_get(n), b.state_get(n)
        d = np.abs(va - vb)
        tol = 1e-4 * max(1.0, float(np.abs(va).max()))
        if not np.isfinite(vb).all() or d.max() > tol:
            bad += 1
            if bad <= int(os.environ.get("SHOW", "12")):
                va2, vb2 = va.reshape(B, -1, va.shape[-1]) if va.ndim == 3 else va.reshape(B, 1, -1), None
                vb2 = vb.reshape(va2.shape)
                dd = np.abs(va2 - vb2)
                rows = np.where(dd.max(axis=(0, 2)) > tol)[0]
                cols = np.where(dd.max(axis=(0, 1)) > tol)[0]
                print("   %-20s max|d| %.3e  rows bad %d/%d [%s..] cols bad %d/%d [%s..]" % (
                    n, float(np.nanmax(d)), len(rows), va2.shape[1], ",".join(map(str, rows[:6])), len(cols), va2.shape[2], ",".join(map(str, cols[:8]))))
    print("   %d / %d state tensors differ" % (bad, len(names)))
    if bad and not os.environ.get("KEEP_GOING"):
        break
a.close(); b.close()
