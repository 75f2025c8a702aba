#!/usr/bin/env python3
"""GPU-box debugging aid: run the HIP engine next to oracle B for a few frames of the golden
clip and print, in dataflow order, the error of every state tensor (= every conv input)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nunet_amd  # noqa: E402
from nunet_amd import NutlsEngine, topology as T  # noqa: E402
from oracle.nutls_ref import NutlsRef  # noqa: E402


def main():
    nframes = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    clip = np.load(os.path.join(ROOT, "tests/golden/clip_4s.npz"))
    mode = sys.argv[2] if len(sys.argv) > 2 else "graph"
    eng = NutlsEngine(batch=1, mode=mode)
    ref = NutlsRef(batch=1)
    print("launches per step:", eng.launches_per_step)
    for i in range(nframes):
        x = clip["mags_in"][i:i + 1]
        o = eng.step(x)
        r = ref.step(x).numpy()
        print("frame %d out rms err %.3e (ref rms %.3e)" % (i, np.sqrt(np.mean((o - r) ** 2)), np.sqrt(np.mean(r ** 2))))
    worst = []
    for base, shp in T.state_specs():
        k = base if len(shp) == 1 else base.format("prev")
        a = eng.state_get(k).reshape(-1)
        b = ref.state[k].numpy().reshape(-1)
        err = np.abs(a - b).max()
        worst.append((k, err, np.abs(b).max()))
    nbad = 0
    for k, err, mag in worst:
        bad = err > 1e-3 * max(1.0, mag)
        nbad += bad
        if bad or "-v" in sys.argv:
            print("%-20s maxerr %.3e  max|ref| %.3e%s" % (k, err, mag, "  <<<<" if bad else ""))
    print("states checked: %d, bad: %d, worst maxerr %.3e" % (len(worst), nbad, max(w[1] for w in worst)))


if __name__ == "__main__":
    main()
