import os, sys
import numpy as np
sys.path.insert(0, "/root/repo")
import nunet_amd
from nunet_amd import NutlsEngine
clip = np.load("/root/repo/tests/golden/clip_4s.npz")
a = NutlsEngine(batch=1, mode="persistent"); b = NutlsEngine(batch=1, mode="fused")
x = clip["mags_in"][40:41]
a.step(x); b.step(x)
for n in ("msfe6_ee_prev1", "msfe6_ee_prev2", "msfe6_ed_prev6"):
    va, vb = a.state_get(n)[0], b.state_get(n)[0]
    print(n, va.shape)
    np.set_printoptions(precision=4, suppress=True, linewidth=200)
    print(" a row5:", va[5, :12]); print(" b row5:", vb[5, :12])
    print(" a row6:", va[6, :12]); print(" b row6:", vb[6, :12])
    if n == "msfe6_ed_prev6":
        print(" a row5 c32:", va[5, 32:44]); print(" b row5 c32:", vb[5, 32:44])
