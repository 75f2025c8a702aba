import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import nunet_amd
eng = nunet_amd.NutlsEngine(batch=256, mode=os.environ.get("NUTLS_MODE", "fused"))
x = (0.25*np.abs(np.random.default_rng(0).standard_normal((256,256)))).astype(np.float32)
for _ in range(20): eng.step(x)
