import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import nunet_amd
variant = os.environ.get("NUTLS_VARIANT", "lstm")
weights = None
if variant == "baseline":
    from nunet_amd.weights import synthetic_weights, write_blob
    weights = write_blob(synthetic_weights("baseline", seed=4321), int8_convs=True)
B = int(os.environ.get("NUTLS_BATCH", "256"))          # (1024 / 2048: the packed plans)
eng = nunet_amd.NutlsEngine(weights, batch=B, mode=os.environ.get("NUTLS_MODE", "fused"), variant=variant)
x = (0.25*np.abs(np.random.default_rng(0).standard_normal((B,256)))).astype(np.float32)
for _ in range(20): eng.step(x)
