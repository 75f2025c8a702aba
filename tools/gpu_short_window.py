import sys, time, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch, nunet_amd
B = 256
eng = nunet_amd.NutlsEngine(batch=B)
pool = torch.from_numpy((0.25 * np.abs(np.random.default_rng(0).standard_normal((8, B, 256)))).astype(np.float32)).cuda()
out = torch.empty(B, 256, device="cuda")
for s in range(32): eng.step(pool[s % 8], out)
torch.cuda.synchronize()
for K in (1, 5, 20, 20, 20, 50, 200, 200):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(K): eng.step(pool[s % 8], out)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("K %3d  total %.3f ms  per step %.4f ms  host enqueue %.3f ms (%.1f us/call)  tail wait %.3f ms" % (K, 1e3*(t2-t0), 1e3*(t2-t0)/K, 1e3*(t1-t0), 1e6*(t1-t0)/K, 1e3*(t2-t1)))
# empty sync cost
t0 = time.perf_counter(); torch.cuda.synchronize(); print("empty sync %.1f us" % (1e6*(time.perf_counter()-t0)))
