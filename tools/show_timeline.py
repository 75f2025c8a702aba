#!/usr/bin/env python3
"""Print a bench.py --profile-json timeline: per family and per layer, with the MFMA-ideal time."""
import json, sys
d = json.load(open(sys.argv[1]))
B = d["batch"]
print("mode", d.get("mode"), "timeline step ms %.4f" % d["timeline_step_ms"])
fam = d["families"]
for k, v in sorted(fam.items(), key=lambda kv: -kv[1]["ms"]):
    ideal = v["flops"] / B / (256 * 2.4e9) * 1e3   # ms at one CU's fp32 MFMA peak
    print("%-18s n=%3d ms=%8.4f avg_us=%7.2f ideal_ms=%.4f  eff=%.2f" % (k, v["n"], v["ms"], 1e3 * v["ms"] / v["n"], ideal, ideal / v["ms"]))
if len(sys.argv) > 2:
    lo, hi = (int(x) for x in sys.argv[2].split(":"))
    for l in d["launches"][lo:hi]:
        ideal = l["flops"] / B / (256 * 2.4e9) * 1e6
        print("%-24s %-16s us=%7.2f ideal_us=%6.2f" % (l["layer"], l["family"], l["ms"] * 1e3, ideal))
