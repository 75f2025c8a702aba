#!/usr/bin/env python3
"""Print a bench.py --profile-json timeline: per op, with the time at one CU's fp32 MFMA peak (2.4 GHz)."""
import json, sys
d = json.load(open(sys.argv[1]))
B = d["batch"]
print("mode", d.get("mode"), "timeline step ms %.4f" % d["timeline_step_ms"])
for o in d["ops"]:
    ideal = o.get("flops", 0.0) / B / (256 * 2.4e9) * 1e6
    print("%-24s us=%7.2f ideal_us=%6.2f" % (o["layer"], o["ms"] * 1e3, ideal))
