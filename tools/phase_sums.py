#!/usr/bin/env python3
"""Sums of the in-op phase stamps (NUTLS_FUSED_PHASES file) by op class:  python tools/phase_sums.py gpurun_out/<tag>/phases.txt [lstm|base]"""
import json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
plan = json.load(open(os.path.join(ROOT, "tests", "golden", "fused_plan_%s.json" % (sys.argv[2] if len(sys.argv) > 2 else "lstm"))))
cls = {}
for o in plan["ops"]:
    if o["type"] == 1:
        cls[o["name"]] = "r32b" if o["path"] == 3 else ("x16b P>=16" if o["P"] >= 16 else "x16b P<16")
    else:
        cls[o["name"]] = {0: "input", 2: "lstm", 3: "ctfa", 4: "ddb"}[o["type"]]
tot, ph, cnt = {}, {}, {}
for l in open(sys.argv[1]):
    m = re.match(r"(\S+)\s+total\s+([\d.]+) \|(.*)", l)
    if not m:
        continue
    c = cls.get(m.group(1), "?")
    tot[c] = tot.get(c, 0) + float(m.group(2))
    cnt[c] = cnt.get(c, 0) + 1
    for k, v in re.findall(r"(\w+)\s+([\d.]+)", m.group(3)):
        ph.setdefault(c, {})
        ph[c][k] = ph[c].get(k, 0) + float(v)
for c in sorted(tot):
    print("%-12s n %3d total %7.2f us | %s" % (c, cnt[c], tot[c], "  ".join("%s %6.2f" % kv for kv in ph[c].items())))
print("sum %.2f us" % sum(tot.values()))
