#!/usr/bin/env python3
"""Turn a rocprofv3 (ROCm 7.2) ``*_results.db`` kernel trace into the per-kernel stats table
(``--stats`` equivalent) as text:  python tools/rocprof_summary.py results.db > profiles/x.txt"""
import re
import sqlite3
import sys


def short(name: str) -> str:
    m = re.match(r"void nutls::(conv_mfma_kernel|conv_bf16x3_kernel)<(.*?)>\(", name)
    if m:
        return "%s<%s>" % (m.group(1), m.group(2).replace(" ", ""))
    return re.sub(r"\(.*", "", name).replace("void ", "")


def main():
    c = sqlite3.connect(sys.argv[1])
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else "kernel_name"
    rows = c.execute("select %s, start, end from kernels" % name_col).fetchall()
    agg = {}
    for name, s, e in rows:
        a = agg.setdefault(short(name), [0, 0, 10 ** 18, 0])
        d = e - s
        a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
    total = sum(a[1] for a in agg.values())
    print("%-62s %8s %12s %10s %10s %10s %6s" % ("KERNEL", "CALLS", "TOTAL_ns", "AVG_ns", "MIN_ns", "MAX_ns", "%"))
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print("%-62s %8d %12d %10.0f %10d %10d %6.2f" % (k, a[0], a[1], a[1] / a[0], a[2], a[3], 100.0 * a[1] / total))
    print("%-62s %8d %12d" % ("TOTAL", sum(a[0] for a in agg.values()), total))


if __name__ == "__main__":
    main()
