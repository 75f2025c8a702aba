#!/usr/bin/env python3
"""CPU side of SURVEY 8(d): the oracle (oracle/nutls_ref.py, torch-CPU fp32) at B = 64 and B = 1 over a sweep of host thread counts,
up to os.cpu_count() -- the evidence behind bench.py's `cpu_baseline.cores` (8: the fastest setting on the GPU boxes' hosts).

    python tools/cpu_thread_sweep.py > profiles/r04_cpu_threads.txt
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    cores = os.cpu_count() or 1
    print("host: %s, os.cpu_count() = %d" % (bench.cpu_model_name(), cores))
    print("%8s %14s %14s" % ("threads", "B=64 frames/s", "B=1 frames/s"))
    for t in sorted({1, 2, 4, 8, 16, 32, 64, 128, cores}):
        if t > cores:
            continue
        b64 = bench._time_oracle(64, t, 4.0, max_steps=64)
        b1 = bench._time_oracle(1, t, 2.0, max_steps=128)
        print("%8d %14.1f %14.1f   (%d / %d steps)" % (t, b64[0], b1[0], b64[1], b1[1]))
        sys.stdout.flush()


if __name__ == "__main__":
    main()
