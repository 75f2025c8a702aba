#!/usr/bin/env python3
"""pmc.txt (tools/pmc_summary.py lines) -> the HBM-traffic record bench.py reads (profiles/pmc_traffic.json).
FETCH_SIZE is doubled (gfx950: it reports half of a wide coalesced read stream, MI355X_MICROARCH.md section HBM)."""
import json, os, re, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import kernel_source_sha16, fused_kernel_name
txt, mode, batch = open(sys.argv[1]).read(), sys.argv[2], int(sys.argv[3])
variant = sys.argv[4] if len(sys.argv) > 4 else "lstm"
streams = int(sys.argv[5]) if len(sys.argv) > 5 else 1          # packed plans: streams per workgroup
def mean(name):
    m = re.search(r"^%s\s+dispatches=\s*(\d+)\s+mean=([\d.e+]+)" % name, txt, re.M)
    return (float(m.group(2)), int(m.group(1))) if m else (None, 0)
f, nf = mean("FETCH_SIZE"); w, nw = mean("WRITE_SIZE")
rec = {"kernel": fused_kernel_name(variant, streams) if mode == "fused" else "nutls_stream_step_kernel", "batch": batch, "mode": mode, "variant": variant,
       "streams_per_workgroup": streams, "kernel_source_sha16": kernel_source_sha16(mode, variant, streams), "FETCH_SIZE_KB": f, "WRITE_SIZE_KB": w,
       "fetch_correction": "x2 (gfx950: FETCH_SIZE reports half of a wide coalesced read stream; all loads here are 16 B/lane) -- MI355X_MICROARCH.md section HBM",
       "traffic_bytes": int((2 * f + w) * 1024) if f and w else None,
       "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, tools/gpu_profile_round.sh) on tools/gpu_pmc_workload.py, mean of %d dispatches" % min(nf, nw)}
print(json.dumps(rec, indent=1))
