#!/bin/bash
# GPU box: profile of the offline / block mode -> gpurun_out/prof_<tag>/
#   bench lines for 1024- and 256-frame blocks at several chunk counts of the block pipeline, and the rocprofv3 kernel trace
#   (per-kernel totals) of `bench.py --offline 1024`.
TAG=${1:-offline}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
( cd $R && for c in 0 1 2 3 4; do timeout 300 python bench.py --no-cpu-baseline --offline 1024 --steps 20 --offline-chunks $c; done; timeout 300 python bench.py --no-cpu-baseline --offline 256 --steps 40; NUTLS_OFFLINE_FP32=1 timeout 300 python bench.py --no-cpu-baseline --offline 1024 --steps 20 ) > $OUT/bench_offline.json 2>/dev/null
rm -rf /tmp/kto; ( cd $R && timeout 600 rocprofv3 --kernel-trace -d /tmp/kto -o kt -- python bench.py --no-cpu-baseline --offline 1024 --steps 20 --offline-chunks 1 > /dev/null 2>&1 )
python $R/tools/rocprof_summary.py $(find /tmp/kto -name "*.db" | head -1) > $OUT/kernel_stats.txt
python - <<PY
import json
for l in open("$OUT/bench_offline.json"):
    d = json.loads(l)
    print("%9.1f frames/s  %.4f ms/block  %s chunks=%s" % (d["value"], d["ms_per_step"], d["config"]["frames_per_block"], d["config"]["pipeline_chunks"]))
PY
head -30 $OUT/kernel_stats.txt
