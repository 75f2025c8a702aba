#!/bin/bash
# GPU box: quick check of the fused step kernel while iterating on it: parity tests of the LSTM variant, step time, per-op timeline.
#   tools/gpu_quick.sh <tag> [pytest -k expression]
TAG=${1:-quick}
KEXPR=${2:-"golden or batch_256_synthetic or kat or modes_agree"}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "$KEXPR" > $OUT/pytest.txt 2>&1
tail -3 $OUT/pytest.txt
NUTLS_FUSED_PHASES=$OUT/phases.txt timeout 600 python tools/gpu_fused_timeline.py $OUT/timeline.json > $OUT/timeline.txt 2>&1
head -2 $OUT/timeline.txt
