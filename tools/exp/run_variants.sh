#!/bin/bash
# GPU box: for each library variant of tools/exp/build_plan_lib.sh -- parity of the one-stream kernel (golden / oracle tests), step time at
# B = 256 (two passes over all variants, alternating, so that clock drift of the box shows), phase table of the profiling twin.
#   tools/exp/run_variants.sh <tag> <name1> <name2> ...        ("main" = the in-tree library)     env: NOPARITY=1, NOPHASES=1
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
P=$R/nested-u-net-based-real-time-speech-enhancement-mobile-app_amd
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
python tools/exp/time_libs.py "$@" "$@" > $OUT/times.txt 2>&1
cat $OUT/times.txt
for n in "$@"; do
  if [ "$n" = main ]; then ENVV=""; else ENVV="NUTLS_DEV=1 NUTLS_LIB=$P/build/exp/libnutls_$n.so"; fi
  if [ -z "$NOPARITY" ]; then
    env $ENVV timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "golden or batch_256_synthetic or kat or carried" > $OUT/pytest_$n.txt 2>&1
    echo "$n: $(tail -1 $OUT/pytest_$n.txt)"
  fi
  if [ -z "$NOPHASES" ]; then
    env $ENVV timeout 600 python tools/gpu_phase_table.py $OUT/phases_$n.txt 30 > $OUT/phase_log_$n.txt 2>&1
    grep -E "^(x16b|r32b|ctfa|lstm|input|sum|step)" $OUT/phase_log_$n.txt | sed "s/^/$n: /"
  fi
done
