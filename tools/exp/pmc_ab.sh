#!/bin/bash
# usage: pmc_ab.sh <tag> <lib1> <lib2> ...  (names of build/exp/libnutls_<name>.so)
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
P=$R/nested-u-net-based-real-time-speech-enhancement-mobile-app_amd
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for n in "$@"; do
  for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA" "SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
    rm -rf /tmp/pm; NUTLS_DEV=1 NUTLS_LIB=$P/build/exp/libnutls_$n.so timeout 300 rocprofv3 --pmc $set -d /tmp/pm -o pm -- python $R/tools/gpu_pmc_workload.py >/dev/null 2>/tmp/pm_err.txt
    db=$(find /tmp/pm -name "*.db" | head -1)
    if [ -n "$db" ]; then python $R/tools/pmc_summary.py $db nutls_fused_step | sed "s/^/$n: /"; else echo "$n: no db for $set"; tail -3 /tmp/pm_err.txt; fi
  done
done > $OUT/pmc_ab.txt 2>&1
cat $OUT/pmc_ab.txt
