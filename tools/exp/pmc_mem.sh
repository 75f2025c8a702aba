#!/bin/bash
# GPU box: memory-system counters of the step kernel at several stream counts (where does the time between 128 and 256 streams go?)
#   tools/exp/pmc_mem.sh <tag> [lib name]      -> gpurun_out/<tag>/pmc_mem_B<b>.txt
TAG=${1:-pmc_mem}; LIB=$2
R=${GRAFT_REPO_ROOT:-/root/repo}
P=$R/nested-u-net-based-real-time-speech-enhancement-mobile-app_amd
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
[ -n "$LIB" ] && export NUTLS_DEV=1 NUTLS_LIB=$P/build/exp/libnutls_$LIB.so
cd /tmp && export TMPDIR=/tmp
for B in 128 256; do
  {
  for set in "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum" \
             "TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum" \
             "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCC_TAG_STALL_sum" \
             "TCC_BUSY_sum TCC_CYCLE_sum TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_RDREQ_32B_sum" \
             "TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_PERMISSION_MISS_sum" \
             "FETCH_SIZE WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY"; do
    rm -rf /tmp/pm; NUTLS_BATCH=$B timeout 300 rocprofv3 --pmc $set -d /tmp/pm -o pm -- python $R/tools/gpu_pmc_workload.py >/dev/null 2>/tmp/pm_err.txt
    python $R/tools/pmc_summary.py $(find /tmp/pm -name "*.db" | head -1) nutls_fused_step 2>/dev/null || { echo "set failed: $set"; tail -2 /tmp/pm_err.txt; }
  done
  } > $OUT/pmc_mem_B$B.txt 2>&1
done
paste $OUT/pmc_mem_B128.txt $OUT/pmc_mem_B256.txt | awk '{printf "%-36s B128 %12s   B256 %12s\n", $1, $3, $6}'
