#!/bin/bash
# GPU-box profiling aid: dynamic instruction mix of the persistent kernel (PMC), per wave and op.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_INSTS_BRANCH" "SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY"; do
  rm -rf /tmp/pm; timeout 250 rocprofv3 --pmc $set -d /tmp/pm -o pm -- python $R/tools/gpu_pmc_workload.py >/dev/null 2>&1
  python $R/tools/pmc_summary.py $(find /tmp/pm -name "*.db" | head -1) nutls_stream
done | awk '{split($4,a,"="); v=a[2]; printf "%-22s %12.4g  per wave-op %8.1f\n", $1, v, v/(2048*161)}'
