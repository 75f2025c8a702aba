#!/bin/bash
# GPU box: profile set of a packed plan of the fused kernel -> gpurun_out/prof_<tag>/
#   tools/gpu_packed_profile.sh <tag> [B=1024] [streams per workgroup=4]
#   bench line, per-op timeline (needs the optional profiling twin: NUTLS_BUILD_G4_PROF=1 at build time), rocprofv3 kernel trace, PMC passes
TAG=${1:-packed}
B=${2:-1024}
S=${3:-4}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export NUTLS_FUSED_STREAMS=$S
KPAT=nutls_fused_step_g$S
( cd $R && timeout 600 python bench.py --no-cpu-baseline --no-other-configs --batch $B --steps 200 > $OUT/bench.json 2> $OUT/bench.err )
( cd $R && B=$B NUTLS_FUSED_PHASES=$OUT/phases.txt timeout 600 python tools/gpu_fused_timeline.py $OUT/timeline.json > $OUT/timeline.txt 2>&1 )
rm -rf /tmp/kt; ( cd $R && timeout 600 rocprofv3 --kernel-trace -d /tmp/kt -o kt -- python bench.py --no-cpu-baseline --no-other-configs --batch $B --steps 200 > /dev/null 2>&1 )
python $R/tools/rocprof_summary.py $(find /tmp/kt -name "*.db" | head -1) > $OUT/kernel_stats.txt
{
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_INSTS_BRANCH" "SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA"; do
  rm -rf /tmp/pm; NUTLS_BATCH=$B timeout 300 rocprofv3 --pmc $set -d /tmp/pm -o pm -- python $R/tools/gpu_pmc_workload.py >/dev/null 2>&1
  python $R/tools/pmc_summary.py $(find /tmp/pm -name "*.db" | head -1) $KPAT
done
} > $OUT/pmc.txt 2>&1
python $R/tools/pmc_traffic.py $OUT/pmc.txt fused $B lstm $S > $OUT/pmc_traffic_g$S.json
tail -1 $OUT/bench.json | cut -c1-400; head -4 $OUT/timeline.txt; head -5 $OUT/kernel_stats.txt; cat $OUT/pmc.txt
