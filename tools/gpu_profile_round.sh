#!/bin/bash
# GPU box: the per-round profile set of the step kernel -> gpurun_out/prof_<tag>/
#   bench line + in-kernel timeline, kernel trace (rocprofv3 --kernel-trace) of `bench.py` itself, PMC passes
#   (HBM bytes, instruction mix; one counter set per run, no tracing flags next to --pmc) on tools/gpu_pmc_workload.py.
TAG=${1:-final}
MODE=${2:-fused}
VARIANT=${3:-lstm}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
KPAT=nutls_fused_step; [ "$VARIANT" = baseline ] && KPAT=nutls_fused_base_step
# 1. bench line (with the CPU baseline leg) and the timeline
( cd $R && timeout 900 python bench.py --mode $MODE --variant $VARIANT --profile-json $OUT/timeline.json > $OUT/bench.json 2> $OUT/bench.err )
# 2. kernel trace of the same command
rm -rf /tmp/kt; ( cd $R && timeout 600 rocprofv3 --kernel-trace -d /tmp/kt -o kt -- python bench.py --mode $MODE --variant $VARIANT --no-cpu-baseline > /dev/null 2>&1 )
python $R/tools/rocprof_summary.py $(find /tmp/kt -name "*.db" | head -1) > $OUT/kernel_stats.txt
# 3. PMC passes
{
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_INSTS_BRANCH" "SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA"; do
  rm -rf /tmp/pm; NUTLS_MODE=$MODE NUTLS_VARIANT=$VARIANT timeout 300 rocprofv3 --pmc $set -d /tmp/pm -o pm -- python $R/tools/gpu_pmc_workload.py >/dev/null 2>&1
  python $R/tools/pmc_summary.py $(find /tmp/pm -name "*.db" | head -1) $KPAT
done
} > $OUT/pmc.txt 2>&1
python $R/tools/pmc_traffic.py $OUT/pmc.txt $MODE 256 $VARIANT > $OUT/pmc_traffic.json
[ "$VARIANT" = lstm ] && ( cd $R && { timeout 300 python bench.py --no-cpu-baseline --variant baseline; timeout 300 python bench.py --no-cpu-baseline --batch 1024 --host-io --steps 100; timeout 300 python bench.py --no-cpu-baseline --batch 2048 --steps 50; timeout 300 python bench.py --no-cpu-baseline --frontend; timeout 300 python bench.py --no-cpu-baseline --offline 1024 --steps 20; } > $OUT/bench_other_configs.json 2>/dev/null )
[ -f $OUT/bench_other_configs.json ] && cut -c1-330 $OUT/bench_other_configs.json
tail -1 $OUT/bench.json | cut -c1-900; head -5 $OUT/kernel_stats.txt; cat $OUT/pmc.txt; cat $OUT/pmc_traffic.json
