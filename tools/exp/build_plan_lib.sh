#!/bin/bash
# A/B experiments on the one-stream step kernel: a library variant built BESIDE the tree from a copy of csrc/ -- with other planner knobs
# (tools/gen_fused_plan.py: NUTLS_PLAN_EPL1=0, NUTLS_PLAN_KFIRST=0, NUTLS_PLAN_LAZY=0 ...) and / or extra hipcc flags:
#   tools/exp/build_plan_lib.sh <name> "<VAR=value ...>" "<extra hipcc flags>"   -> nested-.../build/exp/libnutls_<name>.so
# Run it with NUTLS_DEV=1 NUTLS_LIB=<path> (tools/exp/run_variants.sh does).  Only the one-stream LSTM kernel, its profiling twin and
# fused_host.cpp (which packs the blob by the plan) are compiled from the copy; the other objects are the in-tree build's (build first).
set -e
NAME=$1; PLANENV=$2; FLAGS=$3
R=$(cd $(dirname $0)/../.. && pwd)
P=$R/nested-u-net-based-real-time-speech-enhancement-mobile-app_amd
S=$P/build/exp/src_$NAME
rm -rf $S; mkdir -p $S
cp $P/csrc/* $S/
sed -i "s#\"../../include/nutls.h\"#\"$R/include/nutls.h\"#" $S/*.cpp $S/*.hpp $S/*.hip 2>/dev/null || true
env $PLANENV NUTLS_PLAN_OUT=$S python $R/tools/gen_fused_plan.py > /dev/null
CC="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function $FLAGS"
# EXP_TUS: the kernel translation units compiled from the copy (default: the one-stream kernel and its profiling twin; "fused_step_g2" for the
# two-stream packed kernel, ...); every other object is the in-tree build's
TUS=${EXP_TUS:-"fused_step fused_step_prof"}
for f in $TUS; do $CC -c $S/$f.hip -o $S/$f.o & done
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -c $S/fused_host.cpp -o $S/fused_host.o &
wait
OBJS="$S/fused_host.o"
for s in fused_step fused_step_prof fused_step_stop fused_step_g2 fused_step_g4 fused_base fused_base_prof kernels stft offline weights engine; do
  case " $TUS " in *" $s "*) OBJS="$OBJS $S/$s.o";; *) OBJS="$OBJS $P/build/$s.o";; esac
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $P/build/exp/libnutls_$NAME.so $OBJS
rm -rf $S
echo built $P/build/exp/libnutls_$NAME.so
