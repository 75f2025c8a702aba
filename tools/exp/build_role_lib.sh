#!/bin/bash
# A library whose one-stream plans have ROLE OPS (tools/gen_fused_plan.py NUTLS_PLAN_ROLES=1), built beside the tree without touching it:
#   tools/exp/build_role_lib.sh <name> "<extra hipcc flags>"    -> nested-.../build/exp/libnutls_<name>.so   (run with NUTLS_DEV=1 NUTLS_LIB=<path>)
# A copy of csrc/ gets the role plans; the step kernel, its profiling twin, the baseline pair and fused_host.cpp are compiled from the copy, the
# other objects are taken from the in-tree build (run the normal build first).
set -e
NAME=$1; shift
FLAGS="$*"
R=$(cd $(dirname $0)/../.. && pwd)
P=$R/nested-u-net-based-real-time-speech-enhancement-mobile-app_amd
S=$P/build/exp/src_$NAME
rm -rf $S; mkdir -p $S
cp $P/csrc/* $S/
sed -i "s#\"../../include/nutls.h\"#\"$R/include/nutls.h\"#" $S/*.cpp $S/*.hpp $S/*.hip 2>/dev/null || true
NUTLS_PLAN_ROLES=1 NUTLS_PLAN_OUT=$S python $R/tools/gen_fused_plan.py > /dev/null
CC="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function $FLAGS"
for f in fused_step fused_step_prof fused_base fused_base_prof; do $CC -c $S/$f.hip -o $S/$f.o & done
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -c $S/fused_host.cpp -o $S/fused_host.o &
wait
OBJS="$S/fused_step.o $S/fused_step_prof.o $S/fused_base.o $S/fused_base_prof.o $S/fused_host.o"
for s in fused_step_g2 fused_step_g4 kernels stft offline weights engine; do OBJS="$OBJS $P/build/$s.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $P/build/exp/libnutls_$NAME.so $OBJS
rm -rf $S
echo built $P/build/exp/libnutls_$NAME.so
