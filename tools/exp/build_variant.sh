#!/bin/bash
# Timing experiments on the one-stream step kernel: builds a library variant whose step kernel AND its profiling twin are compiled with
# extra -D flags (FZ_LATE, FZ_HIW, FZ_ABL, ...; see fused_step.hip):   tools/exp/build_variant.sh <name> "<extra hipcc flags>"
# -> nested-.../build/exp/libnutls_<name>.so (git-ignored, travels to the GPU box); run with NUTLS_DEV=1 NUTLS_LIB=<path>.
# The other objects are taken from the in-tree build as they are (run the normal build first).
set -e
NAME=$1; shift
FLAGS="$*"
R=$(cd $(dirname $0)/../.. && pwd)
P=$R/nested-u-net-based-real-time-speech-enhancement-mobile-app_amd
mkdir -p $P/build/exp
CC="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function $FLAGS"
$CC -c $P/csrc/fused_step.hip -o $P/build/exp/fused_step_$NAME.o &
$CC -c $P/csrc/fused_step_prof.hip -o $P/build/exp/fused_step_prof_$NAME.o &
wait
OBJS=""
for s in fused_step_g2 fused_step_g4 fused_base fused_base_prof kernels stft offline weights fused_host engine; do OBJS="$OBJS $P/build/$s.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $P/build/exp/libnutls_$NAME.so $P/build/exp/fused_step_$NAME.o $P/build/exp/fused_step_prof_$NAME.o $OBJS
echo built $P/build/exp/libnutls_$NAME.so
