#!/bin/bash
# Timing experiments: builds libnutls variants whose step kernel is compiled with -DFZ_ABL=<mask> (parts compiled out, see
# fused_step.hip) or any other -D flags:   tools/exp/build_abl.sh <name> "<extra hipcc flags>"
# -> nested-.../build/exp/libnutls_<name>.so (git-ignored, travels to the GPU box); run with NUTLS_LIB=<path>.
set -e
NAME=$1; shift
FLAGS="$*"
R=$(cd $(dirname $0)/../.. && pwd)
P=$R/nested-u-net-based-real-time-speech-enhancement-mobile-app_amd
mkdir -p $P/build/exp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function $FLAGS -c $P/csrc/fused_step.hip -o $P/build/exp/fused_step_$NAME.o
OBJS=""
for s in fused_step_g2 fused_step_g4 fused_step_prof fused_base fused_base_prof kernels stft offline weights fused_host engine; do OBJS="$OBJS $P/build/$s.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $P/build/exp/libnutls_$NAME.so $P/build/exp/fused_step_$NAME.o $OBJS
echo built $P/build/exp/libnutls_$NAME.so
