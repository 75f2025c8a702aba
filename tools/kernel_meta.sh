#!/bin/bash
# Code-object facts of a compiled kernel source: registers, spills, scratch, instruction count.   tools/kernel_meta.sh <object.o>
L=/opt/rocm/lib/llvm/bin
T=$(mktemp -d)
$L/llvm-objcopy --dump-section .hip_fatbin=$T/fb.bin "$1" && \
$L/clang-offload-bundler --unbundle --type=o --input=$T/fb.bin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$T/dev.co && \
$L/llvm-readelf --notes $T/dev.co | grep -E "\.name:|vgpr_count|agpr_count|sgpr_count|spill_count|private_segment_fixed_size" | sed 's/^ *//' | paste -sd' ' && \
$L/llvm-objdump -d $T/dev.co > $T/dis.s && \
echo "instructions: $(grep -c -E '^\s+[a-z_0-9]+ ' $T/dis.s)  mfma: $(grep -c v_mfma $T/dis.s)  scratch: $(grep -c scratch_ $T/dis.s)  s_barrier: $(grep -c s_barrier $T/dis.s)"
rm -rf $T
