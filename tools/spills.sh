#!/bin/bash
# Register / spill table of the frame-loop kernels in a build directory's objects (code object notes):
#   tools/spills.sh [build dir] [frames|lds]      columns: SGPR spills, VGPRs, VGPR spills, kernel
B=${1:-spectroplot-js_amd/build}; K=${2:-frames}
T=$(mktemp -d)
for o in $B/${K}_*.o; do
  objcopy -O binary --only-section=.hip_fatbin $o $T/fb.bin
  /opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=$T/fb.bin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$T/k.co
  /opt/rocm/lib/llvm/bin/llvm-readelf --notes $T/k.co | grep -E "^\s+\.name:|\.vgpr_count|vgpr_spill|sgpr_spill" | paste - - - - | awk '{print $4, $6, $8, $2}' | c++filt | sed 's/(spk::.*//; s/void //'
done
rm -rf $T
