#!/bin/bash
# Per-kernel register / LDS / spill figures of an object file (the code object's metadata notes):
#   scripts/kernel_resources.sh dino_tracker_amd/csrc/vit.o [name pattern]
set -e
OBJ=$(readlink -f $1); PAT=${2:-.}
TMP=$(mktemp -d)
cd $TMP
/opt/rocm/lib/llvm/bin/llvm-objcopy -O binary --only-section=.hip_fatbin $OBJ fat.bin
/opt/rocm/lib/llvm/bin/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=fat.bin --output=dev.co --unbundle
/opt/rocm/lib/llvm/bin/llvm-readelf --notes dev.co | awk -v pat="$PAT" '
/\.name:/ {name=$2}
/\.vgpr_count:/ {v=$2} /\.agpr_count:/ {a=$2} /\.sgpr_count:/ {s=$2}
/\.vgpr_spill_count:/ {vs=$2} /\.sgpr_spill_count:/ {ss=$2}
/\.group_segment_fixed_size:/ {l=$2} /\.private_segment_fixed_size:/ {p=$2}
/\.wavefront_size:/ { if (name ~ pat) printf "%-100s vgpr %3d agpr %3d sgpr %3d spill v%d s%d lds %6d scratch %d\n", substr(name,1,100), v, a, s, vs, ss, l, p }'
rm -rf $TMP
