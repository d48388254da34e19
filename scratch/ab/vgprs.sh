#!/bin/bash
# vgprs.sh <obj.o> <regex>: VGPR / SGPR / LDS / scratch of matching kernels
O=$1; PAT=$2
T=$(mktemp -d)
/opt/rocm/lib/llvm/bin/llvm-objcopy --dump-section .hip_fatbin=$T/fat.bin $O && \
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=$T/fat.bin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$T/dev.co
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $T/dev.co | awk -v pat="$PAT" '
/\.name:/ {name=$2} /\.vgpr_count:/ {v=$2} /\.sgpr_count:/ {s=$2} /\.group_segment_fixed_size:/ {l=$2} /\.private_segment_fixed_size:/ {p=$2} /\.agpr_count:/ {a=$2}
/\.wavefront_size:/ { if (name ~ pat) print name, "vgpr", v, "agpr", a, "sgpr", s, "lds", l, "scratch", p }'
rm -rf $T
