#!/bin/bash
# usage: disasm.sh <lib.so> <kernel-substring> -> writes /tmp/<kernel>.s  (gfx950 code object of the library)
lib=$1; k=$2
tmp=$(mktemp -d)
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=$lib --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$tmp/dev.co 2>/dev/null || { 
  # shared library: extract .hip_fatbin section
  /opt/rocm/lib/llvm/bin/llvm-objcopy --dump-section .hip_fatbin=$tmp/fat.bin $lib && \
  /opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=$tmp/fat.bin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$tmp/dev.co; }
/opt/rocm/lib/llvm/bin/llvm-objdump -d --no-show-raw-insn $tmp/dev.co > $tmp/all.s
echo $tmp/all.s
