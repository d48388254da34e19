#!/bin/bash
# usage: disasm.sh <object or lib> [kernel-substring] -> /tmp/disasm/all.s (and /tmp/disasm/<substring>.s for the first kernel that matches)
# (pass the translation unit's object file, fresnel_amd/_lib/obj/<unit>.o: a .so holds one fat binary per unit and only the first is read)
obj=$1; k=$2
tmp=/tmp/disasm; mkdir -p $tmp
/opt/rocm/lib/llvm/bin/llvm-objcopy --dump-section .hip_fatbin=$tmp/fat.bin $obj && \
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=$tmp/fat.bin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$tmp/dev.co || exit 1
/opt/rocm/lib/llvm/bin/llvm-objdump -d --no-show-raw-insn $tmp/dev.co > $tmp/all.s
if [ -n "$k" ]; then
  awk -v k="$k" '/^[0-9a-f]+ <.*>:/{f = index($0, k) > 0 && !done} f{print} /s_endpgm/{if (f) {done = 1; f = 0}}' $tmp/all.s > $tmp/$k.s
  wc -l $tmp/$k.s
fi
