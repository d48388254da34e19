#!/bin/bash
# the round's parity records on ONE build: randomized sweeps (HIP side replayed against the precomputed oracle) + soak.
# usage (build container): scratch/fuzz/final_records.sh   -- un-ignores the replay files for this one call, stamps the commit
cd "$(dirname "$0")/../.."
C=$(git rev-parse --short HEAD)$(git diff --quiet HEAD -- fresnel_amd include || echo "-dirty")
cp .gpurunignore /tmp/gpurunignore.bak; grep -v "scratch/fuzz/replay" /tmp/gpurunignore.bak > .gpurunignore
gpurun --timeout 1100 -- "python scratch/fuzz/sweep.py run gpurun_out/r05_fuzz_sweeps.txt --commit $C > gpurun_out/sweep_final.log 2>&1; echo sweep exit status \$? >> gpurun_out/r05_fuzz_sweeps.txt; tail -3 gpurun_out/r05_fuzz_sweeps.txt; timeout -k 10 600 python scratch/fuzz/soak.py $C 2>&1 | grep -v amdgpu.ids > gpurun_out/r05_soak.txt; echo soak exit status \${PIPESTATUS[0]} >> gpurun_out/r05_soak.txt; cat gpurun_out/r05_soak.txt"
cp /tmp/gpurunignore.bak .gpurunignore
