#!/bin/bash
# the randomized HIP-vs-oracle sweeps, one log: run_sweeps.sh <tag>   (GPU box; ~10 minutes)
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/fuzz_$1.txt; : > $OUT
for s in 0 1 2 3 4 5; do echo "== fuzz_phase seed $s" >> $OUT; timeout -k 10 200 python scratch/fuzz/fuzz_phase.py $s 2>&1 | grep -v amdgpu.ids | grep "FAIL\|worst" >> $OUT; done
for s in 0 1; do echo "== fuzz seed $s" >> $OUT; timeout -k 10 200 python scratch/fuzz/fuzz.py $s 2>&1 | grep -v amdgpu.ids | grep "FAIL\|worst" >> $OUT; done
for s in 0 1; do echo "== fuzz_batch seed $s (scale range 2)" >> $OUT; timeout -k 10 300 python scratch/fuzz/fuzz_batch.py $s 2 2>&1 | grep -v amdgpu.ids | grep "FAIL\|worst" >> $OUT; done
for s in 0 1 2 3; do echo "== fuzz_asm seed $s" >> $OUT; timeout -k 10 300 python scratch/fuzz/fuzz_asm.py $s 2>&1 | grep -v amdgpu.ids | grep "FAIL\|worst" >> $OUT; done
echo "== fuzz_asm_batched seed 4, 24 cases" >> $OUT; timeout -k 10 600 python scratch/fuzz/fuzz_asm_batched.py 4 24 2>&1 | grep -v amdgpu.ids | tail -30 >> $OUT
cat $OUT
