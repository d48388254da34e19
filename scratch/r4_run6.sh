#!/bin/bash
# round 4, GPU call 6: full GPU suite on the build with the ASM unit compiled without contraction; referee table
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -q > gpurun_out/r4_run6_pytest.log 2>&1; echo "pytest rc=$?"
grep -n "AssertionError\|passed\|failed" gpurun_out/r4_run6_pytest.log | tail -15
timeout -k 10 600 python scratch/referee_table.py > gpurun_out/r4_referee_table.txt 2> gpurun_out/r4_referee_table.err; echo "table rc=$?"
grep "^K\|^G9\|^case\|^G14 r30" gpurun_out/r4_referee_table.txt
