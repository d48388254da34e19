#!/bin/bash
# all round-2 profile runs (each: bench line, kernel stats, four PMC passes)
cd $GRAFT_REPO_ROOT
for run in "config3 saag 8" "config3 decoder_like 8" "config2 saag 16" "config4 saag 16" "config5 saag 1" "config5 saag 8"; do
  set -- $run
  bash scratch/profile_r02.sh $1 $2 $3 > gpurun_out/prof_r02_$1_$2_b$3.log 2>&1 || { echo "$run FAILED"; tail -5 gpurun_out/prof_r02_$1_$2_b$3.log; exit 1; }
  echo "$run ok"
done
