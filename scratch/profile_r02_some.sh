#!/bin/bash
# a subset of the round-2 profile runs: profile_r02_some.sh "config3 saag 8" "config2 saag 16" ...
cd $GRAFT_REPO_ROOT
for run in "$@"; do
  set -- $run
  bash scratch/profile_r02.sh $1 $2 $3 > gpurun_out/prof_r02_$1_$2_b$3.log 2>&1 || { echo "$run FAILED"; tail -5 gpurun_out/prof_r02_$1_$2_b$3.log; exit 1; }
  echo "$run ok"
done
