#!/bin/bash
# all round-4 profile runs (each: bench line, kernel stats, five PMC passes), then the strong-scaling anchor line
cd $GRAFT_REPO_ROOT
for run in "config3 saag 8" "config3 decoder_like 8" "config2 saag 16" "config4 saag 16" "config5 saag 1" "config5 saag 8"; do
  bash scratch/profile/profile_run.sh r04 $run || exit 1
done
timeout -k 10 300 python3 bench.py --images-per-gpu 64 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/prof_r04/bench_config3_saag_b64.json 2> gpurun_out/prof_r04/bench_config3_saag_b64.err
