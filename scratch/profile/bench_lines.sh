#!/bin/bash
# un-profiled bench lines of the profiled configurations, re-run AFTER profiles/<tag>_pmc_summary.json exists so that
# roofline.traffic quotes this round's counters -> gpurun_out/bench_lines/<run>.json   (usage: bench_lines.sh)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/bench_lines
for run in "config3 saag 8" "config3 decoder_like 8" "config2 saag 16" "config4 saag 16" "config5 saag 1" "config5 saag 8"; do
  set -- $run
  timeout -k 10 300 python3 bench.py --workload $1 --distribution $2 --images-per-gpu $3 --steps 20 --warmup 5 > gpurun_out/bench_lines/$1_$2_b$3.json 2> gpurun_out/bench_lines/$1_$2_b$3.err || echo "$run failed"
done
