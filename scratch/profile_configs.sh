#!/bin/bash
# kernel stats + bench line of the other BASELINE configs (2, 4, 5)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/profc; rm -rf $OUT; mkdir -p $OUT; cd $R
for w in config2 config4 config5; do
  timeout -k 10 200 python3 bench.py --workload $w --steps 30 --warmup 5 --no-cpu-baseline > $OUT/bench_$w.json 2> $OUT/bench_$w.err || { echo "bench $w failed"; exit 1; }
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$w -- python3 bench.py --workload $w --steps 30 --warmup 5 --no-cpu-baseline > $OUT/stats_$w.log 2>&1 || { echo "stats $w failed"; exit 1; }
  cp $(ls $OUT/stats_$w/*/*kernel_stats.csv | head -1) $OUT/kernel_stats_$w.csv
  echo "$w done"
done
