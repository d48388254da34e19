#!/bin/bash
# rocprofv3 kernel stats of one bench configuration (no PMC passes): stats_only.sh <workload> <distribution> <images> -> gpurun_out/stats_only/<key>_kernel_stats.csv
cd /tmp && export TMPDIR=/tmp
W=$1; DI=$2; B=$3; R=$GRAFT_REPO_ROOT; KEY=${W}_${DI}_b${B}; OUT=$R/gpurun_out/stats_only; mkdir -p $OUT; rm -rf $OUT/$KEY
cd $R
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$KEY -- python3 bench.py --workload $W --distribution $DI --images-per-gpu $B --steps 20 --warmup 5 --no-cpu-baseline > $OUT/$KEY.log 2>&1 || { echo "$KEY stats failed"; tail -5 $OUT/$KEY.log; exit 1; }
cp $(ls $OUT/$KEY/*/*kernel_stats.csv | head -1) $OUT/${KEY}_kernel_stats.csv && rm -rf $OUT/$KEY && echo "$KEY stats done"
