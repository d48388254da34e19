#!/bin/bash
# kernel-trace stats of one bench configuration: kstats.sh "<bench args>" <tag>  -> gpurun_out/kstats_<tag>.csv (small kernels only printed)
cd /tmp && export TMPDIR=/tmp
ARGS=$1; TAG=$2
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/kstats_$TAG; rm -rf $OUT; mkdir -p $OUT
cd $R
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py $ARGS --steps 20 --warmup 5 --no-cpu-baseline > $OUT/stats.log 2>&1 || { echo "stats failed"; tail -5 $OUT/stats.log; exit 1; }
cp $(ls $OUT/stats/*/*kernel_stats.csv | head -1) $R/gpurun_out/kstats_$TAG.csv
rm -rf $OUT
python3 - <<PY
import csv
for r in csv.DictReader(open("$R/gpurun_out/kstats_$TAG.csv")):
    n = r["Name"].split("(")[0].split("::")[-1][:40]
    print("%-42s calls %4s avg %9.2f us" % (n, r["Calls"], float(r["AverageNs"]) / 1e3))
PY
