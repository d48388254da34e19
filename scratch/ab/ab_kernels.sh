#!/bin/bash
# per-kernel same-box A/B: ab_kernels.sh "<bench args>" "<kernel name regex>" libA.so libB.so ...  (rocprofv3 kernel trace, average us per call)
ARGS=$1; PAT=$2; shift 2
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for lib in "$@"; do
  out=gpurun_out/abk_$$; rm -rf $out
  FGS_LIB=$GRAFT_REPO_ROOT/fresnel_amd/_lib/$lib timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 bench.py $ARGS --steps 30 --warmup 5 --no-cpu-baseline > /dev/null 2>&1 || echo "$lib failed"
  python3 - "$out" "$PAT" "$lib" <<'PY'
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_stats.csv")[0]
row = ["%-24s" % sys.argv[3]]
for r in csv.DictReader(open(f)):
    if re.search(sys.argv[2], r["Name"]):
        row.append("%s %.1f" % (re.sub(r"^.*::", "", r["Name"].split("(")[0])[:28], float(r["AverageNs"]) / 1e3))
print(" | ".join(row))
PY
  rm -rf $out
done
