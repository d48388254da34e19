#!/bin/bash
# two builds of the in-LDS depth sort, same box: stage ms of `depth_sort` at the automatic sort mode
cd $GRAFT_REPO_ROOT
run() { FGS_LIB=$1 timeout -k 10 120 python3 bench.py $2 --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); s = d['roofline']['stage_avg_ms']
print('%-26s %-40s step %.4f ms | depth_sort %.4f lists %.4f' % ('$(basename $1)', '$2', d['ms_per_step'], s.get('depth_sort', 0), s.get('list_building', 0)))" || echo "$1 $2 failed"; }
for round in 1 2 3; do
  for w in "--workload config2" "--workload config4 --tuning sort_mode=1" "--workload config1"; do
    for l in fresnel_amd/_lib/libfgs_hip.so fresnel_amd/_lib/libfgs_hip_$1.so; do run $l "$w"; done
  done
done
