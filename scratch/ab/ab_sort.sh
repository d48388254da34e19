#!/bin/bash
# depth-sort pass implementations, same box: stage ms of `depth_sort` per FgsDims.sort_mode (bits 1-2: 0 fused 11-bit, 2 two-launch 8-bit, 4 fused 8-bit; bit 0 zone keys)
cd $GRAFT_REPO_ROOT
run() { timeout -k 10 120 python3 bench.py $1 --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); s = d['roofline']['stage_avg_ms']
print('%-44s step %.4f ms | depth_sort %.4f lists %.4f project %.4f' % ('$1', d['ms_per_step'], s.get('depth_sort', 0), s.get('list_building', 0), s.get('project', 0)))" || echo "$1 failed"; }
for round in 1 2 3; do
  for m in 0 2 4 6; do run "--workload config2 --tuning sort_mode=$m"; done
  for m in 0 2 4 6; do run "--workload config3 --tuning sort_mode=$m"; done
  for m in 1 3 5 7; do run "--workload config4 --tuning sort_mode=$m"; done
done
