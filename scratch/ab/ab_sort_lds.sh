#!/bin/bash
# whole-segment LDS depth sort (one launch) against the pass forms it replaces, same box: FgsDims.sort_mode bits 1-3 --
# 0 automatic (LDS sort up to 8192 Gaussians per image) | 2 fused 11-bit passes (automatic's choice up to 4096 before) | 8 two-launch 8-bit passes; bit 0 zone keys
cd $GRAFT_REPO_ROOT
run() { timeout -k 10 120 python3 bench.py $1 --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); s = d['roofline']['stage_avg_ms']
print('%-52s step %.4f ms | depth_sort %.4f lists %.4f project %.4f' % ('$1', d['ms_per_step'], s.get('depth_sort', 0), s.get('list_building', 0), s.get('project', 0)))" || echo "$1 failed"; }
for round in 1 2 3; do
  for m in 0 2 8; do run "--workload config1 --tuning sort_mode=$m"; done
  for m in 0 2 8; do run "--workload config2 --tuning sort_mode=$m"; done
  for m in 1 3 9; do run "--workload config4 --tuning sort_mode=$m"; done
  for m in 0 8; do run "--workload config5 --tuning sort_mode=$m"; done
  for m in 0 8; do run "--workload config3 --tuning sort_mode=$m"; done
done
