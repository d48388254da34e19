#!/bin/bash
# same-box A/B with more statistics: ab4.sh "<bench args>" libA.so libB.so ...   (alternates, 4 rounds of 100 steps; step ms only)
ARGS=$1; shift
cd $GRAFT_REPO_ROOT
for round in 1 2 3 4; do
  for lib in "$@"; do
    FGS_LIB=$GRAFT_REPO_ROOT/fresnel_amd/_lib/$lib timeout -k 10 120 python3 bench.py $ARGS --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
s = d['roofline']['stage_avg_ms']
print('%-26s step %.4f ms | fwd %.4f bwd %.4f' % ('$lib', d['ms_per_step'], s.get('composite_fwd', 0), s.get('composite_bwd', 0)))
" || echo "$lib failed"
  done
done
