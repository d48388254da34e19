#!/bin/bash
# tune_sweep.sh "<bench args>" tuningA tuningB ...   ("-" = defaults); alternates, 2 rounds
ARGS=$1; shift
cd $GRAFT_REPO_ROOT
for round in 1 2; do
  for t in "$@"; do
    if [ "$t" = "-" ]; then TA=""; else TA="--tuning $t"; fi
    timeout -k 10 120 python3 bench.py $ARGS $TA --steps 30 --warmup 6 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
s = d['roofline']['stage_avg_ms']
print('%-24s step %.4f ms | ' % ('$t', d['ms_per_step']) + ' '.join('%s %.4f' % (k, v) for k, v in s.items() if k in ('composite_fwd','composite_bwd','project_bwd','list_building')))
" || echo "$t failed"
  done
done
