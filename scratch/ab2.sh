#!/bin/bash
# same-box A/B of library builds: ab2.sh "<bench args>" libA.so libB.so ...   (alternates, 2 rounds; prints stage split)
ARGS=$1; shift
cd $GRAFT_REPO_ROOT
for round in 1 2; do
  for lib in "$@"; do
    FGS_LIB=$GRAFT_REPO_ROOT/fresnel_amd/_lib/$lib timeout -k 10 120 python3 bench.py $ARGS --steps 30 --warmup 6 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
s = d['roofline']['stage_avg_ms']
print('%-28s step %.4f ms | ' % ('$lib', d['ms_per_step']) + ' '.join('%s %.4f' % (k, v) for k, v in s.items()))
" || echo "$lib failed"
  done
done
