#!/bin/bash
# one image per call (the reference's eval / visualisation use, and each call of route A): stage breakdown and the work-split knobs
cd $GRAFT_REPO_ROOT
run() { timeout -k 10 120 python3 bench.py $1 --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); s = d['roofline']['stage_avg_ms']
print('%-64s step %.4f ms | ' % ('$1', d['ms_per_step']) + ' '.join('%s %.4f' % (k, v) for k, v in s.items()))" || echo "$1 failed"; }
for w in config2 config3; do
  for b in 1 2 4; do run "--workload $w --images-per-gpu $b"; done
  for t in seg_len=64 seg_len=256 tile_w=16 tile_w=32 fwd_variant=1 fwd_variant=2 fwd_variant=3; do run "--workload $w --images-per-gpu 1 --tuning $t"; done
done
