#!/bin/bash
# same-box A/B for the ASM path (config 5): ab5.sh "<bench args>" libA.so libB.so ...  -- splat / field stage times
ARGS=$1; shift
cd $GRAFT_REPO_ROOT
for round in 1 2 3; do
  for lib in "$@"; do
    FGS_LIB=$GRAFT_REPO_ROOT/fresnel_amd/_lib/$lib timeout -k 10 120 python3 bench.py $ARGS --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); s = d['roofline']['stage_avg_ms']
print('%-26s step %.4f ms | splat %.4f / %.4f field %.4f / %.4f lists %.4f adjoint %.4f' % ('$lib', d['ms_per_step'], s.get('splat_fwd', 0), s.get('splat_bwd', 0), s.get('field_fwd', 0), s.get('field_bwd', 0), s.get('list_building', 0), s.get('project_bwd', 0)))" || echo "$lib failed"
  done
done
