#!/bin/bash
# the drop-in call pattern next to the batched one: bench lines -> gpurun_out/<tag>_dropin_lines.jsonl
cd $GRAFT_REPO_ROOT; TAG=${1:-r05}; OUT=gpurun_out/${TAG}_dropin_lines.jsonl; : > $OUT
for args in "--workload config1" "--workload config1 --per-image-loop" "--workload config2 --per-image-loop" "" "--per-image-loop" "--workload config5 --images-per-gpu 8 --per-image-loop"; do
  echo "# bench.py $args --steps 50 --warmup 5 --no-cpu-baseline" >> $OUT
  timeout -k 10 200 python3 bench.py $args --steps 50 --warmup 5 --no-cpu-baseline 2>gpurun_out/dropin_err.log >> $OUT || { echo "# FAILED rc $?" >> $OUT; tail -5 gpurun_out/dropin_err.log >> $OUT; }
done
python3 - $OUT <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if l.startswith("#"): print(l.strip()); continue
    d = json.loads(l)
    print("   ms_per_step %.4f  cold %.4f  host_enqueue_us %.1f  value %.3e  %s" % (d["ms_per_step"], d["ms_per_step_cold"], d["host_enqueue_us_per_step"], d["value"], d["call_pattern"]))
PY
