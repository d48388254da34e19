#!/bin/bash
# round 4, GPU call 3: kz^2 fix + dL/dlambda at 1e-4, depth-sort key compression (parity + A/B), referee table
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/r4_run3_pytest.log 2>&1; echo "pytest rc=$?"
tail -15 gpurun_out/r4_run3_pytest.log
timeout -k 10 600 python scratch/referee_table.py > gpurun_out/r4_referee_table.txt 2> gpurun_out/r4_referee_table.err; echo "table rc=$?"; tail -3 gpurun_out/r4_referee_table.err
cat gpurun_out/r4_referee_table.txt
abs() {
  for round in 1 2 3; do for lib in "$@"; do
    FGS_LIB=$GRAFT_REPO_ROOT/fresnel_amd/_lib/$lib timeout -k 10 120 python3 bench.py $ARGS --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); s = d['roofline']['stage_avg_ms']
print('%-22s step %.4f ms | project %.4f sort %.4f lists %.4f fwd %.4f bwd %.4f pbwd %.4f' % ('$lib', d['ms_per_step'], s.get('project', 0), s.get('depth_sort', 0), s.get('list_building', 0), s.get('composite_fwd', 0), s.get('composite_bwd', 0), s.get('project_bwd', 0)))" || echo "$lib failed"
  done; done
}
(echo "== config 4"; ARGS="--workload config4" abs libfgs_hip_scan48.so libfgs_hip.so; echo "== config 2"; ARGS="--workload config2" abs libfgs_hip_scan48.so libfgs_hip.so; echo "== config 3"; ARGS="" abs libfgs_hip_scan48.so libfgs_hip.so) > gpurun_out/r4_ab_sort_key_compression.txt 2>&1
cat gpurun_out/r4_ab_sort_key_compression.txt
timeout -k 10 300 python scratch/dlambda_probe.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r4_dlambda_probe_after.txt; cat gpurun_out/r4_dlambda_probe_after.txt | tail -25
