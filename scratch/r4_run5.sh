#!/bin/bash
# round 4, GPU call 5: cost of -ffp-contract=off on the ASM unit (config 5), sort_mode A/B, sort tests
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_hip_parity.py tests/test_hip_headline.py -m gpu -q -k "depth_sort or integer or config4 or variants" > gpurun_out/r4_run5_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r4_run5_pytest.log
ab5() {
  for round in 1 2 3; do for lib in "$@"; do
    FGS_LIB=$GRAFT_REPO_ROOT/fresnel_amd/_lib/$lib timeout -k 10 120 python3 bench.py $ARGS --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); s = d['roofline']['stage_avg_ms']
print('%-22s step %.4f ms | splat %.4f / %.4f field %.4f / %.4f lists %.4f pbwd %.4f' % ('$lib', d['ms_per_step'], s.get('splat_fwd', 0), s.get('splat_bwd', 0), s.get('field_fwd', 0), s.get('field_bwd', 0), s.get('list_building', 0), s.get('project_bwd', 0)))" || echo "$lib failed"
  done; done
}
(ARGS="--workload config5" ab5 libfgs_hip.so libfgs_hip_asmnc.so; ARGS="--workload config5 --images-per-gpu 8" ab5 libfgs_hip.so libfgs_hip_asmnc.so) > gpurun_out/r4_ab_config5_contract_off.txt 2>&1
cat gpurun_out/r4_ab_config5_contract_off.txt
abs() {
  for round in 1 2 3; do for t in "$@"; do
    timeout -k 10 120 python3 bench.py $ARGS --tuning $t --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); s = d['roofline']['stage_avg_ms']
print('%-22s step %.4f ms | project %.4f sort %.4f lists %.4f fwd %.4f bwd %.4f pbwd %.4f' % ('$t', d['ms_per_step'], s.get('project', 0), s.get('depth_sort', 0), s.get('list_building', 0), s.get('composite_fwd', 0), s.get('composite_bwd', 0), s.get('project_bwd', 0)))" || echo "$t failed"
  done; done
}
(echo "== config 4"; ARGS="--workload config4" abs sort_mode=0 sort_mode=1) > gpurun_out/r4_ab_sort_mode_config4.txt 2>&1
cat gpurun_out/r4_ab_sort_mode_config4.txt
