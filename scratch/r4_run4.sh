#!/bin/bash
# round 4, GPU call 4: FMA-contraction bisect of the ASM unit (dL/dlambda), depth-sort key compression with overlapped plan loads
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_hip_parity.py -m gpu -q -k "depth_sort or integer or g15 or headline" > gpurun_out/r4_run4_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r4_run4_pytest.log
for lib in libfgs_hip.so libfgs_hip_nc1.so libfgs_hip_nc2.so libfgs_hip_nc4.so libfgs_hip_nc8.so libfgs_hip_nc16.so libfgs_hip_nc25.so libfgs_hip_asmnc.so; do
  echo "== $lib"
  FGS_LIB=$GRAFT_REPO_ROOT/fresnel_amd/_lib/$lib timeout -k 10 300 python scratch/dlambda_probe.py 2>&1 | grep -A1 "^K5\|^K4"
  FGS_LIB=$GRAFT_REPO_ROOT/fresnel_amd/_lib/$lib timeout -k 10 300 python -m pytest tests/test_hip_asm.py -m gpu -q -k "plane_recurrence or nonsquare" 2>&1 | grep "AssertionError: dL\|passed\|failed"
done > gpurun_out/r4_contract_bisect.txt 2>&1
cat gpurun_out/r4_contract_bisect.txt
abs() {
  for round in 1 2 3; do for lib in "$@"; do
    FGS_LIB=$GRAFT_REPO_ROOT/fresnel_amd/_lib/$lib timeout -k 10 120 python3 bench.py $ARGS --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); s = d['roofline']['stage_avg_ms']
print('%-22s step %.4f ms | project %.4f sort %.4f lists %.4f fwd %.4f bwd %.4f pbwd %.4f' % ('$lib', d['ms_per_step'], s.get('project', 0), s.get('depth_sort', 0), s.get('list_building', 0), s.get('composite_fwd', 0), s.get('composite_bwd', 0), s.get('project_bwd', 0)))" || echo "$lib failed"
  done; done
}
(echo "== config 4"; ARGS="--workload config4" abs libfgs_hip_scan48.so libfgs_hip.so; echo "== config 2"; ARGS="--workload config2" abs libfgs_hip_scan48.so libfgs_hip.so; echo "== config 3"; ARGS="" abs libfgs_hip_scan48.so libfgs_hip.so) > gpurun_out/r4_ab_sort_key_compression2.txt 2>&1
cat gpurun_out/r4_ab_sort_key_compression2.txt
