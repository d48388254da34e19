#!/bin/bash
# round 4, GPU call 2: splat moment form + bit masks (parity + A/B), phase-kernel variants, K5 dL/dlambda with -ffp-contract=off, blend-bwd what-if
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r4_run2_pytest.log 2>&1; echo "pytest rc=$?"
tail -3 gpurun_out/r4_run2_pytest.log
echo "== config 4: phase variants" > gpurun_out/r4_ab_config4_phase_variants.txt
bash scratch/ab4.sh "--workload config4" libfgs_hip.so libfgs_hip_scan48.so libfgs_hip_scan32.so libfgs_hip_pck4park.so libfgs_hip_pck4.so libfgs_hip_pck4park48.so >> gpurun_out/r4_ab_config4_phase_variants.txt 2>&1
cat gpurun_out/r4_ab_config4_phase_variants.txt
ab5() {  # config 5 prints splat stages
  for round in 1 2 3; do for lib in "$@"; do
    FGS_LIB=$GRAFT_REPO_ROOT/fresnel_amd/_lib/$lib timeout -k 10 120 python3 bench.py $ARGS --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); s = d['roofline']['stage_avg_ms']
print('%-22s step %.4f ms | splat %.4f / %.4f field %.4f / %.4f lists %.4f pbwd %.4f' % ('$lib', d['ms_per_step'], s.get('splat_fwd', 0), s.get('splat_bwd', 0), s.get('field_fwd', 0), s.get('field_bwd', 0), s.get('list_building', 0), s.get('project_bwd', 0)))" || echo "$lib failed"
  done; done
}
ARGS="--workload config5" ab5 libfgs_hip_r3.so libfgs_hip.so > gpurun_out/r4_ab_config5_splat_moments.txt 2>&1
ARGS="--workload config5 --images-per-gpu 8" ab5 libfgs_hip_r3.so libfgs_hip.so >> gpurun_out/r4_ab_config5_splat_moments.txt 2>&1
cat gpurun_out/r4_ab_config5_splat_moments.txt
for lib in libfgs_hip.so libfgs_hip_asmnc.so; do echo "== $lib"; FGS_LIB=$GRAFT_REPO_ROOT/fresnel_amd/_lib/$lib timeout -k 10 300 python scratch/dlambda_probe.py 2>&1 | grep -A1 "^K\|^G16"; done > gpurun_out/r4_dlambda_probe_contract.txt
cat gpurun_out/r4_dlambda_probe_contract.txt
bash scratch/ab4.sh "" libfgs_hip.so libfgs_hip_whatif.so > gpurun_out/r4_ab_whatif_half_reductions.txt 2>&1
cat gpurun_out/r4_ab_whatif_half_reductions.txt
