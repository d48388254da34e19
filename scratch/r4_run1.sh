#!/bin/bash
# round 4, GPU call 1: parity of the rewritten phase path + everything else, dL/dlambda probe, config-4 A/B against the round-3 library
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r4_run1_pytest.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r4_run1_pytest.log
tail -5 gpurun_out/r4_run1_pytest.log
timeout -k 10 300 python scratch/dlambda_probe.py > gpurun_out/r4_dlambda_probe.txt 2>&1; echo "probe rc=$?"
bash scratch/ab4.sh "--workload config4" libfgs_hip_r3.so libfgs_hip.so > gpurun_out/r4_ab_config4_phase_rewrite.txt 2>&1
cat gpurun_out/r4_ab_config4_phase_rewrite.txt
timeout -k 10 300 python bench.py > gpurun_out/r4_bench_config3_first.json 2> gpurun_out/r4_bench_config3_first.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r4_bench_config3_first.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("ms_per_step", "ms_per_step_cold", "spinup_steps", "value")}, d["roofline"]["stage_avg_ms"])
PY
