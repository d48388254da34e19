cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests -m gpu -q > gpurun_out/r4_run11_pytest.log 2>&1; echo rc=$?; grep -n "AssertionError: \|passed\|failed\|^FAILED" gpurun_out/r4_run11_pytest.log | tail -5
bash scratch/ab/ab4.sh "--workload config4" libfgs_hip_scan48.so libfgs_hip.so > gpurun_out/r4_ab_config4_first_moments.txt 2>&1; tail -4 gpurun_out/r4_ab_config4_first_moments.txt
for run in "config4 saag 16" "config5 saag 1" "config5 saag 8"; do bash scratch/profile/profile_run.sh r04 $run > gpurun_out/prof_rerun.log 2>&1 || { tail -5 gpurun_out/prof_rerun.log; exit 1; }; done
echo profiles done
