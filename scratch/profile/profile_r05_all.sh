#!/bin/bash
# all round-5 profile runs (each: bench line, kernel stats, five PMC passes), then the strong-scaling anchor line
cd $GRAFT_REPO_ROOT
for run in "config3 saag 8" "config3 decoder_like 8" "config2 saag 16" "config4 saag 16" "config5 saag 1" "config5 saag 8"; do
  bash scratch/profile/profile_run.sh r05 $run || exit 1
done
timeout -k 10 300 python3 bench.py --images-per-gpu 64 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/prof_r05/bench_config3_saag_b64.json 2> gpurun_out/prof_r05/bench_config3_saag_b64.err
# code-path lines of the N > 1 branch on the one-GPU box (not scaling numbers): one RCCL rank, two gloo ranks sharing the GPU
timeout -k 10 300 python3 bench.py --force-dist --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/prof_r05/bench_rccl_1rank_codepath.json 2> gpurun_out/prof_r05/rccl1.err
timeout -k 10 300 python3 bench.py --gpus 2 --backend gloo --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/prof_r05/bench_gloo_2ranks_codepath.json 2> gpurun_out/prof_r05/gloo2.err
# the drop-in call pattern beside the batched one, and the whole training step
bash scratch/profile/dropin_lines.sh r05
cp gpurun_out/r05_dropin_lines.jsonl gpurun_out/prof_r05/dropin_lines.jsonl
timeout -k 10 300 python3 scratch/profile/train_step_bench.py > gpurun_out/prof_r05/train_step.json 2> gpurun_out/prof_r05/train_step.err
timeout -k 10 120 python3 scratch/profile/host_profile.py > gpurun_out/prof_r05/host_profile_config1.txt 2>&1
