#!/bin/bash
# Round profile of the default bench (config 3, 8 images): un-profiled bench line, rocprofv3 kernel stats, and
# PMC passes (FETCH_SIZE | WRITE_SIZE | SQ groups in separate runs, program directly after `--`).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/prof; rm -rf $OUT; mkdir -p $OUT
cd $R
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err || { echo bench failed; tail -5 $OUT/bench.err; exit 1; }
echo "bench done"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/stats.log 2>&1 || { echo stats failed; exit 1; }
echo "stats done"
i=0
for grp in "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VALU_TRANS_F32 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $grp --output-format csv -d $OUT/g$i -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/g$i.log 2>&1 || { echo "pmc group $i failed"; tail -5 $OUT/g$i.log; exit 1; }
  echo "pmc group $i done"
done
python3 scratch/profile_parse.py $OUT
