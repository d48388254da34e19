#!/bin/bash
# Profile of ONE bench configuration: un-profiled bench line, rocprofv3 kernel stats, PMC passes
# (FETCH_SIZE | WRITE_SIZE | TCC hit / miss | SQ groups in separate runs, the program directly after `--`).
# usage: profile_run.sh <round tag, e.g. r03> <workload> <distribution> <images-per-gpu> [steps]
cd /tmp && export TMPDIR=/tmp
TAG=$1; W=$2; DI=$3; B=$4; STEPS=${5:-20}
R=$GRAFT_REPO_ROOT; KEY=${W}_${DI}_b${B}; OUT=$R/gpurun_out/prof_$TAG/$KEY; rm -rf $OUT; mkdir -p $OUT
cd $R
ARGS="--workload $W --distribution $DI --images-per-gpu $B"
timeout -k 10 300 python3 bench.py $ARGS --steps $STEPS --warmup 5 > $OUT/bench.json 2> $OUT/bench.err || { echo "$KEY bench failed"; tail -5 $OUT/bench.err; exit 1; }
echo "$KEY bench done"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py $ARGS --steps $STEPS --warmup 5 --no-cpu-baseline > $OUT/stats.log 2>&1 || { echo "$KEY stats failed"; tail -5 $OUT/stats.log; exit 1; }
echo "$KEY stats done"
i=0
for grp in "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VALU_TRANS_F32 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $grp --output-format csv -d $OUT/g$i -- python3 bench.py $ARGS --steps 3 --warmup 1 --spinup-ms 0 --no-cpu-baseline > $OUT/g$i.log 2>&1 || { echo "$KEY pmc group $i failed"; tail -5 $OUT/g$i.log; exit 1; }
  echo "$KEY pmc group $i done"
done
python3 scratch/profile/profile_parse.py $OUT $KEY
# keep only the small artefacts (the raw counter CSVs are tens of MB)
cp $(ls $OUT/stats/*/*kernel_stats.csv | head -1) $OUT/kernel_stats.csv
rm -rf $OUT/stats $OUT/g1 $OUT/g2 $OUT/g3 $OUT/g4 $OUT/g5
