#!/bin/bash
# PMC passes over bench.py (config 3, 8 images): one rocprofv3 run per counter group, program directly after `--`.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc
rm -rf $OUT; mkdir -p $OUT
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" \
           "SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_BRANCH SQ_IFETCH SQ_INST_CYCLES_SALU SQ_INSTS_SMEM" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_LEVEL_WAVES" \
           "SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1))
  (cd $R && timeout -k 10 240 rocprofv3 --pmc $grp --output-format csv -d $OUT/g$i -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > $OUT/g$i.log 2>&1) || { echo "group $i failed"; tail -5 $OUT/g$i.log; exit 1; }
  echo "group $i done"
done
cd $R && python3 scratch/pmc_parse.py $OUT
