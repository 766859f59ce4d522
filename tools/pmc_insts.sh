#!/bin/bash
# Dynamic instruction mix per kernel (run on the GPU box): bash tools/pmc_insts.sh NAME <command...>  -> gpurun_out/NAME_insts.csv
set -u
NAME=$1; shift
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
mkdir -p $OUT
i=0
for ctr in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
  i=$((i+1))
  rocprofv3 --pmc $ctr -d $OUT/prof_${NAME}_ins$i -o p -- "$@" > $OUT/${NAME}_ins$i.log 2>&1
done
python profiles/summarize.py pmc $(find $OUT/prof_${NAME}_ins* -name "*.db" | sort) > $OUT/${NAME}_insts.csv
rm -rf $OUT/prof_${NAME}_ins*
grep -E "cfft|cgemm|kernel,counter" $OUT/${NAME}_insts.csv
