#!/bin/bash
# rocprofv3 kernel stats of the fp32 headline config (B=64): bash tools/prof_fp32.sh NAME
set -u
NAME=$1
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT/prof_${NAME}_stats -o s -- python bench.py --dtype fp32 --steps 2 --warmup 1 --cpu-reps 0 > $OUT/${NAME}_stats.log 2>&1
python profiles/summarize.py stats $(find $OUT/prof_${NAME}_stats -name "*.db" | head -1) > $OUT/${NAME}_kernel_stats.csv
rm -rf $OUT/prof_${NAME}_stats
head -8 $OUT/${NAME}_kernel_stats.csv | cut -c1-150
