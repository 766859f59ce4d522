#!/bin/bash
# rocprofv3 kernel stats of a bench config: bash tools/prof_stats.sh NAME <bench args>
set -u
NAME=$1; shift
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT/prof_${NAME}_stats -o s -- python bench.py --steps 2 --warmup 1 --cpu-reps 0 $* > $OUT/${NAME}_stats.log 2>&1
DB=$(find $OUT/prof_${NAME}_stats -name "*.db" | head -1)
python profiles/summarize.py stats $DB > $OUT/${NAME}_kernel_stats.csv
# the channel GEMM launches of the last forward, in layer order
python profiles/summarize.py trace $DB cgemm_split ${GEMM_LAST:-11} > $OUT/${NAME}_gemm_trace.csv
rm -rf $OUT/prof_${NAME}_stats
head -30 $OUT/${NAME}_kernel_stats.csv | cut -c1-150
