#!/bin/bash
# rocprofv3 passes over one bench configuration (run on the GPU box, from the repo root):
#   bash tools/prof_full.sh NAME <bench args, e.g. --dtype bf16>
# kernel-trace/stats in its own run; counters in separate --pmc runs (no tracing combined with --pmc);
# FETCH_SIZE and WRITE_SIZE cannot share a pass (TCC slots).  Summaries -> gpurun_out/NAME_*.csv
set -u
NAME=$1; shift
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
mkdir -p $OUT
CMD="python bench.py --steps 2 --warmup 1 --cpu-reps 0 $*"
rocprofv3 --kernel-trace --stats -d $OUT/prof_${NAME}_stats -o s -- $CMD > $OUT/${NAME}_stats.log 2>&1
i=0
for ctr in "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $ctr -d $OUT/prof_${NAME}_pmc$i -o p -- $CMD > $OUT/${NAME}_pmc$i.log 2>&1
done
DB=$(find $OUT/prof_${NAME}_stats -name "*.db" | head -1)
python profiles/summarize.py stats $DB > $OUT/${NAME}_kernel_stats.csv
python profiles/summarize.py trace $DB cgemm_split ${GEMM_LAST:-11} > $OUT/${NAME}_gemm_trace.csv
python profiles/summarize.py pmc $(find $OUT/prof_${NAME}_pmc* -name "*.db" | sort) > $OUT/${NAME}_pmc.csv
# the dominant launches only (conv4_fullres / conv5 GEMMs are the dispatches of cgemm_split_kernel longer than 0.9 ms)
python profiles/summarize.py pmc_min ${DOM_US:-900} $(find $OUT/prof_${NAME}_pmc* -name "*.db" | sort) | grep -E "kernel,|cgemm" > $OUT/${NAME}_pmc_dominant.csv
rm -rf $OUT/prof_${NAME}_stats $OUT/prof_${NAME}_pmc* $OUT/${NAME}_pmc?.log
head -14 $OUT/${NAME}_kernel_stats.csv | cut -c1-160
cat $OUT/${NAME}_pmc_dominant.csv | cut -c1-200
