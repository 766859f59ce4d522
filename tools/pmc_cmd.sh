#!/bin/bash
# PMC passes over an arbitrary command (run on the GPU box): bash tools/pmc_cmd.sh NAME <command...>
# counters in separate passes (no tracing combined with --pmc); summary -> gpurun_out/NAME_pmc.csv
set -u
NAME=$1; shift
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
mkdir -p $OUT
i=0
for ctr in "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $ctr -d $OUT/prof_${NAME}_pmc$i -o p -- "$@" > $OUT/${NAME}_pmc$i.log 2>&1
done
python profiles/summarize.py pmc $(find $OUT/prof_${NAME}_pmc* -name "*.db" | sort) > $OUT/${NAME}_pmc.csv
rm -rf $OUT/prof_${NAME}_pmc* $OUT/${NAME}_pmc?.log
cat $OUT/${NAME}_pmc.csv | grep -v "pack_"
