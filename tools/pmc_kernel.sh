#!/bin/bash
# Counters of one kernel (name pattern) in the bf16 batch-256 step, one --pmc pass per counter group:
#   bash tools/pmc_kernel.sh kxfold "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"
# (run on the GPU box; JCM_LIB selects an experiment build; PMC_CMD replaces the profiled command, e.g. "python tools/sm_time.py")
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
PAT=$1; shift
for grp in "$@"; do
  rm -rf $OUT/prof_x
  rocprofv3 --pmc $grp -d $OUT/prof_x -o p -- ${PMC_CMD:-python bench.py --dtype bf16 --steps 1 --warmup 0 --cpu-reps 0 --no-sm} > /dev/null 2>&1
  python - "$PAT" $(find $OUT/prof_x -name "*.db") <<'PY'
import sqlite3, sys
con = sqlite3.connect(sys.argv[2])
rows = con.execute("select kernel_name, counter_name, count(*), avg(value), avg(duration) from counters_collection where kernel_name like ? group by kernel_name, counter_name", ('%' + sys.argv[1] + '%',)).fetchall()
for r in rows:
    print('%-28s n=%d  value %.6g  duration %.3f ms' % (r[1], r[2], r[3], r[4] / 1e6))
d = {r[1]: r for r in rows}
if 'GRBM_GUI_ACTIVE' in d:
    g = d['GRBM_GUI_ACTIVE']
    print('clock %.3f GHz' % (g[3] / 8 / (g[4] * 1e-9) / 1e9))
    if 'SQ_VALU_MFMA_BUSY_CYCLES' in d:
        print('mfma busy %.3f' % (d['SQ_VALU_MFMA_BUSY_CYCLES'][3] / 1024 / (g[3] / 8)))
PY
done
rm -rf $OUT/prof_x
