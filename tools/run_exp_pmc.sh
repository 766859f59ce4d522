#!/bin/bash
# Effective clock and MFMA-busy fraction of the strip kernel per experiment build (one --pmc pass each):
#   EXP_VARIANTS="name1 name2" bash tools/run_exp_pmc.sh
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
for v in "" $EXP_VARIANTS; do
  if [ -z "$v" ]; then unset JCM_LIB; else export JCM_LIB=$GRAFT_REPO_ROOT/joint-cnn-mrf_amd/exp/libjcm_$v.so; fi
  rm -rf $OUT/prof_x
  rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES -d $OUT/prof_x -o p -- python bench.py --dtype bf16 --steps 1 --warmup 0 --cpu-images 0 --no-sm > /dev/null 2>&1
  python - "$v" $(find $OUT/prof_x -name "*.db") <<'PY'
import sqlite3, sys
con = sqlite3.connect(sys.argv[2])
rows = con.execute("select kernel_name, counter_name, count(*), avg(value), avg(duration) from counters_collection where kernel_name like '%strip%' group by kernel_name, counter_name").fetchall()
d = {r[1]: r for r in rows}
g, m = d['GRBM_GUI_ACTIVE'], d['SQ_VALU_MFMA_BUSY_CYCLES']
clk = g[3] / 8 / (g[4] * 1e-9) / 1e9
print('%-22s strip kernel avg %.2f ms  clock %.3f GHz  mfma busy %.3f' % (sys.argv[1] or 'base', g[4] / 1e6, clk, m[3] / 1024 / (g[3] / 8)))
PY
done
rm -rf $OUT/prof_x
