#!/bin/bash
# Same-box A/B of the in-tree library against alternate builds (joint-cnn-mrf_amd/exp/libjcm_NAME.so, selected with JCM_LIB), interleaved:
#   AB_VARIANTS="name1 name2" AB_ROUNDS=3 bash tools/ab_libs.sh
for r in $(seq 1 ${AB_ROUNDS:-3}); do
  for v in base $AB_VARIANTS; do
    if [ "$v" = base ]; then unset JCM_LIB; else export JCM_LIB=$PWD/joint-cnn-mrf_amd/exp/libjcm_$v.so; fi
    for dt in bf16 fp32; do
      python bench.py --dtype $dt --steps 20 --warmup 5 --cpu-reps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v $dt %.3f ms' % d['ms_per_step'])"
    done
  done
done
