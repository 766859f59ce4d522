#!/bin/bash
# A/B of the register row kernels (JCM_FFT_REG) on one box: accuracy of the bf16 layers, then both bench configs
for r in 0 1; do
  echo "=== JCM_FFT_REG=$r"
  JCM_FFT_REG=$r python tools/bf16_layer_stats.py --time 2>&1 | head -6
  for dt in bf16 fp32; do
    JCM_FFT_REG=$r python bench.py --dtype $dt --steps 10 --warmup 3 --cpu-reps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$dt', round(d['value']), round(d['ms_per_step'],3))"
  done
done
JCM_FFT_REG=1 python -m pytest tests/test_gpu_golden.py -x -q 2>&1 | tail -3
