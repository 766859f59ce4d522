JCM_LIB= python -m pytest tests/test_gpu_argmax_agreement.py tests/test_gpu_configs.py -x -q 2>&1 | tail -2
for r in 1 2 3; do
  for v in base new; do
    if [ $v = base ]; then export JCM_LIB=$PWD/joint-cnn-mrf_amd/exp/libjcm_base.so; else unset JCM_LIB; fi
    python bench.py --dtype bf16 --steps 20 --warmup 5 --cpu-reps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v %.3f ms  gemm %.3f' % (d['ms_per_step'], d['roofline']['launch_ms']))"
  done
done
