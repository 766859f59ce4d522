#!/bin/bash
# A/B the dominant bf16 conv launch across experiment builds (joint-cnn-mrf_amd/exp/libjcm_NAME.so, selected with
# JCM_LIB): run on the GPU box
#   EXP_VARIANTS="name1 name2" bash tools/run_exp.sh [extra bench args]
for v in "" $EXP_VARIANTS; do
  if [ -z "$v" ]; then unset JCM_LIB; else export JCM_LIB=$GRAFT_REPO_ROOT/joint-cnn-mrf_amd/exp/libjcm_$v.so; fi
  echo "== variant: ${v:-base}"
  python bench.py --dtype bf16 --batch 256 --steps 5 --warmup 2 --cpu-images 0 "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('img/s %.0f  ms/step %.2f  %s launch_ms %.2f  TF %.0f frac %.4f' % (d['value'], d['ms_per_step'], r.get('kernel_name'), r['launch_ms'], r['achieved'], r['frac']))"
done
