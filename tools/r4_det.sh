#!/bin/bash
export JCM_LIB=$GRAFT_REPO_ROOT/joint-cnn-mrf_amd/exp/libjcm_nopk.so
for prec in fp32 bf16; do
  DET_CHAIN=0 DET_ITERS=400 DET_LAYER_ITERS=100 DET_ONLY=model,conv5,conv6 timeout 900 python tools/determinism.py $prec 2>&1 | grep "two engines: engine . pd\|layer "
done
for v in base nopk; do
  if [ $v = base ]; then unset JCM_LIB; else export JCM_LIB=$GRAFT_REPO_ROOT/joint-cnn-mrf_amd/exp/libjcm_nopk.so; fi
  for dt in fp32 bf16; do
    python bench.py --dtype $dt --steps 20 --warmup 5 --cpu-images 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v $dt img/s %.0f ms/step %.2f gemm launch_ms %.3f' % (d['value'], d['ms_per_step'], d['roofline']['launch_ms']))"
  done
done
