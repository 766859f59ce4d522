#!/bin/bash
# same-box A/B of an environment switch: tools/ab_env.sh VAR "bench flags" [rounds]   (interleaved runs of VAR=0 and VAR=1)
VAR=$1; FLAGS=$2; N=${3:-3}
for i in $(seq $N); do
  for v in 0 1; do
    env $VAR=$v python bench.py $FLAGS --cpu-reps 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$VAR=$v', d['dtype'][:12], 'ms %.3f median %.3f  img/s %.0f' % (d['ms_per_step'], d['median_ms_per_step'], d['value']))"
  done
done
