#!/bin/bash
# same-box, interleaved A/B of the fused hand-overs of fp32 handles (option "fft_fuse"): headline configuration only
for r in 1 2 3; do
  for f in 0 3 1 2; do
    echo -n "fft_fuse=$f: "
    python bench.py --steps 20 --warmup 3 --extras 0 --fft-fuse $f 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['ms_per_step_profiled'], d['median_ms_per_step'], d['value'])"
  done
done
