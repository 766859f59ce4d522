#!/bin/bash
# same-box, interleaved A/B of the bf16 merge hand-over (option "fft_fuse" bit 1) at configs[2]
for r in 1 2 3; do
  for f in 1 3; do
    echo -n "bf16 fft_fuse=$f: "
    python bench.py --steps 10 --warmup 3 --extras 0 --dtype bf16 --fft-fuse $f 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['ms_per_step_profiled'], d['median_ms_per_step'], d['value'])"
  done
done
