#!/bin/bash
# per-kernel times of the weight-gradient probe in its three modes (0 = as shipped, 1 = no P stores, 2 = no operand loads)
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
for m in 0 1 2; do
  for shape in "64 96 128 128 9" "128 192 128 128 5"; do
    rm -rf /tmp/wfp; rocprofv3 --kernel-trace --stats -d /tmp/wfp -o s -- tools/wgrad_fft_probe$m $shape > /tmp/wfp.log 2>&1
    echo "mode $m shape $shape: $(grep 'us per call' /tmp/wfp.log)"
    python profiles/summarize.py stats $(find /tmp/wfp -name "*.db" | head -1) | cut -c1-110 | head -5
  done
done
