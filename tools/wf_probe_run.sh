#!/bin/bash
# per-kernel times of the weight-gradient probe (conv5 / conv2_fullres shapes at 128 channels; pass other shapes as arguments)
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
[ -x tools/wgrad_fft_probe ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I joint-cnn-mrf_amd/csrc tools/wgrad_fft_probe.hip -o tools/wgrad_fft_probe
for shape in "${@:-64 96 128 128 9}" ; do
  rm -rf /tmp/wfp; rocprofv3 --kernel-trace --stats -d /tmp/wfp -o s -- tools/wgrad_fft_probe $shape > /tmp/wfp.log 2>&1
  echo "shape $shape: $(grep 'us per call' /tmp/wfp.log)"
  python profiles/summarize.py stats $(find /tmp/wfp -name "*.db" | head -1) | cut -c1-110 | head -5
done
