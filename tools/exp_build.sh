#!/bin/bash
# Kernel A/B experiments: build an alternate libjcm with extra -D flags on ONE source file and select it at run
# time with JCM_LIB=... (see joint-cnn-mrf_amd/_lib.py).
#   tools/exp_build.sh NAME "-DEXP_FLAG ..." [source.hip]   ->  joint-cnn-mrf_amd/exp/libjcm_NAME.so
set -e
cd "$(dirname "$0")/../joint-cnn-mrf_amd/csrc"
SRC=${3:-conv_strip_bf16.hip}
BASE=${SRC%.hip}
make -s
mkdir -p ../exp build_exp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Xclang -target-feature -Xclang -packed-fp32-ops $2 -c $SRC -o build_exp/${BASE}_$1.o
OBJS=$(ls build/*.o | grep -v "build/${BASE}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../exp/libjcm_$1.so $OBJS build_exp/${BASE}_$1.o -L/opt/rocm/lib -lhipfft -ldl
