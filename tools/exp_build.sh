#!/bin/bash
# Kernel A/B experiments: build an alternate libjcm with extra -D flags on the bf16 conv kernel
# and select it at run time with JCM_LIB=... (see joint-cnn-mrf_amd/_lib.py).
#   tools/exp_build.sh NAME "-DEXP_FLAG ..."   ->  joint-cnn-mrf_amd/exp/libjcm_NAME.so
set -e
cd "$(dirname "$0")/../joint-cnn-mrf_amd/csrc"
make -s
mkdir -p ../exp build_exp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off $2 -c conv_igemm_bf16.hip -o build_exp/cb_$1.o
OBJS=$(ls build/*.o | grep -v conv_igemm_bf16.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../exp/libjcm_$1.so $OBJS build_exp/cb_$1.o -L/opt/rocm/lib -lhipfft
