#!/bin/bash
# usage: exp_build.sh NAME "-DFLAGS"   -> builds joint-cnn-mrf_amd/exp/libjcm_NAME.so with conv_igemm_bf16.hip recompiled
set -e
cd /root/repo/joint-cnn-mrf_amd/csrc
mkdir -p ../exp build_exp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off $2 -c conv_igemm_bf16.hip -o build_exp/cb_$1.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../exp/libjcm_$1.so build/jcm_api.o build/conv_igemm.o build_exp/cb_$1.o build/conv1.o build/glue.o build/spatial_model.o build/sm_fft.o -L/opt/rocm/lib -lhipfft
