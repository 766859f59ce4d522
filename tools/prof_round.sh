#!/bin/bash
# end-of-round profiles: full passes for both configs + default bench line
bash tools/prof_full.sh r04b_bf16_b256 --dtype bf16 > gpurun_out/r04b_bf16.log 2>&1
DOM_US=700 bash tools/prof_full.sh r04b_fp32_b64 --dtype fp32 > gpurun_out/r04b_fp32.log 2>&1
bash tools/pmc_insts.sh r04b_bf16_b256 python bench.py --dtype bf16 --steps 2 --warmup 1 --cpu-images 0 > gpurun_out/r04b_insts.log 2>&1
python bench.py > gpurun_out/r04b_bench_default.json 2> gpurun_out/r04b_bench.err
tail -c 300 gpurun_out/r04b_bench_default.json
bash tools/prof_stats.sh r04b_train --train > gpurun_out/r04b_train.log 2>&1
