#!/bin/bash
# end-of-round profiles: full passes for both configs + default bench line + training stats:   bash tools/prof_round.sh [PREFIX=r05]
# then, back in the build container:   python profiles/refresh_round.py PREFIX
P=${1:-r05}
DOM_US=700 bash tools/prof_full.sh ${P}_bf16_b256 --dtype bf16 > gpurun_out/${P}_bf16.log 2>&1      # (dominant = the conv4_fullres + conv5 GEMM launches: 0.85 / 1.5 ms; conv4_halfres is 0.35 ms)
DOM_US=700 bash tools/prof_full.sh ${P}_fp32_b64 --dtype fp32 > gpurun_out/${P}_fp32.log 2>&1
bash tools/pmc_insts.sh ${P}_bf16_b256 python bench.py --dtype bf16 --steps 2 --warmup 1 --cpu-reps 0 > gpurun_out/${P}_insts.log 2>&1
python bench.py --steps 20 --warmup 5 > gpurun_out/${P}_bench_default.json 2> gpurun_out/${P}_bench.err
cp gpurun_out/bench_detail.json gpurun_out/${P}_bench_detail.json
tail -c 300 gpurun_out/${P}_bench_default.json
bash tools/prof_stats.sh ${P}_train --train > gpurun_out/${P}_train.log 2>&1
