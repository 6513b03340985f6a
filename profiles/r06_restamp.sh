#!/bin/bash
# after a source change: the counter passes (re-stamped with the new library digest) and the two driver-form bench lines
#   gpurun --timeout 2400 -- 'bash profiles/r06_restamp.sh'    then copy gpurun_out/r06_pmc_traffic.json -> profiles/pmc_traffic.json
mkdir -p gpurun_out
timeout 2000 python profiles/collect_pmc.py --round r06 --legs headline:f64,headline:f32,cfg3:f64,cfg5:f64,cfg4:f64,cfg3:f32,cfg5:f32,cfg4:f32 > gpurun_out/r06_collect_pmc.log 2>&1
cp gpurun_out/r06_pmc_traffic.json profiles/pmc_traffic.json
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_bench_steps20.json 2> gpurun_out/r06_bench_steps20.err
timeout 600 python bench.py > gpurun_out/r06_bench_default.json 2> gpurun_out/r06_bench_default.err
tail -2 gpurun_out/r06_collect_pmc.log | cut -c1-200
