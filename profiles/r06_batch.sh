#!/bin/bash
# Round-6 measurement batch (one gpurun call): GPU test suite, smoke, the driver-form bench lines, the rocprofv3 kernel trace
# of the same bench command, and the counter passes behind every roofline.traffic (profiles/collect_pmc.py).
#   gpurun --timeout 3000 -- 'bash profiles/r06_batch.sh'
mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
rm -f gpurun_out/parity_measured.jsonl
(timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -vE "^(RCCL|HIP|ROCm|Hostname|Librccl)" | tail -15) > gpurun_out/r06_pytest_gpu.log
cp gpurun_out/parity_measured.jsonl gpurun_out/r06_parity_measured.jsonl 2>/dev/null
(timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -vE "^(RCCL|HIP|ROCm|Hostname|Librccl)" | tail -12) > gpurun_out/r06_smoke.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_bench_steps20.json 2> gpurun_out/r06_bench_steps20.err
timeout 600 python bench.py > gpurun_out/r06_bench_default.json 2> gpurun_out/r06_bench_default.err
cd /tmp; export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_bench
timeout 1200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_bench -o b -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 > $R/gpurun_out/r06_bench_prof.json 2>/dev/null
cd $R
DB=$(find gpurun_out/prof_bench -name "*_results.db" | head -1)
python profiles/summarize_rocpd.py $DB > gpurun_out/r06_bench_kernel_stats.txt 2>&1
rm -rf gpurun_out/prof_bench
timeout 2400 python profiles/collect_pmc.py --round r06 --legs headline:f64,headline:f32,cfg3:f64,cfg5:f64,cfg4:f64,cfg3:f32,cfg5:f32,cfg4:f32 > gpurun_out/r06_collect_pmc.log 2>&1
tail -3 gpurun_out/r06_pytest_gpu.log; tail -2 gpurun_out/r06_smoke.log; head -12 gpurun_out/r06_bench_kernel_stats.txt; tail -5 gpurun_out/r06_collect_pmc.log | cut -c1-300
