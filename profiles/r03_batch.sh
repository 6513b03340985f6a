mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; rm -f gpurun_out/parity_measured.jsonl
(timeout 1200 python -m pytest tests -m gpu -q 2>&1 | grep -vE "^(RCCL|HIP|ROCm|Hostname|Librccl)" | tail -15) > gpurun_out/r03_pytest_gpu.log
timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r03_bench_steps20.json 2> gpurun_out/r03_bench_steps20.err
timeout 300 python bench.py > gpurun_out/r03_bench_default.json 2> gpurun_out/r03_bench_default.err
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_bench -o b -- python $R/bench.py --no-cpu-baseline --no-final-error --steps 20 --warmup 5 > $R/gpurun_out/r03_bench_prof.json 2>/dev/null
for nf in 10000 1000000; do for c in FETCH_SIZE WRITE_SIZE; do rocprofv3 --pmc $c --kernel-trace -d $R/gpurun_out/pmc_${c}_$nf -o p -- python $R/profiles/pmc_eval.py $nf > /dev/null 2>&1; done; done
cd $R
python profiles/summarize_rocpd.py gpurun_out/prof_bench/b_results.db > gpurun_out/r03_bench_kernel_stats.txt 2>&1
python profiles/summarize_pmc.py gpurun_out/pmc_FETCH_SIZE_10000/p_results.db gpurun_out/pmc_WRITE_SIZE_10000/p_results.db > gpurun_out/r03_pmc_fetch_write.txt 2>&1
python profiles/summarize_pmc.py gpurun_out/pmc_FETCH_SIZE_1000000/p_results.db gpurun_out/pmc_WRITE_SIZE_1000000/p_results.db > gpurun_out/r03_pmc_fetch_write_nf1e6.txt 2>&1
python profiles/time_configs.py > gpurun_out/r03_time_configs.txt 2>&1
rm -rf gpurun_out/prof_bench gpurun_out/pmc_FETCH_SIZE_10000 gpurun_out/pmc_WRITE_SIZE_10000 gpurun_out/pmc_FETCH_SIZE_1000000 gpurun_out/pmc_WRITE_SIZE_1000000
tail -5 gpurun_out/r03_pytest_gpu.log; head -c 400 gpurun_out/r03_bench_steps20.json; echo; head -8 gpurun_out/r03_bench_kernel_stats.txt; cat gpurun_out/r03_time_configs.txt
