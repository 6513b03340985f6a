mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; rm -f gpurun_out/parity_measured.jsonl
(timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -vE "^(RCCL|HIP|ROCm|Hostname|Librccl)" | tail -25) > gpurun_out/r04_pytest_gpu.log
timeout 400 python bench.py --steps 20 --warmup 5 > gpurun_out/r04_bench_steps20.json 2> gpurun_out/r04_bench_steps20.err
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_bench -o b -- python $R/bench.py --no-cpu-baseline --no-final-error --steps 20 --warmup 5 > $R/gpurun_out/r04_bench_prof.json 2>/dev/null
for leg in cfg3 cfg4; do for c in FETCH_SIZE WRITE_SIZE; do rocprofv3 --pmc $c --kernel-trace -d $R/gpurun_out/pmc_${c}_$leg -o p -- python $R/profiles/pmc_eval.py $leg > /dev/null 2>&1; done; done
cd $R
python profiles/summarize_rocpd.py gpurun_out/prof_bench/b_results.db > gpurun_out/r04_bench_kernel_stats.txt 2>&1
for leg in cfg3 cfg4; do python profiles/summarize_pmc.py gpurun_out/pmc_FETCH_SIZE_$leg/p_results.db gpurun_out/pmc_WRITE_SIZE_$leg/p_results.db > gpurun_out/r04_pmc_fetch_write_$leg.txt 2>&1; done
python profiles/time_configs.py > gpurun_out/r04_time_configs.txt 2>&1
mkdir -p gpurun_out/pmc_db; for leg in cfg3 cfg4; do for c in FETCH_SIZE WRITE_SIZE; do cp gpurun_out/pmc_${c}_$leg/p_results.db gpurun_out/pmc_db/${leg}_$c.db; done; done
rm -rf gpurun_out/prof_bench gpurun_out/pmc_FETCH_SIZE_* gpurun_out/pmc_WRITE_SIZE_*
tail -8 gpurun_out/r04_pytest_gpu.log; head -c 600 gpurun_out/r04_bench_steps20.json; echo; head -12 gpurun_out/r04_bench_kernel_stats.txt; cat gpurun_out/r04_time_configs.txt
