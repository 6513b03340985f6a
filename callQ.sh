cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT/gpurun_out
(cd /tmp && timeout 60 rocprofv3 -L 2>/dev/null | grep -i -o "SQC\?_[A-Z_]*ICACHE[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQ_WAIT_IFETCH\|SQ_INST_LEVEL[A-Z_]*\|SQC_INST[A-Z_]*" | sort -u > $R/icache_counters.txt)
cat $R/icache_counters.txt
(cd /tmp && timeout 200 rocprofv3 --pmc SQ_IFETCH SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace -d $R/pmc_ic -o i -- python $GRAFT_REPO_ROOT/profiles/pmc_eval.py > $R/pmc_ic.log 2>&1)
tail -3 $R/pmc_ic.log
for d in $(find $R/pmc_ic -name "*results.db"); do python profiles/summarize_pmc.py $d > $R/pmc_icache.txt; done
cat $R/pmc_icache.txt | grep "fused20"
find $R -name "*.db" -delete
