export TMPDIR=/tmp
out=gpurun_out/pmc_extra
mkdir -p $out
timeout -s KILL 120 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES --output-format csv -d $out/a -o a -- python3 bench.py --no-cpu --iterations 40 --steps 1 --warmup 0 > $out/a.log 2>&1 || { tail -3 $out/a.log; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM --output-format csv -d $out/b -o b -- python3 bench.py --no-cpu --iterations 40 --steps 1 --warmup 0 > $out/b.log 2>&1 || { tail -3 $out/b.log; exit 1; }
python3 scripts/pmc_report.py $out k_trace
