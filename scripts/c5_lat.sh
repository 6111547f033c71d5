#!/bin/bash
# Development: latency and scalar-cache counters of the large-scene scan kernel.
#   scripts/c5_lat.sh <tag>
set -u
tag=$1
export TMPDIR=/tmp
out=gpurun_out/c5lat_$tag
mkdir -p $out
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_SMEM SQ_INST_LEVEL_SMEM SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVES SQ_INST_CYCLES_SMEM --output-format csv -d $out/a -o a -- python3 bench.py --config C5 --iterations 20 --steps 1 --warmup 0 --no-cpu > $out/a.log 2>&1 || { tail -5 $out/a.log; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_BUSY_CYCLES SQC_TC_STALL SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY --output-format csv -d $out/b -o b -- python3 bench.py --config C5 --iterations 20 --steps 1 --warmup 0 --no-cpu > $out/b.log 2>&1 || { tail -5 $out/b.log; exit 1; }
python3 scripts/pmc_report.py $out k_wf_bounce
