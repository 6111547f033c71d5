#!/bin/bash
# Development: kernel-trace statistics and VALU counters of the large-scene (wavefront) pipeline.
set -u
tag=$1
export TMPDIR=/tmp
out=gpurun_out/c5_$tag
mkdir -p $out
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o t -- python3 bench.py --config C5 --iterations 170 --steps 1 --warmup 0 --no-cpu > $out/bench.json 2> $out/rocprof.log || { tail -5 $out/rocprof.log; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU --output-format csv -d $out/pmc -o p -- python3 bench.py --config C5 --iterations 85 --steps 1 --warmup 0 --no-cpu > $out/pmc.log 2>&1 || { tail -5 $out/pmc.log; exit 1; }
cut -d, -f1-4 $out/trace/t_kernel_stats.csv | cut -c1-150 | head -12
python3 scripts/pmc_report.py $out/pmc k_wf_bounce
