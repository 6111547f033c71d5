#!/bin/bash
# Development: HBM traffic and time per kernel of the large-scene pipeline.
#   scripts/c5_traffic.sh <tag>
set -u
tag=$1
export TMPDIR=/tmp
out=gpurun_out/c5tr_$tag
mkdir -p $out
timeout -s KILL 120 rocprofv3 --pmc GRBM_GUI_ACTIVE FETCH_SIZE --output-format csv -d $out/p4 -o p4 -- python3 bench.py --config C5 --iterations 85 --steps 1 --warmup 0 --no-cpu > $out/p4.log 2>&1 || { tail -5 $out/p4.log; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $out/p5 -o p5 -- python3 bench.py --config C5 --iterations 85 --steps 1 --warmup 0 --no-cpu > $out/p5.log 2>&1 || { tail -5 $out/p5.log; exit 1; }
python3 scripts/traffic_report.py $out
