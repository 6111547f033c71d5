#!/bin/bash
# Kernel-trace statistics and bench lines of the current build for profiles/.
#   scripts/profile_round.sh <tag>       (run on the GPU box from the repository root)
set -u
tag=$1
export TMPDIR=/tmp
out=gpurun_out/prof_$tag
mkdir -p $out
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o trace -- python3 bench.py --no-cpu > $out/bench_under_rocprof.json 2> $out/rocprof.log || { echo "rocprof pass failed"; tail -5 $out/rocprof.log; exit 1; }
timeout -k 10 300 python3 bench.py > $out/bench_nee15.json 2> $out/bench.err || { echo "bench failed"; tail -5 $out/bench.err; exit 1; }
timeout -k 10 200 python3 bench.py --no-cpu --scene main32 > $out/bench_main32.json 2>> $out/bench.err || exit 1
timeout -k 10 200 python3 bench.py --no-cpu --scene raytrace3 > $out/bench_raytrace3.json 2>> $out/bench.err || exit 1
timeout -k 10 200 python3 bench.py --no-cpu --scene nee15_lit --iterations 720 > $out/bench_nee15_lit.json 2>> $out/bench.err || exit 1
timeout -k 10 200 python3 bench.py --no-cpu --scene nee15_measured --iterations 720 > $out/bench_nee15_measured.json 2>> $out/bench.err || exit 1
timeout -k 10 200 python3 bench.py --no-cpu --width 3840 --height 2160 --iterations 256 > $out/bench_nee15_4k.json 2>> $out/bench.err || exit 1
timeout -k 10 300 python3 bench.py --no-cpu --scene synthetic10k --iterations 63 --steps 1 > $out/bench_synthetic10k.json 2>> $out/bench.err || exit 1
ls $out $out/trace
timeout -k 10 200 python3 bench.py --no-cpu --pipeline wavefront --iterations 240 > $out/bench_nee15_wavefront.json 2>> $out/bench.err || exit 1
timeout -k 10 200 python3 bench.py --no-cpu --scene light_room32 --iterations 720 > $out/bench_light_room32.json 2>> $out/bench.err || exit 1
timeout -k 10 200 python3 bench.py --no-cpu --scene bidir32 --iterations 720 > $out/bench_bidir32.json 2>> $out/bench.err || exit 1
